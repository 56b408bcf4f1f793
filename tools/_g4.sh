set -o pipefail
mkdir -p gpurun_out/r04
NO_STRIPS=1 ROUNDS=2 WLS="C3 C5 C2 C3/white C5/needles C3/needles" timeout -k 10 1100 bash tools/abn_build.sh "-DHMRM_GIVEUP=0 -DHMRM_WAVE_GIVEUP=0" "-DHMRM_GIVEUP=4 -DHMRM_WAVE_GIVEUP=0" "-DHMRM_GIVEUP=0 -DHMRM_WAVE_GIVEUP=8" "-DHMRM_GIVEUP=4 -DHMRM_WAVE_GIVEUP=8" > gpurun_out/r04/giveup_ab.txt 2>&1
grep -E "===|median" gpurun_out/r04/giveup_ab.txt | tail -60
