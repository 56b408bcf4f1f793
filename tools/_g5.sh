set -o pipefail
mkdir -p gpurun_out/r04
NO_STRIPS=1 ROUNDS=2 WLS="C3 C5 C2 C4 C5/needles C5/white C3/white C3/needles" timeout -k 10 1150 bash tools/abn_build.sh "-DHMRM_ALIGN=0" "-DHMRM_ALIGN=1" "-DHMRM_ALIGN=1 -DHMRM_GIVEUP=3" > gpurun_out/r04/align_ab.txt 2>&1
grep -E "===|median" gpurun_out/r04/align_ab.txt | tail -60
