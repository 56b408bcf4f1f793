set -o pipefail
mkdir -p gpurun_out/r04
ROUNDS=3 WLS="C3 C5 C2 C4" timeout -k 10 1100 bash tools/abn_build.sh "-DHMRM_PREFETCH=0" "-DHMRM_PREFETCH=1" > gpurun_out/r04/prefetch_ab.txt 2>&1
grep -E "===|median|rows" gpurun_out/r04/prefetch_ab.txt | tail -60
