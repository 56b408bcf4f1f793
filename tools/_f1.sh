set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 560 python tests/deep_fuzz.py 20270000 1000000 520 > gpurun_out/r04/fuzz_a.txt 2>&1; echo "deep_fuzz rc $?"; tail -2 gpurun_out/r04/fuzz_a.txt
timeout -k 10 560 python tests/deep_fuzz_big.py 20270000 1000000 520 4096 > gpurun_out/r04/fuzz_b.txt 2>&1; echo "deep_fuzz_big rc $?"; tail -2 gpurun_out/r04/fuzz_b.txt
