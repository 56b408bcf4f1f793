set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 400 python tests/deep_fuzz_edges.py 20270000 1000000 360 > gpurun_out/r04/fuzz_c.txt 2>&1; echo "edges rc $?"; tail -2 gpurun_out/r04/fuzz_c.txt
timeout -k 10 400 python tests/deep_fuzz_binades.py 20270000 1000000 360 > gpurun_out/r04/fuzz_d.txt 2>&1; echo "binades rc $?"; tail -2 gpurun_out/r04/fuzz_d.txt
