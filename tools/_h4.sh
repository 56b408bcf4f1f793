set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r04/gpu_suite_2.txt 2>&1; tail -16 gpurun_out/r04/gpu_suite_2.txt; cat gpurun_out/gpu_suite_wall.txt
