set -o pipefail
mkdir -p gpurun_out/r04
WLS="C3 C3h C5 C2 C4" bash tools/profile_r04.sh > gpurun_out/r04/profile_round.log 2>&1
tail -3 gpurun_out/r04/profile_round.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r04/gpu_suite_3.txt 2>&1; tail -12 gpurun_out/r04/gpu_suite_3.txt; cat gpurun_out/gpu_suite_wall.txt
