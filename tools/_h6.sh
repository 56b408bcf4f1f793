set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 500 python bench.py > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err; echo "bench rc $?"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_final_steps20.json 2> gpurun_out/r04/bench_final_steps20.err; echo "bench20 rc $?"
HMRM_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_dist1.json 2> gpurun_out/r04/bench_dist1.err; echo "dist rc $?"
