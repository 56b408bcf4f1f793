set -o pipefail
mkdir -p gpurun_out/r04
bash tools/profile_r04.sh > gpurun_out/r04/profile_round.log 2>&1
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_b.json 2> gpurun_out/r04/bench_b.err; echo "bench rc $?"
HMRM_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04/bench_dist1.json 2> gpurun_out/r04/bench_dist1.err; echo "dist rc $?"
cat gpurun_out/r04/profile_round.log; du -sh gpurun_out/prof_r04_*
