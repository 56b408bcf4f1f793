set -o pipefail
mkdir -p gpurun_out/r04
WLS="C3 C3h C5 C2 C4" bash tools/profile_r04.sh > gpurun_out/r04/profile_round.log 2>&1
timeout -k 10 500 python bench.py > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err; echo "bench rc $?"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_final_steps20.json 2> gpurun_out/r04/bench_final_steps20.err; echo "bench20 rc $?"
HMRM_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_dist1.json 2> gpurun_out/r04/bench_dist1.err; echo "dist rc $?"
timeout -k 10 200 python bench.py --gpus 2 --dry-launch --steps 3 --warmup 1 > gpurun_out/r04/bench_dry2.json 2> gpurun_out/r04/bench_dry2.err; echo "dry rc $?"
timeout -k 10 200 python bench.py --gpus 2 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/r04/bench_two_ranks_one_gpu.json 2> gpurun_out/r04/bench_two_ranks_one_gpu.err; echo "two ranks on a one-GPU box rc $? (expected: non-zero, rank 1 has no GPU)"
tail -3 gpurun_out/r04/bench_two_ranks_one_gpu.err
cat gpurun_out/r04/profile_round.log
