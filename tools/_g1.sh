set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "hostile or several_scenes or baseline_config_full_frame or launch_order or many_streams" > gpurun_out/r04/t1.txt 2>&1; echo "tests rc $?" >> gpurun_out/r04/t1.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_a.json 2> gpurun_out/r04/bench_a.err; echo "bench rc $?" >> gpurun_out/r04/t1.txt
HMRM_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04/bench_dist1.json 2> gpurun_out/r04/bench_dist1.err; echo "dist rc $?" >> gpurun_out/r04/t1.txt
tail -3 gpurun_out/r04/t1.txt
