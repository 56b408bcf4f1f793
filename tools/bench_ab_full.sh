#!/bin/bash
# Like tools/bench_ab.sh with the whole default bench.py line per variant (every workload block): gpurun_out/bench_full_<v>.json
cd "$(dirname "$0")/.."
P=heightmap-ray-marcher_amd
trap 'cp $P/_variants/libhmrm_A.so $P/libhmrm.so' EXIT
for v in ${1:-A K}; do
  cp $P/_variants/libhmrm_$v.so $P/libhmrm.so
  timeout -k 5 40 python bench.py --no-cpu-baseline > gpurun_out/bench_full_$v.json 2> gpurun_out/bench_full_$v.err
  echo "variant $v rc $?"
done
