#!/bin/bash
# usage: tools/ab_build.sh "<flags A>" "<flags B>" [workloads...] -- interleaved A/B of two builds of libhmrm.so on the GPU box:
# kernel ms of the production kernel per workload (tools/variants_bench.py) and the C3 row strips (tools/strip_time.py)
set -e
trap 'bash "$(dirname "$0")/sweep_build.sh" ""' EXIT
A="$1"; B="$2"; shift 2
WLS="${@:-C3 C5 C2 C4}"
for round in 1 2; do
  for flags in "$A" "$B"; do
    bash "$(dirname "$0")/sweep_build.sh" "$flags"
    echo "=== build [$flags] round $round"
    VARIANTS=leap timeout -k 10 200 python tools/variants_bench.py $WLS 2>&1 | grep -E "median"
    timeout -k 10 100 python tools/strip_time.py 2>&1 | grep -E "784.. 800|0..2160| 800..2160"
  done
done
