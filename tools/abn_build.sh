#!/bin/bash
# usage: [WLS="C3 C5"] [ROUNDS=2] tools/abn_build.sh "<flags A>" "<flags B>" ... -- interleaved n-way comparison of builds of
# libhmrm.so on the GPU box: kernel ms of the production kernel per workload (tools/variants_bench.py) and the heavy C3 row
# strip (tools/strip_time.py).  Restores the default build at the end.
set -e
# (a failing or timed-out variant run must not leave libhmrm.so built with experimental flags: ADVICE r03)
trap 'bash "$(dirname "$0")/sweep_build.sh" ""' EXIT
WLS="${WLS:-C3 C5 C2 C4}"
for round in $(seq 1 "${ROUNDS:-2}"); do
  for flags in "$@"; do
    bash "$(dirname "$0")/sweep_build.sh" "$flags"
    echo "=== build [$flags] round $round"
    VARIANTS=leap timeout -k 10 300 python tools/variants_bench.py $WLS 2>&1 | grep -E "median|diag"
    if [ -z "$NO_STRIPS" ]; then timeout -k 10 100 python tools/strip_time.py 2>&1 | grep -E "784.. 800|   0..2160"; fi
  done
done
