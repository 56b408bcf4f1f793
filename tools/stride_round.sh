#!/bin/bash
# Experiment (round 5): stride groups (-DHMRM_STRIDE=1) against the default build, interleaved, under a few level policies
# Needs the kernel sources of commit f191f00 (the HMRM_STRIDE block was measured slower and taken out again:
# profiles/r05_experiments.txt §8); on any other tree both builds are the default one.
# (HMRM_MIN_LEVEL / HMRM_FINEST_PAUSE: where the ray stops looking at windows and takes its steps in groups).
set -u
trap 'bash "$(dirname "$0")/sweep_build.sh" ""' EXIT
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_stride; mkdir -p "$out"
bash tools/sweep_build.sh "-DHMRM_STRIDE=1"
(HMRM_FUZZ_BUDGET_S=50 timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "bit_exact or baseline or hostile or fuzz or degenerate or binade or edges" 2>&1 | tail -4) > "$out/tests_stride.txt" 2>&1
tail -3 "$out/tests_stride.txt"
for round in 1 2; do
  for flags in "" "-DHMRM_STRIDE=1"; do
    bash tools/sweep_build.sh "$flags"
    for pol in "-1 -1" "1 0" "1 1" "2 0" "2 1" "2 2"; do
      set -- $pol
      if [ "$1" = "-1" ]; then unset HMRM_MIN_LEVEL HMRM_FINEST_PAUSE; else export HMRM_MIN_LEVEL=$1 HMRM_FINEST_PAUSE=$2; fi
      echo "=== build [$flags] min_level $1 finest_pause $2 round $round"
      VARIANTS=leap timeout -k 10 200 python tools/variants_bench.py C3 C3h 2>&1 | grep -E "median|diag"
    done
    unset HMRM_MIN_LEVEL HMRM_FINEST_PAUSE
    echo "=== build [$flags] other workloads round $round"
    VARIANTS=leap timeout -k 10 200 python tools/variants_bench.py C5 C2 C4 2>&1 | grep -E "median"
  done
done > "$out/stride_ab.txt" 2>&1
grep -E "build|median" "$out/stride_ab.txt"
