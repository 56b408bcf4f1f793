#!/bin/bash
# Record kernel: positions per group (HMRM_GROUP_REC) and back-off cap (HMRM_REC_BACKOFF), interleaved builds, C3 / C5 over
# needles and white noise (tools/rec_bench.py).  Restores the default build at the end.
set -u
trap 'bash "$(dirname "$0")/sweep_build.sh" ""' EXIT
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_rec_ab; mkdir -p "$out"
for round in 1 2; do
  for flags in "" "-DHMRM_GROUP_REC=4" "-DHMRM_GROUP_REC=8" "-DHMRM_REC_BACKOFF=4" "-DHMRM_REC_BACKOFF=8"; do
    bash tools/sweep_build.sh "$flags"
    echo "=== build [$flags] round $round"
    KINDS=needles,white timeout -k 10 300 python tools/rec_bench.py C3 C5 2>&1 | grep -E "^C[35]"
  done
done > "$out/rec_ab.txt" 2>&1
cat "$out/rec_ab.txt"
