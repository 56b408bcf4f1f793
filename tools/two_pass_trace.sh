#!/bin/bash
# GPU box: per-kernel durations of the two passes for a few hand-over points (rocprofv3 kernel trace).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/two_pass_trace
rm -rf "$out"; mkdir -p "$out"
for t in ${TRIPS:-0 16 24 48}; do
  HMRM_PASS1_TRIPS=$t rocprofv3 --kernel-trace --stats --output-format csv -d "$out/t$t" -- python tools/prof_run.py ${WL:-C3} leap 30 > "$out/t$t.log" 2>&1
  echo "== trips $t"; cat "$out"/t$t/*/*kernel_stats.csv | cut -d, -f1-4,6-7 | grep -E "k_render|k_march" | sed 's/hmrm::DevFrame.*"/..."/' | cut -c1-160
done
