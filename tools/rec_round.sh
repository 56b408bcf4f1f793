#!/bin/bash
# Round 5: the record kernel (HMRM_KERNEL=rec) -- parity subset with "rec" among the variants, the fuzzers under it, then
# its time against the plain groups and the production kernel on every content kind.
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_rec; mkdir -p "$out"
(HMRM_FUZZ_BUDGET_S=40 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "bit_exact or baseline or hostile or fuzz or degenerate or binade or edges or bilinear or record or probe" 2>&1 | tail -15) > "$out/tests_rec.txt" 2>&1
tail -3 "$out/tests_rec.txt"
grep -q " passed" "$out/tests_rec.txt" && ! grep -q "failed" "$out/tests_rec.txt" || exit 1
for fz in deep_fuzz.py deep_fuzz_cells.py deep_fuzz_edges.py deep_fuzz_binades.py; do
  echo "== $fz under HMRM_KERNEL=rec"
  HMRM_KERNEL=rec timeout -k 10 200 python tests/$fz 1 100000 60 2>&1 | grep -v amdgpu.ids | tail -4
done > "$out/fuzz_rec.txt" 2>&1
cat "$out/fuzz_rec.txt"
grep -q "MISMATCH" "$out/fuzz_rec.txt" && exit 1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/build_prof" -- python3 "$GRAFT_REPO_ROOT/tools/records_build_time.py" 4096 > "$GRAFT_REPO_ROOT/$out/build_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find "$out/build_prof" -name "*kernel_stats.csv" | head -1 | xargs -r cat | cut -c1-160 > "$out/build_kernel_stats.csv"
cat "$out/build_kernel_stats.csv"
timeout -k 10 500 python tools/rec_bench.py C3 C5 > "$out/rec_bench.txt" 2>&1
cat "$out/rec_bench.txt"
