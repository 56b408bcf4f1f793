#!/bin/bash
# After a change to the records' builder or the record kernel: the table against numpy, the record-kernel parity tests,
# the builder's time under rocprofv3, then the long fuzz under HMRM_KERNEL=rec.
# usage: tools/rec_check_round.sh <first seed> <seconds per fuzzer>
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_rec_check; mkdir -p "$out"
(HMRM_FUZZ_BUDGET_S=30 timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "window_records or record_kernel or hostile or probe" 2>&1 | tail -6) > "$out/tests.txt" 2>&1
cat "$out/tests.txt"
grep -q " passed" "$out/tests.txt" && ! grep -q "failed\|error" "$out/tests.txt" || exit 1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/build_prof" -- python3 "$GRAFT_REPO_ROOT/tools/records_build_time.py" 4096 > "$GRAFT_REPO_ROOT/$out/build_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find "$out/build_prof" -name "*kernel_stats.csv" | head -1 | xargs -r cat | cut -c1-140 | tee "$out/build_kernel_stats.csv"
bash tools/rec_fuzz_round.sh "${1:-40000000}" "${2:-100}"
