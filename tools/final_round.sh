set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_final2; mkdir -p "$out"
t0=$(date +%s)
timeout -k 10 900 python -m pytest tests -m gpu -q -x > "$out/gpu_suite.txt" 2>&1; rc=$?
t1=$(date +%s)
echo "pytest -m gpu: $((t1 - t0)) s wall, exit status $rc" | tee "$out/gpu_suite_wall.txt"
tail -5 "$out/gpu_suite.txt"
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 && \
python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc $?"; tail -c 600 "$out/bench.json"
