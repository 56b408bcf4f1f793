#!/bin/bash
# Long fuzz of the record kernel: the five GPU-vs-oracle fuzzers under HMRM_KERNEL=rec, then the records' own fuzzer.
# usage: tools/rec_fuzz_round.sh <first seed> <seconds per fuzzer>
seed=${1:-40000000}; secs=${2:-120}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_rec_fuzz; mkdir -p "$out"
export HMRM_KERNEL=rec
for f in deep_fuzz.py deep_fuzz_big.py deep_fuzz_edges.py deep_fuzz_binades.py deep_fuzz_cells.py deep_fuzz_records.py; do
  extra=""; [ "$f" = deep_fuzz_big.py ] && extra="4096"
  timeout -k 10 $((secs + 120)) python tests/$f $seed 10000000 $secs $extra > "$out/$f.txt" 2>&1
  echo "$f: $(grep -v amdgpu.ids "$out/$f.txt" | tail -1)"
  grep -n "MISMATCH" -A 3 "$out/$f.txt" | head -20
done
