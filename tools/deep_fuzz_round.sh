#!/bin/bash
# A long run of the five GPU-vs-oracle fuzzers (tests/deep_fuzz*.py) on fresh seeds; output under gpurun_out/<tag>/.
# usage: tools/deep_fuzz_round.sh <tag> <first seed> <seconds per fuzzer>
tag=${1:-fuzz}; seed=${2:-30000000}; secs=${3:-200}
out=gpurun_out/$tag; mkdir -p "$out"
for f in deep_fuzz.py deep_fuzz_big.py deep_fuzz_edges.py deep_fuzz_binades.py deep_fuzz_cells.py; do
  extra=""; [ "$f" = deep_fuzz_big.py ] && extra="4096"
  timeout -k 10 $((secs + 120)) python tests/$f $seed 10000000 $secs $extra > "$out/$f.txt" 2>&1
  echo "$f: $(grep -v amdgpu.ids "$out/$f.txt" | tail -1)"
  grep -n "MISMATCH" -A 8 "$out/$f.txt" | head -40
done
