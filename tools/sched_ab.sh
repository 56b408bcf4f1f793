#!/bin/bash
# Interleaved A/B of prebuilt libhmrm.so variants that differ only in the compiler's instruction scheduling of
# render_fast.hip (built here with EXTRA flags, shipped under heightmap-ray-marcher_amd/_variants/, git-ignored):
#   A  the shipped build            B  -mllvm -amdgpu-sched-strategy=max-ilp
#   G  -mllvm -amdgpu-sched-strategy=max-memory-clause      I  -mllvm -amdgpu-use-amdgpu-trackers
# usage (GPU box): tools/sched_ab.sh [rounds] [workloads...]; prints kernel ms of the production kernel per variant.
cd "$(dirname "$0")/.."
P=heightmap-ray-marcher_amd
rounds=${1:-2}; shift
WLS="${@:-C3 C2}"
trap 'cp $P/_variants/libhmrm_A.so $P/libhmrm.so' EXIT
for r in $(seq 1 $rounds); do
  for v in ${SCHED_VARIANTS:-A B G I}; do
    cp $P/_variants/libhmrm_$v.so $P/libhmrm.so
    if [ "$r" = 1 ] && [ "$v" != A ]; then
      timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|error" | sed "s/^/variant $v: /"
    fi
    VARIANTS=leap timeout -k 5 90 python tools/variants_bench.py $WLS 2>&1 | grep median | sed "s/^/variant $v round $r: /"
  done
done
