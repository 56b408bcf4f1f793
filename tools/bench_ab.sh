#!/bin/bash
# Calibrated headline A/B of prebuilt libhmrm.so variants (see tools/sched_ab.sh for how they are made): bench.py's own
# timed loop (static pose, calibrated launch order), headline only, interleaved.  usage: tools/bench_ab.sh "A K" [rounds]
cd "$(dirname "$0")/.."
P=heightmap-ray-marcher_amd
trap 'cp $P/_variants/libhmrm_A.so $P/libhmrm.so' EXIT
for r in $(seq 1 ${2:-2}); do
  for v in ${1:-A K}; do
    cp $P/_variants/libhmrm_$v.so $P/libhmrm.so
    timeout -k 5 40 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant $v round $r: ms_per_step %.5f kernel_ms %.5f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
