set -e
for L in 4 5 6; do
  bash tools/sweep_build.sh "-DHMRM_MIP_LEVELS=$L"
  echo "=== levels $L"
  VARIANTS=leap timeout -k 10 200 python tools/variants_bench.py C3 C5 C2 C4 2>&1 | grep -E "median|diag"
  timeout -k 10 100 python tools/strip_time.py 2>&1 | grep -E "784.. 800|0..2160| 800..2160"
done
