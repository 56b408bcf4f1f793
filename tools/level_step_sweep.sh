set -e
for flags in "" "-DHMRM_LEVEL_STEP=1" "-DHMRM_LEVEL_STEP=1 -DHMRM_DOWN=2" "-DHMRM_LEVEL_STEP=1 -DHMRM_DOWN=3" ""; do
  bash tools/sweep_build.sh "$flags"
  echo "=== build [$flags]"
  VARIANTS=leap timeout -k 10 200 python tools/variants_bench.py C3 C5 C2 C4 2>&1 | grep -E "median|diag"
  timeout -k 10 100 python tools/strip_time.py 2>&1 | grep -E "784.. 800"
  STREAMS=3 timeout -k 10 100 python tools/streams_overlap.py C3 2>&1 | grep -E "stream"
done
bash tools/sweep_build.sh ""
