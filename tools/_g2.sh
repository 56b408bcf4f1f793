set -o pipefail
mkdir -p gpurun_out/r04
for w in 0 6 7; do
  if [ $w = 0 ]; then flags=""; else flags="-DHMRM_PERSIST_WAVES_PER_EU=$w"; fi
  if [ $w = 0 ]; then flags="-DHMRM_PERSIST_WAVES_PER_EU=5"; fi
  bash tools/sweep_build.sh "$flags"
  echo "=== build [$flags]"
  timeout -k 10 300 python tools/persist_bench.py C3 C5 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04/persist1.txt 2>&1
bash tools/sweep_build.sh ""
tail -40 gpurun_out/r04/persist1.txt
