set -o pipefail
mkdir -p gpurun_out/r04
{
timeout -k 10 300 python tools/persist_bench.py C3 C5 C2 2>&1 | grep -v amdgpu.ids
CONFIGS="8:4096:-1,8:6144:-1,8:3072:-1,32:0:-1" timeout -k 10 300 python tools/persist_bench.py C3 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r04/persist2.txt 2>&1
timeout -k 10 400 python tools/attempt_diag.py C3 C5 C3/white > gpurun_out/r04/attempt_diag.txt 2>&1
timeout -k 10 500 python tools/content_bench.py C3 C5 > gpurun_out/r04/content_a.txt 2>&1
tail -30 gpurun_out/r04/persist2.txt; tail -12 gpurun_out/r04/content_a.txt
