#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel trace + stats of the same command,
# PMC passes (separate runs, counters only with --kernel-trace), A/B of the kernel variants and the
# diagnostic tools.  Everything lands under gpurun_out/prof_<tag>/ ; summarise afterwards with
#   python tools/pmc_summary.py gpurun_out/prof_<tag> C3 profiles/<tag>_C3_rocprof
# usage: tools/profile_round.sh <tag> [quick]      (WL=C5 tools/profile_round.sh ... for another workload)
set -u
tag=${1:-round}
quick=${2:-}
WL=${WL:-C3}
out=gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 300 python bench.py --workload $WL > "$out/bench.log" 2>&1
# (the headline loop only -- one stream, launches back to back: the trace then holds the durations of single launches,
# which is what roofline.kernel_ms of the bench line is; the secondary blocks overlap launches or render other cameras)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py --workload $WL --no-cpu-baseline --no-secondary > "$out/bench_under_rocprof.log" 2>&1
pmc() { # <dir> <counters...>
	d=$1; shift
	rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$d" -- python tools/prof_run.py $WL leap 30 > /dev/null 2>&1
}
pmc fetch FETCH_SIZE TCC_HIT_sum
pmc write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pmc sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
pmc sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
if [ -z "$quick" ]; then
timeout -k 10 400 python tools/variants_bench.py C3 C3h C2 C5 C4 2>&1 | grep -v "amdgpu.ids" > "$out/variants.log"
timeout -k 10 300 python tools/iter_map.py C3 2>&1 | grep -v amdgpu.ids > "$out/iter_map.log"
timeout -k 10 300 python tools/e2e_time.py C3 2>&1 | grep -v amdgpu.ids > "$out/e2e.log"
timeout -k 10 300 python tools/experiments.py 2>&1 | grep -v amdgpu.ids | grep leap > "$out/experiments.log"
timeout -k 10 300 python tools/gw_bench.py 2>&1 | grep -v amdgpu.ids > "$out/gw_bench.log"
fi
grep "^{" "$out/bench.log" | cut -c1-300
