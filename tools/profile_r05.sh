#!/bin/bash
# Round-5 profile round on the GPU box (via gpurun): the bench line, the rocprofv3 kernel trace + stats of the headline
# loop, separate PMC passes (counters only with --kernel-trace) for C3, C3h and -- fetch / write only -- the literal
# kernel on C3 (WLS="C3 C3h C5 C2 C4" for more workloads).  Summarise afterwards, here:
#   python tools/pmc_summary.py gpurun_out/prof_r05_C3 C3 profiles/r05_C3_rocprof      (likewise C3h, C3_literal, C3_group, C3_needles_rec)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pmc() { # <outdir> <workload> <variant> <launches> <counters...>
	d=$1; wl=$2; v=$3; n=$4; shift 4
	rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$d" -- python tools/prof_run.py $wl $v $n > /dev/null 2>&1
}
for WL in ${WLS:-C3 C3h}; do
	out=gpurun_out/prof_r05_$WL
	rm -rf "$out"; mkdir -p "$out"
	rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py --workload $WL --no-cpu-baseline --no-secondary --steps ${STEPS:-200} > "$out/bench_under_rocprof.log" 2>&1
	pmc "$out/fetch" $WL leap 30 FETCH_SIZE TCC_HIT_sum
	pmc "$out/write" $WL leap 30 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
	pmc "$out/sq" $WL leap 30 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
	pmc "$out/sq2" $WL leap 30 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
	pmc "$out/sq3" $WL leap 30 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
	echo "$WL done"
done
out=gpurun_out/prof_r05_C3_literal
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python tools/prof_run.py C3 simple 4 > "$out/run_under_rocprof.log" 2>&1
pmc "$out/fetch" C3 simple 4 FETCH_SIZE TCC_HIT_sum
pmc "$out/write" C3 simple 4 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pmc "$out/sq" C3 simple 4 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
echo "literal done"
# the every-load kernel: the speculative groups without leaps
out=gpurun_out/prof_r05_C3_group
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python tools/prof_run.py C3 group 10 > "$out/run_under_rocprof.log" 2>&1
pmc "$out/fetch" C3 group 6 FETCH_SIZE TCC_HIT_sum
pmc "$out/write" C3 group 6 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pmc "$out/sq" C3 group 6 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
echo "group done"
# the record kernel on the map it is for: groups + leaps over window records, C3's camera over needles on a plateau
out=gpurun_out/prof_r05_C3_needles_rec
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python tools/prof_run.py C3/needles rec 30 > "$out/run_under_rocprof.log" 2>&1
pmc "$out/fetch" C3/needles rec 20 FETCH_SIZE TCC_HIT_sum
pmc "$out/write" C3/needles rec 20 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pmc "$out/sq" C3/needles rec 20 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pmc "$out/sq2" C3/needles rec 20 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
echo "records done"
