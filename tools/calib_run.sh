#!/bin/bash
# GPU box: what the SQ VALU counters mean on gfx950 (tools/valu_calib.hip) -> gpurun_out/calib/;
# summarise with  python tools/calib_summary.py gpurun_out/calib profiles/<tag>_valu_calibration.txt
# usage: tools/calib_run.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/calib
rm -rf "$out"; mkdir -p "$out"
rocprofv3 -L > "$out/counters_list.txt" 2>&1
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/valu_calib tools/valu_calib.hip 2> /dev/null || exit 1
/tmp/valu_calib 8 20000 > "$out/plain_8waves.txt" 2>&1
/tmp/valu_calib 1 20000 > "$out/plain_1wave.txt" 2>&1
for w in 8 1; do
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc${w}" -- /tmp/valu_calib $w 20000 > "$out/pmc${w}.log" 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d "$out/cls8" -- /tmp/valu_calib 8 2000 > "$out/cls8.log" 2>&1
cat "$out/plain_8waves.txt"
