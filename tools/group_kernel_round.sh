#!/bin/bash
# The every-load kernel (HMRM_KERNEL=group: speculative groups, no leaps -- every height load of the reference is executed):
# group length / filtered-compare A/B (-DHMRM_GROUP_PLAIN=n, -DHMRM_FILTER32=0; interleaved builds) and one PMC round (fetch / write / SQ) of the default build on C3.
# Output: gpurun_out/r05_group/.
set -u
trap 'bash "$(dirname "$0")/sweep_build.sh" ""' EXIT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_group
mkdir -p "$out"
# builds to compare (override: BUILDS_STR="flagsA|flagsB|..."); "" = the tree's defaults
IFS='|' read -r -a BUILDS <<< "${BUILDS_STR:-|-DHMRM_FILTER32=0|-DHMRM_GROUP_PLAIN=8|-DHMRM_GROUP_PLAIN=12}"
[ ${#BUILDS[@]} -eq 0 ] && BUILDS=("")
for round in 1 2; do
  for flags in "${BUILDS[@]}"; do
    bash tools/sweep_build.sh "$flags"
    echo "=== build [$flags] round $round"
    VARIANTS=group timeout -k 10 300 python tools/variants_bench.py ${WLS:-C3 C5 C2} 2>&1 | grep -E "median"
  done
done > "$out/group_len_ab.txt" 2>&1
bash tools/sweep_build.sh ""
pmc() { # <outdir> <workload> <variant> <launches> <counters...>
	d=$1; wl=$2; v=$3; n=$4; shift 4
	rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$d" -- python tools/prof_run.py $wl $v $n > /dev/null 2>&1
}
p="$out/prof_C3_group"
rm -rf "$p"; mkdir -p "$p"
rocprofv3 --kernel-trace --stats --output-format csv -d "$p/trace" -- python tools/prof_run.py C3 group 10 > "$p/run_under_rocprof.log" 2>&1
pmc "$p/fetch" C3 group 6 FETCH_SIZE TCC_HIT_sum
pmc "$p/write" C3 group 6 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pmc "$p/sq" C3 group 6 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
echo "group kernel round done"
