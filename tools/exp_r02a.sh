#!/bin/bash
# round-2 experiments: occupancy cap, cost of the sky rows in the mix, two-pass on small launches
set -u
cd "$GRAFT_REPO_ROOT"
export HMRM_PASS1_TRIPS=0
run() { python tools/prof_run.py $1 leap 50 2>&1 | grep "kernel ms"; }
echo "== baseline"; run C3; run C5; run C2
bash tools/sweep_build.sh "-DHMRM_WAVES_PER_EU=8"; echo "== waves_per_eu 8"; run C3; run C5; run C2
bash tools/sweep_build.sh "-DHMRM_EXP_SKIP_ROWS_BELOW=700"; echo "== rows < 700 skipped"; run C3
bash tools/sweep_build.sh "-DHMRM_EXP_SKIP_ROWS_BELOW=1300"; echo "== rows < 1300 skipped"; run C3
bash tools/sweep_build.sh ""; 
unset HMRM_PASS1_TRIPS
echo "== two-pass on a small launch"; TRIPS=0,4,8,16,24 python tools/trips_sweep.py C1 2>&1 | grep trips
