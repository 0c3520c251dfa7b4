#!/bin/bash
# One rocprofv3 counter pass over the texture-addresser / L1 path (is a gather-heavy kernel bound by TA/TCP rather than VALU?).
# usage (GPU box, repo root): tools/pmc_ta.sh <out_dir_under_gpurun_out> [cfg] [spp] [kernel filter] [sample block, 0 = auto]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; CFG=${2:-c3}; SPP=${3:-64}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/ta -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $CFG 1 $SPP ${5:-0} > $OUT/ta.log 2>&1 || { echo "pass failed rc=$?"; tail -3 $OUT/ta.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT ${4:-paths_lbvh}
