#!/bin/bash
# rocprofv3 counter passes for the render kernel (one --pmc run per pass; <= 8 SQ / 4 TCC / 2 GRBM counters each).
# usage (on the GPU box, from the repo root): tools/pmc_passes.sh <out_dir_under_gpurun_out> [cfg] [spp] [sample block, 0 = auto]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; CFG=${2:-c2}; SPP=${3:-256}; BLK=${4:-128}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $CFG 1 $SPP $BLK > $OUT/$name.log 2>&1 || { echo "pass $name failed rc=$?"; tail -3 $OUT/$name.log; return 1; }
  echo "pass $name ok"; }
run p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE &&
run p2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA &&
run p3 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_SMEM &&
run p4 FETCH_SIZE TCC_HIT_sum &&
run p5 WRITE_SIZE TCC_MISS_sum
