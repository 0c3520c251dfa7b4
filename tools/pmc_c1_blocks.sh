#!/bin/bash
# One SQ counter pass of the path kernel on the C1 frame at 1024 spp for two sample-block sizes: is a small block
# instruction-bound (more VALU work per sample) or stall-bound (same work, more cycles per instruction)?
# usage (GPU box, repo root): tools/pmc_c1_blocks.sh <out_dir_under_gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for blk in 4 16; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/b$blk/p1 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py c1 1 1024 $blk > $OUT/b$blk.log 2>&1 || { echo "pass b$blk failed"; tail -3 $OUT/b$blk.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b$blk/p2 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py c1 1 1024 $blk > $OUT/b$blk.2.log 2>&1 || { echo "pass b$blk p2 failed"; exit 1; }
  echo "== block $blk"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/b$blk paths_kernel
done
