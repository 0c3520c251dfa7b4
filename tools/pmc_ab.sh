#!/bin/bash
# Two SQ counter passes of one C3 (or C2) launch for the in-tree library and for each variant named on the command line.
# usage: tools/pmc_ab.sh <tag> <cfg> <spp> <block> [variant.so ...]
set -u
tag=$1; cfg=$2; spp=$3; blk=$4; shift 4
for v in base "$@"; do
  name=$(basename $v .so)
  if [ "$v" = base ]; then unset IR_DEBUG_LIB; else export IR_DEBUG_LIB=$GRAFT_REPO_ROOT/path_tracer_amd/lib/variants/$v; fi
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag/$name; mkdir -p $OUT
  ( cd /tmp; export TMPDIR=/tmp
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $cfg 1 $spp $blk > $OUT/p1.log 2>&1 || { echo "p1 $name failed"; exit 1; }
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $cfg 1 $spp $blk > $OUT/p2.log 2>&1 || { echo "p2 $name failed"; exit 1; }
  ) || exit 1
  echo "== $name"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT $([ $cfg = c3 ] && echo paths_lbvh || echo paths_kernel) 2>&1 | tail -40
done
