#!/bin/bash
# render-kernel time vs waves per SIMD, forced by padding the workgroup's LDS allocation (4-wave workgroups):
# 160 KiB / pad = workgroups per CU.  usage: tools/occupancy_sweep.sh <out-file>
out=${1:-gpurun_out/occupancy.txt}
: > $out
for pad in 0 20000 26000 32000 40000 53000 80000; do
  echo "pad $pad" >> $out
  IR_DEBUG_LDS_PAD=$pad timeout -k 10 120 python3 tools/prof_driver.py c2 2 256 >> $out 2>&1 || exit 1
done
cat $out
