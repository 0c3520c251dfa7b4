#!/bin/bash
# C1 under the microscope: per-kernel durations of the launch sequence for a few sample-block sizes, and the sustained rate.
# usage: tools/c1_probe.sh <tag>
set -o pipefail
tag=${1:-c1probe}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for blk in 0 8 4; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/$out/prof_b$blk -o c1 --output-format csv -- python3 $R/tools/prof_driver.py c1 20 64 $blk > $R/$out/run_b$blk.txt 2>&1 || { tail -5 $R/$out/run_b$blk.txt; exit 1; }
  f=$(ls $R/$out/prof_b$blk/*/c1_kernel_stats.csv 2>/dev/null | head -1)
  echo "== block $blk"; tail -1 $R/$out/run_b$blk.txt; [ -n "$f" ] && cut -d, -f1-5 "$f" | head -8
done
cd $R && timeout -k 10 200 python -u tools/c1_rate.py 200 > $out/rate.txt 2>&1; cat $out/rate.txt
