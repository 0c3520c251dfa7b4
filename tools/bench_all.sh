#!/bin/bash
# Every BASELINE config on one GPU, no CPU baseline: C1 (tools/c1_full.py shape through bench --workload is not there, so prof_driver), C2, C3, C4, C5.
set -o pipefail
tag=${1:-ball}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 120 python -u tools/prof_driver.py c1 5 64 > $out/c1.txt 2>&1 || { tail -3 $out/c1.txt; exit 1; }
timeout -k 10 120 python -u tools/prof_driver.py c1 5 64 0 > $out/c1_auto.txt 2>&1 || { tail -3 $out/c1_auto.txt; exit 1; }
tail -1 $out/c1.txt; tail -1 $out/c1_auto.txt
for wl in c2 c3 c4 c5; do
  steps=5; [ $wl = c5 ] && steps=2; [ $wl = c4 ] && steps=3
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --workload $wl --steps $steps > $out/$wl.json 2> $out/$wl.err || { echo "$wl failed"; tail -3 $out/$wl.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/$wl.json"))
print("%-3s %9.0f Msamples/s  %8.3f ms/step  frac %.3f" % ("$wl", d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
done
