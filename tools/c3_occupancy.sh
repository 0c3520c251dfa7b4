#!/bin/bash
# C3 with the tree kernel's LDS allocation padded (IR_DEBUG_LDS_PAD): how much does the rate depend on resident waves?
# 25 KB per workgroup = 6 per CU; +2 KB -> 5 (27 KB x 6 > 160 KB), +8 KB -> 4, +16 KB -> 3.  usage: tools/c3_occupancy.sh <tag>
set -o pipefail
tag=${1:-c3occ}; out=gpurun_out/$tag; mkdir -p $out
for pad in 0 2048 8192 16384 0; do
  IR_DEBUG_LDS_PAD=$pad timeout -k 10 200 python -u bench.py --no-cpu-baseline --workload c3 --steps 3 > $out/pad$pad.json 2> $out/pad$pad.err || { echo "pad $pad failed"; tail -3 $out/pad$pad.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/pad$pad.json"))
print("LDS pad %-6s %8.0f Msamples/s  kernel %.3f ms" % ("$pad", d["value"], d["roofline"]["avg_launch_ms"]))
PY
done
