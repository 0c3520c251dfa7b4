#!/bin/bash
# The list of path pixels dealt out from its start (IR_DEBUG_TASK_ORDER=0) against from its end (the default): C1 .. C5.
# usage: tools/order_ab.sh <tag>
set -o pipefail
tag=${1:-order}; out=gpurun_out/$tag; mkdir -p $out
for o in 0 1 0 1; do
  export IR_DEBUG_TASK_ORDER=$o
  timeout -k 10 120 python -u tools/prof_driver.py c1 5 64 0 2>/dev/null | tail -1 | sed "s/^/order $o: /"
  for wl in c2 c3 c4 c5; do
    steps=5; extra=""; [ $wl = c5 ] && extra="--spp 4096"; [ $wl = c4 ] && steps=2
    timeout -k 10 300 python -u bench.py --no-cpu-baseline --workload $wl --steps $steps $extra > $out/${wl}_$o.json 2> $out/${wl}_$o.err || { echo "$wl failed"; exit 1; }
    python - <<PY
import json
d=json.load(open("$out/${wl}_$o.json")); print("order $o: %s %8.0f Msamples/s" % ("$wl", d["value"]))
PY
  done
done
