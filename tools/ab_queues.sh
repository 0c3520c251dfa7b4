#!/bin/bash
# Same box, same binary: C2 and C1 with the task queues forced to 1 / 4 / 16 (IR_DEBUG_TASK_QUEUES) and as launch_render
# picks them.  usage: tools/ab_queues.sh <tag>
set -o pipefail
tag=${1:-abq}; out=gpurun_out/$tag; mkdir -p $out
for q in 0 1 16 4 0 1 16; do
  IR_DEBUG_TASK_QUEUES=$q timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 5 > $out/c2_q$q.json 2> $out/c2_q$q.err || { echo "q$q failed"; tail -3 $out/c2_q$q.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/c2_q$q.json"))
print("C2 queues %-2s (0 = auto) %8.0f Msamples/s  %7.3f ms/step  kernel %.3f ms" % ("$q", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
done
for q in 0 1 16; do
  echo "C1 queues $q:"; IR_DEBUG_TASK_QUEUES=$q timeout -k 10 100 python -u tools/c1_lanes.py 16 8 4 | cut -c1-60 || exit 1
done
