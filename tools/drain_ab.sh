#!/bin/bash
# Launch drain against the length of the last block: bench lines for C2 and C3 over sample-block sizes that leave a short
# last block (block-major task order: the launch ends on the tasks of the last block).  usage: tools/drain_ab.sh <tag>
set -o pipefail
tag=${1:-drain}; out=gpurun_out/$tag; mkdir -p $out
run() { wl=$1; sb=$2
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --workload $wl --steps 5 --sample-block $sb > $out/${wl}_$sb.json 2> $out/${wl}_$sb.err || { echo "$wl $sb failed"; tail -3 $out/${wl}_$sb.err; return 1; }
  python - <<PY
import json
d=json.load(open("$out/${wl}_$sb.json")); print("%s block %4d: %8.0f Msamples/s  kernel %.3f ms" % ("$wl", $sb, d["value"], d["roofline"]["avg_launch_ms"]))
PY
}
for sb in 128 124 120 114 128; do run c2 $sb || exit 1; done
for sb in 64 60 57 52 64; do run c3 $sb || exit 1; done
