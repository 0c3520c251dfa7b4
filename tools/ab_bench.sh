#!/bin/bash
# A/B: bench.py (C2, no CPU baseline) with the in-tree library and with every variant under path_tracer_amd/lib/variants/
# (IR_DEBUG_LIB).  usage: tools/ab_bench.sh <tag> [extra bench args]
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; lib=$2; shift 2
  IR_DEBUG_LIB=$lib timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 5 "$@" > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -3 $out/$name.err; return 1; }
  python - <<PY
import json
d=json.load(open("$out/$name.json"))
print("%-14s %8.0f Msamples/s  %7.3f ms/step  kernel %.3f ms" % ("$name", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
}
run base "" "$@" || exit 1
for v in path_tracer_amd/lib/variants/*.so; do
  [ -e "$v" ] || continue
  run $(basename $v .so) $PWD/$v "$@" || exit 1
done
