#!/bin/bash
# GPU tests + C2 bench only (fast iteration).  usage: tools/gpu_quick.sh <tag>
set -o pipefail
tag=${1:-q}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/gputests.log 2>&1
rc=$?
tail -15 $out/gputests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err || { tail -5 $out/bench_c2.err; exit 1; }
python - <<PY
import json
d=json.load(open("$out/bench_c2.json"))
print("C2 Msamples/s", round(d["value"]), "ms", round(d["ms_per_step"],2), "frac", round(d["roofline"]["frac"],4))
PY
