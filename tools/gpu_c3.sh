#!/bin/bash
# GPU tests + C3 LBVH bench.  usage: tools/gpu_c3.sh <tag>
set -o pipefail
tag=${1:-c3}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/gputests.log 2>&1
rc=$?
tail -8 $out/gputests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u bench.py --workload c3 --steps 3 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err || { tail -5 $out/bench_c3.err; exit 1; }
python - <<PY
import json
d=json.load(open("$out/bench_c3.json"))
r=d["roofline"]
print("C3 Msamples/s", round(d["value"]), "ms", round(d["ms_per_step"],2), "flops/sample", round(r["algorithmic_flops_per_sample"]), "frac", round(r["frac"],4))
PY
