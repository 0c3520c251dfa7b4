#!/bin/bash
# C3 A/B: LBVH integrator with 64-byte nodes, compact nodes without the LDS treelet, compact nodes with it; parity tests first.
set -o pipefail
tag=${1:-c3ab}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_lbvh_build.py tests/test_gpu_sky_sampling.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for mode in wide notreelet full; do
  case $mode in wide) export IR_DEBUG_WIDE_NODES=1;; notreelet) unset IR_DEBUG_WIDE_NODES; export IR_DEBUG_NO_TREELET=1;; full) unset IR_DEBUG_WIDE_NODES IR_DEBUG_NO_TREELET;; esac
  timeout -k 10 200 python -u bench.py --workload c3 --steps 3 --no-cpu-baseline > $out/$mode.json 2> $out/$mode.err || { echo "$mode failed"; tail -5 $out/$mode.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/$mode.json"))
print("%-10s %8.0f Msamples/s  %7.3f ms/step  useful flop/sample %.0f" % ("$mode", d["value"], d["ms_per_step"], d["roofline"]["algorithmic_flops_per_sample"]))
PY
done
