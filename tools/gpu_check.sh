#!/bin/bash
# One GPU-box pass: GPU tests, smoke, C2 bench, C3 benches.  Every GPU step has its own time limit; the chain stops
# at the first failure.  Output under gpurun_out/<tag>/.
set -o pipefail
tag=${1:-run}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/gputests.log 2>&1 &&
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 &&
timeout -k 10 300 python -u bench.py > $out/bench_c2.json 2> $out/bench_c2.err &&
timeout -k 10 300 python -u bench.py --workload c3 --steps 3 > $out/bench_c3_lbvh.json 2> $out/bench_c3_lbvh.err &&
timeout -k 10 300 python -u bench.py --workload c3 --accel linear --spp 8 --steps 2 --no-cpu-baseline > $out/bench_c3_linear.json 2> $out/bench_c3_linear.err
rc=$?
tail -3 $out/gputests.log; cat $out/smoke.log; cat $out/bench_c2.json $out/bench_c3_lbvh.json $out/bench_c3_linear.json 2>/dev/null
exit $rc
