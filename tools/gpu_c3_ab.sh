#!/bin/bash
# LBVH-touching GPU tests, then the C3 bench with the in-tree library and every variant (tools/ab_bench.sh).
# usage: tools/gpu_c3_ab.sh <tag> [pytest -k expression]
set -o pipefail
tag=${1:-c3ab}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/test_gpu_lbvh_build.py tests/test_gpu_parity.py tests/test_gpu_wavefront.py tests/test_gpu_edges.py tests/test_gpu_sky_sampling.py tests/test_gpu_fuzz.py -x -q --timeout 180 --timeout-method thread > $out/gputests.log 2>&1
rc=$?
tail -8 $out/gputests.log
[ $rc -ne 0 ] && exit $rc
tools/ab_bench.sh $tag --workload c3 --steps 3
