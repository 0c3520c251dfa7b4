#!/bin/bash
# Whole GPU suite (as the round's end runs it), then the no-alloc test alone in a fresh process, then smoke.  usage: tools/gpu_all.sh <tag>
set -o pipefail
tag=${1:-all}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/gputests.log 2>&1 &&
timeout -k 10 200 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > $out/noalloc_alone.log 2>&1 &&
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
rc=$?
tail -4 $out/gputests.log; tail -3 $out/noalloc_alone.log; cat $out/smoke.log
exit $rc
