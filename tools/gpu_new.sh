#!/bin/bash
# Selected GPU tests (files given as arguments) with a time limit each, then an optional C2 bench.  usage: tools/gpu_new.sh <tag> <test files...>
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python -u -m pytest "$@" -m gpu -x -v --timeout 300 --timeout-method thread > $out/tests.log 2>&1
rc=$?
tail -25 $out/tests.log
exit $rc
