#!/bin/bash
# One-off wider fuzz: tests/test_gpu_fuzz.py under several seed shifts (240 cases each).  usage: tools/fuzz_wide.sh <tag> <shift> [<shift> ...]
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out/$tag
for s in "$@"; do
  IR_FUZZ_SHIFT=$s timeout -k 10 400 python -u -m pytest tests/test_gpu_fuzz.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/$tag/fuzz_$s.log 2>&1 || { echo "shift $s FAILED"; tail -30 gpurun_out/$tag/fuzz_$s.log; exit 1; }
  echo "shift $s: $(tail -1 gpurun_out/$tag/fuzz_$s.log)"
done
