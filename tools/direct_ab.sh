#!/bin/bash
# usage: tools/direct_ab.sh <tag>: parity tests that render 1-spp launches, then tools/direct_crossover.py with the DIRECT kernel off / on
set -o pipefail
out=gpurun_out/${1:-direct}
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_sky_sampling.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
IR_DEBUG_DIRECT=0 timeout -k 10 200 python -u tools/direct_crossover.py > $out/crossover_off.txt 2>&1 || { tail -5 $out/crossover_off.txt; exit 1; }
IR_DEBUG_DIRECT=1 timeout -k 10 200 python -u tools/direct_crossover.py > $out/crossover_on.txt 2>&1 || { tail -5 $out/crossover_on.txt; exit 1; }
cat $out/crossover_off.txt $out/crossover_on.txt
