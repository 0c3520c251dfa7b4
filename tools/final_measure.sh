#!/bin/bash
# Everything profiles/ quotes for the current tree, from one box: every BASELINE config, the default bench line, C1 tools,
# lane profiles, interactive latency, kernel stats + PMC passes of C2, PMC passes of C3.  usage: tools/final_measure.sh <tag>
set -o pipefail
tag=${1:-final}; out=gpurun_out/$tag; mkdir -p $out
bash tools/bench_all.sh $tag/bench || exit 1
timeout -k 10 400 python -u bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; }
for t in c1_full c1_rate c1_floor c1_sweep c1_lanes; do
  timeout -k 10 300 python -u tools/$t.py > $out/$t.txt 2>&1 || { echo "$t failed"; tail -5 $out/$t.txt; exit 1; }
done
timeout -k 10 200 python -u tools/lane_profile.py 1024 c2 > $out/lanes_c2.txt 2>&1 || exit 1
timeout -k 10 200 python -u tools/lane_profile.py 256 c3 > $out/lanes_c3.txt 2>&1 || exit 1
timeout -k 10 300 python -u tools/interactive_latency.py > $out/interactive_latency.txt 2>&1 || exit 1
bash tools/gpu_profile.sh $tag/prof || exit 1
bash tools/pmc_passes.sh $tag/pmc_c3 c3 256 || exit 1
python3 tools/pmc_summary.py gpurun_out/$tag/pmc_c3 paths_lbvh_kernel > $out/pmc_summary_c3.txt || exit 1
tail -8 $out/pmc_summary_c3.txt
tail -3 $out/c1_rate.txt; tail -1 $out/bench_default.json | cut -c1-600
