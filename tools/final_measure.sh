#!/bin/bash
# Everything profiles/r3_* quotes for the current tree.  Two calls (each fits one gpurun call), same tree:
#   tools/final_measure.sh <tag> a   counter passes of C2/C3/C4/C5 (-> r3_pmc_roofline_*.json), then the bench lines that read
#                                    them: default (C2 with the CPU baseline), C3, C4, C5, C2 at the reference's depth 5,
#                                    rocprofv3 kernel stats of bench.py itself
#   tools/final_measure.sh <tag> b   C1 tools, lane profiles, interactive latency, TA/TD pass of C3
set -o pipefail
tag=${1:-final}; part=${2:-a}; out=gpurun_out/$tag; mkdir -p $out
if [ $part = a ]; then
  bash tools/pmc_roofline_all.sh $tag || exit 1
  timeout -k 10 400 python -u bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; }
  for wl in c3 c4 c5; do
    steps=5; [ $wl = c5 ] && steps=2; [ $wl = c4 ] && steps=3
    timeout -k 10 300 python -u bench.py --no-cpu-baseline --workload $wl --steps $steps > $out/bench_$wl.json 2> $out/bench_$wl.err || { echo "$wl failed"; tail -3 $out/bench_$wl.err; exit 1; }
  done
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --bounces 5 > $out/bench_c2_bounces5.json 2> $out/bench_c2_bounces5.err || exit 1
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/bench_under_rocprof.err ) || { tail -5 $out/bench_under_rocprof.err; exit 1; }
  python - <<PY
import json
for n in ("default", "c3", "c4", "c5", "c2_bounces5", "under_rocprof"):
    d = json.load(open("$out/bench_%s.json" % n)); r = d["roofline"]
    print("%-14s %9.0f Msamples/s %9.3f ms/step  bound %s frac %.3f frac_executed %.3f traffic %s" % (n, d["value"], d["ms_per_step"], r["bound"], r["frac"], r["frac_executed"], r["traffic"]))
PY
else
  timeout -k 10 120 python -u tools/prof_driver.py c1 5 64 0 > $out/c1_auto.txt 2>&1 || exit 1
  for t in c1_full c1_rate c1_floor c1_sweep c1_lanes; do
    timeout -k 10 300 python -u tools/$t.py > $out/$t.txt 2>&1 || { echo "$t failed"; tail -5 $out/$t.txt; exit 1; }
  done
  timeout -k 10 200 python -u tools/lane_profile.py 1024 c2 > $out/lanes_c2.txt 2>&1 || exit 1
  timeout -k 10 200 python -u tools/lane_profile.py 256 c3 > $out/lanes_c3.txt 2>&1 || exit 1
  timeout -k 10 300 python -u tools/interactive_latency.py > $out/interactive_latency.txt 2>&1 || exit 1
  bash tools/pmc_ta.sh $tag/ta_c3 c3 256 > $out/ta_c3.txt 2>&1 || { tail -5 $out/ta_c3.txt; exit 1; }
  # when the path kernel's waves end: uniform blocks (128 / 64) against the AUTO size, whose last block is a quarter block
  for cfg in "c1 64 0" "c2 1024 128" "c2 1024 0" "c3 256 64" "c3 256 0"; do
    timeout -k 10 120 python -u tools/wave_timeline.py $cfg >> $out/wave_timeline.txt 2>&1 || { tail -5 $out/wave_timeline.txt; exit 1; }
  done
  tail -3 $out/c1_rate.txt; tail -12 $out/ta_c3.txt
fi
