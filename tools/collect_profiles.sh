#!/bin/bash
# Copies what tools/final_measure.sh <tag> {a,b} left under gpurun_out/<tag> into profiles/ under the round's names.
# usage (here, after the gpurun calls): tools/collect_profiles.sh <tag> [round prefix, default r3]
tag=$1; r=${2:-r3}; o=gpurun_out/$tag; p=profiles
[ -f $o/bench_default.json ] && {
  cp $o/bench_default.json $p/${r}_bench_default.json
  for w in c3 c4 c5; do cp $o/bench_$w.json $p/${r}_bench_${w}_1gpu.json; done
  cp $o/bench_c2_bounces5.json $p/${r}_bench_c2_bounces5.json
  cp $o/bench_under_rocprof.json $p/${r}_bench_c2_under_rocprof.json
  cp $o/stats/*/*_kernel_stats.csv $p/${r}_kernel_stats.csv
  for w in c2 c3 c4 c5; do
    cp $o/${r}_pmc_roofline_$w.json $o/${r}_pmc_summary_$w.txt $p/
    for k in 1 2 3 4 5; do cp $(ls $o/pmc_$w/p$k/*/*counter_collection.csv | head -1) $p/${r}_pmc_${w}_p$k.csv; done
  done
  python3 - <<PY
import csv, glob, json
f = glob.glob("$o/stats/*/*_kernel_trace.csv")[0]
ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "paths_kernel<false" in r["Kernel_Name"] or "paths_kernelILb0" in r["Kernel_Name"]]
d = json.load(open("$o/bench_under_rocprof.json"))["roofline"]
timed = ms[-d["launches_averaged"]:]
open("$p/${r}_kernel_trace_timed_steps.txt", "w").write(
    "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline   (tools/final_measure.sh $tag a; MI355X)\n"
    "paths_kernel<false, false, 256, false> launches in order, ms (End - Start of the kernel trace): " + "  ".join("%.3f" % v for v in ms) + "\n"
    "the %d timed ones (after the warm-up launch) average %.3f ms in the trace; bench.py's roofline.avg_launch_ms of the same run "
    "(HIP events around each launch, ${r}_bench_c2_under_rocprof.json): %.3f ms\n" % (len(timed), sum(timed) / len(timed), d["avg_launch_ms"]))
print(open("$p/${r}_kernel_trace_timed_steps.txt").read())
PY
}
[ -f $o/c1_rate.txt ] && {
  for t in c1_full c1_rate c1_floor c1_sweep c1_lanes lanes_c2 lanes_c3 interactive_latency; do cp $o/$t.txt $p/${r}_$t.txt; done
  cp $o/ta_c3.txt $p/${r}_c3_ta_td_busy_256spp.txt
  [ -f $o/wave_timeline.txt ] && grep -v amdgpu.ids $o/wave_timeline.txt > $p/${r}_wave_timeline.txt
}
[ -f $o/wavefront_mis.txt ] && grep -v amdgpu.ids $o/wavefront_mis.txt > $p/${r}_wavefront_mis.txt
true
