"""Median duration per (kernel, grid size) of a rocprofv3 --kernel-trace CSV.  usage: python tools/trace_summary.py <dir with *_kernel_trace.csv> [min_calls]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"].replace("pt::(anonymous namespace)::", "").replace("(pt::RenderArgs)", "")[:70], r["Grid_Size_X"], r["Workgroup_Size_X"])].append(
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g, wg), v in sorted(agg.items()):
    if len(v) >= min_calls:
        v.sort()
        print("%-70s grid %9s wg %4s n %4d median %9.1f us  min %9.1f" % (k, g, wg, len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
