#!/bin/bash
# The 1-spp redraw loop (tools/interactive_latency.py --frames-only) with the in-tree library and with every variant under
# path_tracer_amd/lib/variants/ (IR_DEBUG_LIB), then a kernel trace of the in-tree library.  usage: tools/interactive_ab.sh <tag>
set -o pipefail
tag=${1:-interactive}
out=gpurun_out/$tag
mkdir -p $out
echo "== in-tree" > $out/latency.txt
timeout -k 10 120 python -u tools/interactive_latency.py --frames-only >> $out/latency.txt 2>&1 || { tail -5 $out/latency.txt; exit 1; }
for v in path_tracer_amd/lib/variants/*.so; do
  [ -e "$v" ] || continue
  echo "== $(basename $v .so)" >> $out/latency.txt
  IR_DEBUG_LIB=$PWD/$v timeout -k 10 120 python -u tools/interactive_latency.py --frames-only >> $out/latency.txt 2>&1 || { tail -5 $out/latency.txt; exit 1; }
done
cat $out/latency.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/interactive_latency.py --frames-only > $GRAFT_REPO_ROOT/$out/trace.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$out/trace.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# group by (kernel, grid size): the three frame sizes launch different grids
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(agg.items()):
    if len(v) >= 50:
        v.sort()
        print("%-60s grid %9s n %4d median %8.1f us" % (k, g, len(v), v[len(v)//2] / 1e3))
PY
