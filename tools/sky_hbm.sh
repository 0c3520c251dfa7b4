#!/bin/bash
# The sky in HBM: C5 (env-lit S1, 1080p) at 4096 spp under the default 1024x512 sky and under an 8192x4096 one, nearest and
# bilinear: bench lines, then FETCH_SIZE / TCC hit-miss passes of one launch each.  usage: tools/sky_hbm.sh <tag>
set -o pipefail
tag=${1:-skyhbm}; out=gpurun_out/$tag; mkdir -p $out
for cfg in "1024x512 nearest" "1024x512 bilinear" "8192x4096 nearest" "8192x4096 bilinear"; do
  set -- $cfg
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --workload c5 --spp 4096 --steps 5 --warmup 2 --sky-size $1 --sky-filter $2 > $out/bench_c5_$1_$2.json 2> $out/bench_c5_$1_$2.err || { echo "bench $cfg failed"; tail -3 $out/bench_c5_$1_$2.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/bench_c5_$1_$2.json")); r=d["roofline"]
print("C5 4096 spp sky %-10s %-8s %8.0f Msamples/s  kernel %.3f ms  algorithmic %.0f GB/s  frac_executed %.3f" % ("$1", "$2", d["value"], r["avg_launch_ms"], r.get("algorithmic_gbs", r["achieved"]), r["frac_executed"]))
PY
done
for cfg in "8192x4096 nearest" "8192x4096 bilinear" "1024x512 nearest"; do
  set -- $cfg
  export IR_PROF_SKY=$1; if [ $2 = bilinear ]; then export IR_PROF_FILTER=bilinear; else unset IR_PROF_FILTER; fi
  O=$GRAFT_REPO_ROOT/$out/pmc_$1_$2; mkdir -p $O
  ( cd /tmp; export TMPDIR=/tmp
    timeout -s KILL 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $O/p4 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py c5 1 4096 128 > $O/p4.log 2>&1 || { echo "p4 failed"; exit 1; }
    timeout -s KILL 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $O/p5 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py c5 1 4096 128 > $O/p5.log 2>&1 || { echo "p5 failed"; exit 1; }
    timeout -s KILL 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py c5 1 4096 128 > $O/p1.log 2>&1 || { echo "p1 failed"; exit 1; }
    timeout -s KILL 180 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/p2 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py c5 1 4096 128 > $O/p2.log 2>&1 || { echo "p2 failed"; exit 1; }
  ) || exit 1
  echo "== PMC sky $1 $2"; python3 tools/pmc_summary.py $O paths_kernel | grep -E "FETCH|WRITE|TCC|kernel ms|WAIT_ANY|WAVE_CYCLES|lane util"
done
