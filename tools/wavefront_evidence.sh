#!/bin/bash
# rocprofv3 evidence for the schedule choice: per-stage kernel times and HBM-side bytes of IR_SCHEDULE_WAVEFRONT next to
# IR_SCHEDULE_REGENERATION on C2 and C3 (256 spp).  usage (GPU box, repo root): tools/wavefront_evidence.sh <tag>
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-wf}
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for cfg in c2 c3; do for sch in regen wavefront; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${cfg}_${sch}_stats -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $cfg 1 256 128 $sch > $out/${cfg}_${sch}.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${cfg}_${sch}_fetch -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $cfg 1 256 128 $sch > /dev/null 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${cfg}_${sch}_write -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $cfg 1 256 128 $sch > /dev/null 2>&1 || exit 1
  echo "$cfg $sch done: $(tail -1 $out/${cfg}_${sch}.log)"
done; done
