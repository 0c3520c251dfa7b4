#!/bin/bash
# rocprofv3 evidence for the schedule choice: per-stage kernel times and HBM-side bytes of IR_SCHEDULE_WAVEFRONT next to
# IR_SCHEDULE_REGENERATION on C2 and C3 (256 spp).  usage (GPU box, repo root): tools/wavefront_evidence.sh <tag>
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-wf}
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
# "wavefrontbig": the whole 256-spp render as ONE batch (512 Mi paths, ~41 GB of queues) and 256-sample blocks
for cfg in c2 c3; do for sch in regen wavefront wavefrontbig; do
  args="$cfg 1 256 128 $sch"
  [ $sch = wavefront ] && args="$cfg 1 256 128 wavefront 33554432"
  [ $sch = wavefrontbig ] && args="$cfg 1 256 256 wavefront 536870912"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${cfg}_${sch}_stats -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $args > $out/${cfg}_${sch}.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${cfg}_${sch}_fetch -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $args > /dev/null 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${cfg}_${sch}_write -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py $args > /dev/null 2>&1 || exit 1
  echo "$cfg $sch done"
done; done
