#!/bin/bash
# Profiles of the bench workload for profiles/: rocprofv3 kernel stats of bench.py itself, then PMC passes (own runs,
# --pmc with --kernel-trace only) of the same launch (C2, 1024 spp) through tools/prof_driver.py.
# usage (GPU box, repo root): tools/gpu_profile.sh <tag>
set -o pipefail
tag=${1:-prof}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err || { tail -5 $out/bench_under_rocprof.err; exit 1; }
cd $GRAFT_REPO_ROOT && bash tools/pmc_passes.sh $tag/pmc c2 1024
python3 tools/pmc_summary.py gpurun_out/$tag/pmc paths_kernel --json gpurun_out/$tag/pmc_roofline.json --source "rocprofv3 --pmc passes of the C2 launch (1920x1080x1024 spp, path kernel), tools/gpu_profile.sh $tag" > gpurun_out/$tag/pmc_summary.txt
cat gpurun_out/$tag/pmc_summary.txt | tail -12
