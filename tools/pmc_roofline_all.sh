#!/bin/bash
# The per-workload counter files bench.py's roofline object reads (profiles/r3_pmc_roofline_{c2,c3,c4,c5}.json): five --pmc
# passes (own runs) of one launch of each workload at the launch size bench.py uses, summarised by tools/pmc_summary.py.
# The JSONs are written under gpurun_out/<tag>/ (copy them to profiles/ afterwards) and into profiles/ of the tree on the
# box, so that bench lines run later in the same call read them.  usage (GPU box, repo root): tools/pmc_roofline_all.sh <tag>
set -o pipefail
tag=${1:-roof}; out=gpurun_out/$tag; mkdir -p $out
one() {  # key cfg spp block kernel bound limiter
  bash tools/pmc_passes.sh $tag/pmc_$1 $2 $3 $4 || return 1
  python3 tools/pmc_summary.py $out/pmc_$1 $5 --bound $6 --limiter "$7" --json $out/r3_pmc_roofline_$1.json \
    --source "rocprofv3 --pmc passes (separate runs) of one $2 launch, $3 spp, sample block $4 (0 = auto), tools/pmc_roofline_all.sh $tag" > $out/r3_pmc_summary_$1.txt || return 1
  cp $out/r3_pmc_roofline_$1.json profiles/r3_pmc_roofline_$1.json
  echo "== $1"; grep -E "FETCH_SIZE|WRITE_SIZE|lane utilisation|kernel ms" $out/r3_pmc_summary_$1.txt
}
one c2 c2 1024 0 paths_kernel valu "VALU issue: the launch's instruction mix priced with the per-opcode issue costs measured on this chip (tools/ubench_valu.hip) accounts for the kernel's cycles; 2.6 cycles per VALU wave-instruction and SIMD, 0.68 of the lanes active" &&
one c3 c3 256 0 paths_lbvh valu "L1 data path (TD busy 0.85 of the kernel's cycles, tools/pmc_ta.sh) with VALU issue close behind; the object prices the launch against the FP32 vector peak" &&
one c4 c4 4096 0 paths_kernel valu "VALU issue, as C2 (same kernel and scene at 3840x2160)" &&
one c5 c5 4096 0 paths_kernel valu "VALU issue; every sample ends in a sky lookup (1024x512 sky: 8 MB, L2/Infinity-Cache resident)"
