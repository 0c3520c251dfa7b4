#!/bin/bash
# Same box: C1 sustained rate (tools/c1_rate.py, auto block line and block 4 line) of the in-tree library and of every
# variant under path_tracer_amd/lib/variants/, twice, alternating.  usage: tools/ab_c1.sh
set -o pipefail
for round in 1 2; do
  echo "in-tree:"; timeout -k 10 100 python -u tools/c1_rate.py 200 | grep -E "auto block|block 4:" || exit 1
  for v in path_tracer_amd/lib/variants/*.so; do
    [ -e "$v" ] || continue
    echo "$(basename $v .so):"; IR_DEBUG_LIB=$PWD/$v timeout -k 10 100 python -u tools/c1_rate.py 200 | grep -E "auto block|block 4:" || exit 1
  done
done
