#!/bin/bash
# HBM-side traffic of the split-operand PosMLP kernels (GPU box, through gpurun): bash tools/bx_pmc_traffic.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/bxt_$c -o p -- python3 tools/bx_prof.py > /dev/null 2> $OUT/bxt_$c.err
  python tools/summarize_rocprof.py $OUT/bxt_$c --filter mlp_ > $OUT/r02_pmc_bx_$c.csv
  rm -rf $OUT/bxt_$c
done
