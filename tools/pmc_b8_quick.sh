#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the 8 x 512x512 one-phase loop (steady state), condensed: what tools/pmc_passes_r06.sh does for the step kernel alone
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
B8="bench.py --images-per-gpu 8 --mode fused_one_phase --no-extras --no-cpu-baseline"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_q -o p -- python3 $B8 --steps 40 --warmup 310 > /dev/null 2> $OUT/pmc_q.err
  python tools/summarize_rocprof.py $OUT/pmc_q | grep -a "pstep\|pwalk\|loss_sums2" | cut -c1-160
  rm -rf $OUT/pmc_q
done
