#!/bin/bash
# rocprofv3 kernel trace of the none-mode loop alone: 8 x 512x512 (BASELINE configs[2] per-GPU shard) and one image.
#   usage: bash tools/trace_fused.sh <tag>   -> gpurun_out/<tag>_trace_b8.csv, gpurun_out/<tag>_trace_b1.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r03}
OUT=gpurun_out
for B in 8 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$B -o t -- python3 bench.py --images-per-gpu $B --mode fused --no-extras --no-cpu-baseline --steps 400 --warmup 50 > $OUT/${TAG}_trace_b$B.json 2> $OUT/${TAG}_trace_b$B.err
  python tools/summarize_rocprof.py $OUT/tr_$B > $OUT/${TAG}_trace_b$B.csv
  rm -rf $OUT/tr_$B
done
head -20 $OUT/${TAG}_trace_b8.csv; head -20 $OUT/${TAG}_trace_b1.csv
