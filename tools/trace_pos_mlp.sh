#!/bin/bash
# rocprofv3 kernel trace of the headline loop alone (pos_mlp, 512x512): every kernel of 220 iterations, and the launch sequence of one.
#   usage: bash tools/trace_pos_mlp.sh <tag>   -> gpurun_out/<tag>_trace_pos_mlp_iteration.csv, gpurun_out/<tag>_pos_mlp_sequence.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r03}
OUT=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_p -o t -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/${TAG}_trace_pos_mlp.json 2> $OUT/${TAG}_trace_pos_mlp.err
python tools/summarize_rocprof.py $OUT/tr_p --all > $OUT/${TAG}_trace_pos_mlp_iteration.csv
python tools/iter_trace.py $(find $OUT/tr_p -name "*kernel_trace.csv" | head -1) mlp_chain_prep 100 > $OUT/${TAG}_pos_mlp_sequence.txt
rm -rf $OUT/tr_p
head -32 $OUT/${TAG}_trace_pos_mlp_iteration.csv | cut -c1-150; cat $OUT/${TAG}_pos_mlp_sequence.txt
