cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_8 -o t -- python3 bench.py --images-per-gpu 8 --mode fused_arm --no-extras --no-cpu-baseline --steps 1000 --warmup 300 > $OUT/r06_trace_b8_arm.json 2> $OUT/r06_trace_b8_arm.err
python tools/summarize_rocprof.py $OUT/tr_8 > $OUT/r06_trace_b8_arm.csv
rm -rf $OUT/tr_8
head -12 $OUT/r06_trace_b8_arm.csv | cut -c1-200
