#!/bin/bash
# rocprofv3 passes of round 5 (run on the GPU box through gpurun; kernel trace alone, then one counter group per pass).
#   usage: bash tools/pmc_passes_r06.sh   -> condensed CSVs in gpurun_out/r06_*.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
B8="bench.py --images-per-gpu 8 --mode fused_one_phase --no-extras --no-cpu-baseline"   # the shard as ONE phase: the 8 x 512x512 launches the canonical bytes are defined on
B1="bench.py --mode fused --no-extras --no-cpu-baseline"
# 1. kernel trace of the 8 x 512x512 loop ALONE, the shard as ONE phase (the durations the roofline fractions are recomputed from), and of one image;
#    then the same shard as the product runs it: two groups of four images on two streams (loop.PipelinedBrdfPhase)
for B in 8 1; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$B -o t -- python3 bench.py --images-per-gpu $B --mode fused_one_phase --no-extras --no-cpu-baseline --steps 1000 --warmup 300 > $OUT/r06_trace_b$B.json 2> $OUT/r06_trace_b$B.err
  python tools/summarize_rocprof.py $OUT/tr_$B > $OUT/r06_trace_b$B.csv
  rm -rf $OUT/tr_$B
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_p -o t -- python3 bench.py --images-per-gpu 8 --mode fused --no-extras --no-cpu-baseline --steps 1000 --warmup 300 > $OUT/r06_trace_b8_two_groups.json 2> $OUT/r06_trace_b8_two_groups.err
python tools/summarize_rocprof.py $OUT/tr_p > $OUT/r06_trace_b8_two_groups.csv
rm -rf $OUT/tr_p
pass() {  # name, counters, program...
  local name=$1 ctr=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o p -- python3 "$@" > /dev/null 2> $OUT/pmc_$name.err
  python tools/summarize_rocprof.py $OUT/pmc_$name > $OUT/r06_pmc_$name.csv
  rm -rf $OUT/pmc_$name
}
# 2. counters (caches are flushed between dispatches under --pmc: the bytes are what a launch moves with nothing resident)
pass b8_FETCH_SIZE FETCH_SIZE $B8 --steps 40 --warmup 310
pass b8_WRITE_SIZE WRITE_SIZE $B8 --steps 40 --warmup 310
pass b1_FETCH_SIZE FETCH_SIZE $B1 --steps 40 --warmup 310
pass b1_WRITE_SIZE WRITE_SIZE $B1 --steps 40 --warmup 310
pass b8_sq "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" $B8 --steps 40 --warmup 310
pass b8_grbm "GRBM_GUI_ACTIVE" $B8 --steps 40 --warmup 310
# 3. the 256-wide layer kernels of the headline iteration as built (mlp_nt_gx forward / input gradient, mlp_wgrad_bx): matrix-pipe busy cycles, LDS
# conflicts, issue stalls, HBM bytes
gx() {
  local name=$1 ctr=$2
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/gxp_$name -o p -- python3 tools/bx_prof.py > /dev/null 2> $OUT/gxp_$name.err
  python tools/summarize_rocprof.py $OUT/gxp_$name --filter mlp_ > $OUT/r06_pmc_gx_$name.csv
  rm -rf $OUT/gxp_$name
}
gx sq1 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
gx sq2 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
gx FETCH_SIZE FETCH_SIZE
gx WRITE_SIZE WRITE_SIZE
PMC_ROUND=r06 python tools/pmc_to_traffic.py $OUT
ls -la $OUT/r06_*.csv
