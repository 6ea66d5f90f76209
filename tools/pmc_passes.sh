#!/bin/bash
# rocprofv3 PMC passes of round 2 (run on the GPU box through gpurun; one counter group per pass, kernel-trace only).
#   usage: bash tools/pmc_passes.sh   -> condensed CSVs in gpurun_out/r02_pmc_*.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
B8="bench.py --images-per-gpu 8 --mode fused --no-extras --no-cpu-baseline --steps 6 --warmup 2"
B1="bench.py --mode fused --no-extras --no-cpu-baseline --steps 6 --warmup 2"
pass() {  # name, counters, program...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o p -- python3 "$@" > /dev/null 2> $OUT/pmc_$name.err
  python tools/summarize_rocprof.py $OUT/pmc_$name > $OUT/r02_pmc_$name.csv
  rm -rf $OUT/pmc_$name
}
pass b8_FETCH_SIZE FETCH_SIZE $B8
pass b8_WRITE_SIZE WRITE_SIZE $B8
pass b1_FETCH_SIZE FETCH_SIZE $B1
pass b1_WRITE_SIZE WRITE_SIZE $B1
pass b8_sq "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" $B8
pass b8_grbm "GRBM_GUI_ACTIVE" $B8
pass env_FETCH_SIZE FETCH_SIZE tools/env_profile.py 8
pass env_WRITE_SIZE WRITE_SIZE tools/env_profile.py 8
ls -la $OUT/r02_pmc_*.csv
