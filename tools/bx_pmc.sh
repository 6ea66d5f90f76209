#!/bin/bash
# counter passes over tools/bx_prof.py (GPU box, through gpurun): bash tools/bx_pmc.sh -> gpurun_out/bx_pmc_*.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
pass() {
  local name=$1 ctr=$2
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/bxp_$name -o p -- python3 tools/bx_prof.py > /dev/null 2> $OUT/bxp_$name.err
  python tools/summarize_rocprof.py $OUT/bxp_$name --filter mlp_ > $OUT/bx_pmc_$name.csv
  rm -rf $OUT/bxp_$name
}
pass sq1 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
pass sq2 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
pass sq3 "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_FLAT"
