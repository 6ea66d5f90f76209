#!/bin/bash
# the MLP GPU tests, then the kernel trace of the headline loop.  usage: bash tools/mlp_trace.sh <tag> [pytest -k expression]
TAG=${1:-r05}
KEXPR=${2:-"mlp or posmlp or split_operand or two_piece or sines"}
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$KEXPR" 2>&1 | tail -8
bash tools/trace_pos_mlp.sh $TAG 2>&1 | tail -45
