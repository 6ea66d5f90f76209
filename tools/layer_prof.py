"""Launches of the split-operand forward / input-gradient kernels in the layouts the pos_mlp iteration uses (packed sines), for rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
if len(sys.argv) > 1:
    from materialist_amd import _lib
    _lib.load().matpbr_mlp_set_lds_dma(int(sys.argv[1]))
M, N, K = 512 * 512, 256, 256
x = torch.randn(M, 256, device=dev)
w = torch.randn(N, 256, device=dev) / 16
b = torch.randn(N, device=dev)
g = torch.randn(M, 256, device=dev)
s = torch.empty(M, 256, device=dev)
gp, db = torch.empty(M, 256, device=dev), torch.empty(N, device=dev)
ws = ops.mlp_split_weights(w, N, K)
for _ in range(40):
    ops.mlp_layer_fwd_bx(x, ws, b, s, None, N, K, 6)
    ops.mlp_layer_bwd_input_bx(g, ws, s, gp, N, K, db, 6, packed=True)
    ops.mlp_layer_bwd_weight_bx(g, x, N, K, 6)
torch.cuda.synchronize()
