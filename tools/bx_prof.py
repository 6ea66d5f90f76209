"""A few launches of every split-operand sine-layer kernel (and the f32 ones beside them) for rocprofv3 counter passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 512 * 512, 256, 256
x = torch.randn(M, 256, device=dev)
w = torch.randn(N, 256, device=dev) / 16
b = torch.randn(N, device=dev)
g = torch.randn(M, 256, device=dev)
s, c = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
gp, db = torch.empty(M, 256, device=dev), torch.empty(N, device=dev)
ws = ops.mlp_split_weights(w, N, K)
for _ in range(4):
    ops.mlp_layer_fwd(x, w, b, s, c, K)
    ops.mlp_layer_fwd_bx(x, ws, b, s, c, N, K, 6)
    ops.mlp_layer_fwd_bx(x, ws, b, s, None, N, K, 6)                            # as the iteration runs it: sign-packed sines (mlp_nt_gx)
    ops.mlp_layer_bwd_input_bx(g, ws, s, gp, N, K, db, 6, packed=True)          # the input gradient on the packed sines (mlp_nt_gx<mul cos>)
    ops.mlp_layer_bwd_input_bx(g, ws, c, gp, N, K, db, 6)
    ops.mlp_layer_bwd_weight(g, x, N, K)
    ops.mlp_layer_bwd_weight_bx(g, x, N, K, 6)
    ops.mlp_layer_bwd_weight_bx(g, x, N, K, 9)
torch.cuda.synchronize()
