"""hipEvent timing of the output layer's backward pass (mlp_skinny_tn_kernel<8, dgrad> + its reduction) at 512 x 512 (MATPBR_LIB selects the library)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 512 * 512
pre = torch.randn(M, 256, device=dev) * 3
s = torch.sin(pre)
s = ((s.view(torch.int32) & ~1) | (torch.cos(pre) < 0).to(torch.int32)).view(torch.float32).contiguous()
d_x = torch.zeros(M, 8, device=dev)
d_x[:, :5] = torch.randn(M, 5, device=dev) * 1e-6
w = torch.randn(5, 256, device=dev) / 16
g_prev, gw, gb, gbp = torch.empty(M, 256, device=dev), torch.empty(5, 256, device=dev), torch.empty(8, device=dev), torch.empty(256, device=dev)
tm = ops.mlp_tile_max(M, dev)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for _ in range(2):
    print(f"out_layer_bwd_tmax {timed(lambda: ops.mlp_out_layer_bwd_tmax(d_x, s, w, g_prev, tm, gw, gb, gbp, 5, 256)):.1f} us", flush=True)
