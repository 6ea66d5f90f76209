"""hipEvent timing of the f16 weight-gradient and input-gradient kernels at 512 x 512 (MATPBR_LIB selects the library)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = 512 * 512
x = torch.sin(torch.randn(M, 256, device=dev) * 3)
w = torch.randn(256, 256, device=dev) / 16
g = torch.randn(M, 256, device=dev) * 1e-6
gp, db, dw = torch.empty(M, 256, device=dev), torch.empty(256, device=dev), torch.empty(256, 256, device=dev)
ws3t = ops.mlp_split_weights(w, 256, 256, transposed=True, f16=True)
tmx = g.abs().view(M // 128, -1).amax(1).contiguous().view(torch.int32)
tmo = ops.mlp_tile_max(M, dev)


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
    print(f"wgrad_blk {timed(lambda: ops.mlp_layer_bwd_weight_blk(g, tmx, x, 256, 256, out=dw)):.1f} us   "
          f"bwd_input_blk {timed(lambda: ops.mlp_layer_bwd_input_blk(g, tmx, ws3t, x, gp, 256, 256, db, tmo)):.1f} us", flush=True)
