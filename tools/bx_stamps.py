"""Cycle stamps inside mlp_nt_bx (LDS-DMA loop): build with -DMATPBR_BX_STAMPS (tools/bx_stamps.sh), run one layer at 512 x 512.
Per super-step of workgroup 0's second tile, waves 0 and 4: issue of the DMA pieces / fragment reads + split + products / wait + barrier."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
M = 512 * 512
x = torch.randn(M, 256, device=dev)
w = torch.randn(256, 256, device=dev) / 16
b = torch.randn(256, device=dev)
g = torch.randn(M, 256, device=dev)
ws = ops.mlp_split_weights(w, 256, 256)
s = torch.empty(M, 256, device=dev)
gp, db = torch.empty(M, 256, device=dev), torch.empty(256, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for _ in range(5):
    if which == "fwd":
        ops.mlp_layer_fwd_bx(x, ws, b, s, None, 256, 256, 6)
    else:
        ops.mlp_layer_bwd_input_bx(g, ws, s, gp, 256, 256, db, 6, packed=True)
torch.cuda.synchronize()
out = np.zeros((2, 8, 8), dtype=np.uint64)
fn = lib.matpbr_debug_bx_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(out.ctypes.data) == 0
t = out.astype(np.int64)
for wv in range(2):
    print("wave", 4 * wv)
    for ks in range(8):
        r = t[wv, ks]
        nxt = t[wv, ks + 1, 0] if ks < 7 else r[3]
        print(f"  step {ks}: issue {r[1] - r[0]:5d}  reads+split+products {r[2] - r[1]:5d}  wait+barrier {r[3] - r[2]:5d}   (step {r[3] - r[0]:5d})")
    print("  epilogue", t[wv, 0, 4] - t[wv, 7, 3], " tile", t[wv, 0, 4] - t[wv, 0, 0])
