"""Cycle stamps inside mlp_nt_gx: build with -DMATPBR_BX_STAMPS (tools/bx_stamps.sh gx).  Workgroup 0's second tile, waves 0 and 1:
per half super-step: issue of the DMA pieces / fragment reads + split + products / wait + barrier."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.matpbr_mlp_set_lds_dma(2)
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
out = np.zeros(130, dtype=np.uint64)
fn = lib.matpbr_debug_bx_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(out.ctypes.data) == 0
t = out[:128].astype(np.int64).reshape(2, 8, 8)
e = out[128:].astype(np.int64)
for wv in range(2):
    print(which, "wave", wv)
    for g_ in range(8):
        r = t[wv, g_]
        print("  granule %d:" % g_, " | ".join(f"issue {r[4*h+1]-r[4*h]:5d} work {r[4*h+2]-r[4*h+1]:5d} wait {r[4*h+3]-r[4*h+2]:5d}" for h in range(2)),
              "  (granule %d)" % ((t[wv, g_ + 1, 0] if g_ < 7 else r[7]) - r[0]))
    print("  epilogue", e[wv] - t[wv, 7, 7], " tile", e[wv] - t[wv, 0, 0])
