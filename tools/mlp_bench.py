"""Times the PosMLP MFMA kernels one by one at M = 512*512 (20 back-to-back launches between HIP events) next to the BLAS calls
they replace.  usage: python tools/mlp_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    M = 512 * 512
    x = torch.randn(M, 256, device=dev)
    w = torch.randn(256, 256, device=dev) / 16
    b = torch.randn(256, device=dev)
    s, c, g = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev), torch.randn(M, 256, device=dev)
    gp = torch.empty(M, 256, device=dev)
    db = torch.empty(256, device=dev)
    flop = 2.0 * M * 256 * 256
    rows = [
        ("mlp fwd  N=256 K=256 bias only", lambda: ops.mlp_layer_fwd(x, w, b, s, None, 256)),
        ("mlp fwd  N=256 K=256 sincos", lambda: ops.mlp_layer_fwd(x, w, b, s, c, 256)),
        ("mlp fwd  N=241 K=256 sincos", lambda: ops.mlp_layer_fwd(x, w[:241], b[:241], s, c, 256)),
        ("mlp fwd  N=256 K=16 sincos", lambda: ops.mlp_layer_fwd(x[:, :16], w[:, :16], b, s, c, 15)),
        ("mlp bwd_input n_prev=256 n_red=256", lambda: ops.mlp_layer_bwd_input(g, w, c, gp, 256, 256, db)),
        ("mlp bwd_input n_prev=241 n_red=256", lambda: ops.mlp_layer_bwd_input(g, w, c, gp, 241, 256, db)),
        ("mlp bwd_input n_prev=256 n_red=241", lambda: ops.mlp_layer_bwd_input(g, w, c, gp, 256, 241, db)),
        ("mlp bwd_weight N=256 K=256", lambda: ops.mlp_layer_bwd_weight(g, x, 256, 256)),
        ("torch addmm", lambda: torch.addmm(b, x, w.t())),
        ("torch mm (g @ w)", lambda: torch.mm(g, w)),
        ("torch sin", lambda: torch.sin(x)),
        ("torch copy", lambda: s.copy_(x)),
    ]
    for name, fn in rows + rows[:3]:
        us = timeit(fn)
        print(f"{name:42s} {us:8.1f} us   {flop / us / 1e6:7.1f} TFLOP/s-equivalent")


if __name__ == "__main__":
    main()
