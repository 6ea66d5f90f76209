"""Accuracy and speed of the split-operand (bf16 x bf16 -> f32) sine-layer kernels against the exact-f32 MFMA kernels and an fp64 product.
usage: python tools/bx_check.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import ops  # noqa: E402


def bench(fn, reps=20):
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
    torch.manual_seed(0)
    M = 512 * 512
    for N, K in ((256, 256), (241, 256), (256, 241)):
        x = torch.randn(M, 256, device=dev)
        x[:, 0] *= 50.0
        w = (torch.rand(N, K, device=dev) * 2 - 1) / 16
        wp = torch.zeros(N, 256, device=dev)
        wp[:, :K] = w
        b = torch.randn(N, device=dev) * 0.1
        # fp64 reference on a slice of rows
        rows = slice(0, 4096)
        pre = (x[rows, :K].double() @ w.double().t()) + b.double()
        outs = {}
        for name, nprod in (("f32", 0), ("bx6", 6), ("bx9", 9)):
            s, c = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
            if nprod == 0:
                fn = lambda: ops.mlp_layer_fwd(x, wp, b, s, c, K)
            else:
                ws = ops.mlp_split_weights(wp, N, K)
                fn = lambda: ops.mlp_layer_fwd_bx(x, ws, b, s, c, N, K, nprod)
            us = bench(fn)
            err_s = (s[rows, :N].double() - torch.sin(pre)).abs().max().item()
            err_c = (c[rows, :N].double() - torch.cos(pre)).abs().max().item()
            outs[name] = (us, err_s, err_c)
            print(f"fwd  N={N} K={K} {name}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s(f32-equivalent)  max|sin err| {err_s:.2e}  max|cos err| {err_c:.2e}")
        # backward product: g [M, 256] (n_red = K columns used) x wt [n_prev = N, n_red = K]
        g = torch.randn(M, 256, device=dev)
        cprev = torch.rand(M, 256, device=dev)
        ref = (g[rows, :K].double() @ wp[:, :K].double().t()) * cprev[rows, :N].double()
        for name, nprod in (("f32", 0), ("bx6", 6), ("bx9", 9)):
            gp = torch.empty(M, 256, device=dev)
            db = torch.empty(N, device=dev)
            if nprod == 0:
                fn = lambda: ops.mlp_layer_bwd_input(g, wp, cprev, gp, N, K, db)
            else:
                ws = ops.mlp_split_weights(wp, N, K)
                fn = lambda: ops.mlp_layer_bwd_input_bx(g, ws, cprev, gp, N, K, db, nprod)
            us = bench(fn)
            err = (gp[rows, :N].double() - ref).abs().max().item() / ref.abs().max().item()
            dbe = (db.double() - gp[:, :N].double().sum(0)).abs().max().item() / (gp[:, :N].double().sum(0).abs().max().item() + 1e-30)
            print(f"bwd  N={N} K={K} {name}: {us:7.1f} us  rel err {err:.2e}  bias-grad consistency {dbe:.2e}")

        # weight gradient: g^T x over all rows, against fp64 on a slab-sized prefix and against the f32 kernel on everything
        xin = torch.randn(M, 256, device=dev)
        if K < 256:
            xin[:, K:] = float("nan")                 # scratch columns must not leak
        gg = g.clone()
        if N < 256:
            gg[:, N:] = float("inf")
        ref_full = None
        for name, nprod in (("f32", 0), ("bx6", 6), ("bx9", 9)):
            fn = (lambda: ops.mlp_layer_bwd_weight(gg, xin, N, K)) if nprod == 0 else (lambda: ops.mlp_layer_bwd_weight_bx(gg, xin, N, K, nprod))
            us = bench(fn)
            dw = fn()
            if ref_full is None:
                ref_full = torch.zeros(N, K, dtype=torch.float64, device=dev)
                for c in range(0, M, 32768):
                    ref_full += gg[c:c + 32768, :N].double().t() @ xin[c:c + 32768, :K].double()
            err = (dw.double() - ref_full).abs().max().item() / ref_full.abs().max().item()
            print(f"wgrad N={N} K={K} {name}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s(f32-equivalent)  rel err {err:.2e}")


if __name__ == "__main__":
    main()
