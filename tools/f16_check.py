"""The forward sine layer on two f16 pieces (nprod 3) against fp64, the exact-f32 MFMA kernel and the three-bf16-piece form (nprod 6):
errors on operands of several magnitudes (do the matrix cores keep f16 subnormals?), then hipEvent timings at 512 x 512.
usage: f16_check.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(5)


def errors(M, xscale, wscale, K=256, N=256):
    x = torch.sin(torch.randn(M, 256, device=dev) * 3) * xscale
    w = (torch.rand(N, 256, device=dev) * 2 - 1) * wscale
    b = torch.randn(N, device=dev) * 0.1
    pre = x[:, :K].double() @ w[:, :K].double().t() + b.double()
    ref = torch.sin(pre)
    out = {}
    s0, c0 = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    ops.mlp_layer_fwd(x, w, b, s0, c0, K)
    out["f32"] = s0
    for nprod in (6, 3):
        ws = ops.mlp_split_weights(w, N, K, f16=(nprod == 3))
        s1 = torch.empty(M, 256, device=dev)
        ops.mlp_layer_fwd_bx(x, ws, b, s1, None, N, K, nprod)
        out[nprod] = s1
    torch.cuda.synchronize()
    res = {}
    for k, v in out.items():
        e = (v[:, :N].double() - ref).abs()
        res[k] = (e.max().item(), e.pow(2).mean().sqrt().item())
    return res


for xs, wsc in ((1.0, 1 / 16), (1.0, 1e-3), (1e-2, 1 / 16), (1e-4, 1 / 16), (1e-4, 1.0), (1.0, 0.5), (500.0, 1 / 16)):
    r = errors(128 * 64, xs, wsc)
    print(f"x~{xs:g} w~{wsc:g}: " + "  ".join(f"{k}: max {v[0]:.3e} rms {v[1]:.3e}" for k, v in r.items()), flush=True)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = 512 * 512
x = torch.sin(torch.randn(M, 256, device=dev) * 3)
w = (torch.rand(256, 256, device=dev) * 2 - 1) / 16
b = torch.randn(256, device=dev) * 0.1
s = torch.empty(M, 256, device=dev)
for nprod in (6, 3, 6, 3):
    ws = ops.mlp_split_weights(w, 256, 256, f16=(nprod == 3))
    for _ in range(5):
        ops.mlp_layer_fwd_bx(x, ws, b, s, None, 256, 256, nprod)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        ops.mlp_layer_fwd_bx(x, ws, b, s, None, 256, 256, nprod)
    t1.record()
    torch.cuda.synchronize()
    print(f"nprod {nprod}: {t0.elapsed_time(t1) / reps * 1e3:.1f} us per forward layer (512 x 512, packed sines)", flush=True)
