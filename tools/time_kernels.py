"""Time the shading kernels back-to-back with HIP events (dev tool).  usage: python tools/time_kernels.py [size] [spp] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from materialist_amd import ops, synthetic

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
sc = synthetic.make_scene(0, size, size)
t = lambda x: torch.from_numpy(np.stack([x] * B) if B > 1 else x).to(dev)
n = ops.normals_from_depth(t(sc.depth))
a, r, m, l = t(sc.albedo), t(sc.roughness), t(sc.metallic), t(sc.light)
out = ops.shade_fwd(a, r, m, n, l, spp)
d_out = torch.randn_like(out)


def timeit(f, reps=20):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


px = size * size * B
for name, f, byt in (
    ("fwd", lambda: ops.shade_fwd(a, r, m, n, l, spp), 44),
    ("bwd<mat>", lambda: ops.shade_bwd(a, r, m, n, l, d_out, spp, want_mat=True), 64),
    ("bwd<light>", lambda: ops.shade_bwd(a, r, m, n, l, d_out, spp, want_mat=False, want_light=True), 44),
    ("bwd<mat,light>", lambda: ops.shade_bwd(a, r, m, n, l, d_out, spp, want_mat=True, want_light=True), 64),
    ("bwd<mat,n>", lambda: ops.shade_bwd(a, r, m, n, l, d_out, spp, want_mat=True, want_n=True), 76),
):
    us = timeit(f)
    print(f"{name:16s} {us:9.1f} us   {byt * px / us / 1e3:8.1f} GB/s algorithmic   {us * 1e3 / (px * spp):7.3f} ns/pixel-sample")
