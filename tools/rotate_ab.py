"""None-mode lazy loop with SaveBest by rotation (MATPBR_FLAG_ROTATE_BEST) against the copying step, same process, alternating."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import loop, render, synthetic  # noqa: E402

dev = torch.device("cuda:0")
H = W = 512
spp = 64


def setup(B):
    scs = [synthetic.make_scene(i, H, W) for i in range(B)]
    t = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev)
    stack = lambda f: torch.stack([t(f(s)) for s in scs]) if B > 1 else t(f(scs[0]))
    scene = render.load_estimated_mesh(stack(lambda s: s.depth), use_mesh_normal=True)
    scene._set("emitter.data", stack(lambda s: s.light))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, stack(lambda s: s.albedo), stack(lambda s: s.roughness), stack(lambda s: s.metallic), None, spp)
    init = [stack(lambda s: s.init_albedo), stack(lambda s: s.init_roughness), stack(lambda s: s.init_metallic)]
    return scene, gt, init


for B in (8, 1):
    scene, gt, init = setup(B)
    for part in ("rm", "a"):
        res = {}
        for rnd in range(2):
            for rot in (True, False):
                ph = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp, rotate_best=rot, history_len=8)
                ph.run(300)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ph.run(1000)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                res.setdefault(rot, []).append(B * 1000 / dt)
        print(f"B={B} part={part}: rotate {[round(v) for v in res[True]]}  copy {[round(v) for v in res[False]]} image-it/s", flush=True)
