"""8 x 512x512 none-mode loop: one FusedBrdfPhase over the batch against the batch cut into groups that step on streams of their own (the walk and
statistics launches of one group -- latency-bound, a handful of waves -- run under the streaming step of another).  usage: python tools/pipeline_ab.py [part]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import loop, render, synthetic  # noqa: E402


def main():
    part = sys.argv[1] if len(sys.argv) > 1 else "rm"
    dev = torch.device("cuda:0")
    B, H, W = 8, 512, 512
    scs = [synthetic.make_scene(i, H, W) for i in range(B)]
    t = lambda f, sel: torch.stack([torch.as_tensor(f(scs[i]), dtype=torch.float32) for i in sel]).to(dev)

    def phase(sel):
        scene = render.load_estimated_mesh(t(lambda s: s.depth, sel), use_mesh_normal=True)
        scene._set("emitter.data", t(lambda s: s.light, sel))
        with torch.no_grad():
            gt = render.render_w_brdf(scene, t(lambda s: s.albedo, sel), t(lambda s: s.roughness, sel), t(lambda s: s.metallic, sel), None, 64)
        return loop.FusedBrdfPhase(scene, gt, t(lambda s: s.init_albedo, sel), t(lambda s: s.init_roughness, sel), t(lambda s: s.init_metallic, sel),
                                   optimize_part=part, spp=64)

    # the product's form of it: loop.PipelinedBrdfPhase (two groups, the step on 512 workgroups: MATPBR_FLAG_SHARE_GPU)
    sel = range(B)
    scene = render.load_estimated_mesh(t(lambda s: s.depth, sel), use_mesh_normal=True)
    scene._set("emitter.data", t(lambda s: s.light, sel))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, t(lambda s: s.albedo, sel), t(lambda s: s.roughness, sel), t(lambda s: s.metallic, sel), None, 64)
    for g2 in (2, 4):
        pp = loop.PipelinedBrdfPhase(scene, gt, t(lambda s: s.init_albedo, sel), t(lambda s: s.init_roughness, sel), t(lambda s: s.init_metallic, sel),
                                     groups=g2, optimize_part=part, spp=64)
        pp.run(300)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pp.run(500)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"part {part} PipelinedBrdfPhase groups {g2}: {el / 500 * 1e6:.1f} us per iteration of 8 images = {500 * B / el:.0f} image-iterations/s, "
              f"mse {float(pp.stats[:, 1].mean()):.6f}")
        del pp
    for groups in (1, 2):
        per = B // groups
        phs = [phase(range(g * per, (g + 1) * per)) for g in range(groups)]
        streams = [torch.cuda.Stream(dev) for _ in range(groups)]
        torch.cuda.synchronize()

        def run(n):
            for _ in range(n):
                for ph, st in zip(phs, streams):
                    with torch.cuda.stream(st):
                        ph.step()
        run(300)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(500)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"part {part} groups {groups}: {el / 500 * 1e6:.1f} us per iteration of 8 images = {500 * B / el:.0f} image-iterations/s, "
              f"mse {float(torch.cat([p.stats[:, 1] for p in phs]).mean()):.6f}")


main()
