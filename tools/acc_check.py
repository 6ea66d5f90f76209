"""In-step statistics (round 6) against the statistics launch: the folded step (statistics of iteration t + 1 formed where its render is formed) against
the generic step (loss_sums2 launch on the render) on the same part: MSE, L1, ratio per iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from materialist_amd import loop, ops, render, synthetic
dev = torch.device("cuda:0")
H, W, spp = 96, 131, 64
sc = synthetic.make_scene(3, H, W)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
for masked in (False, True):
    mask = None
    if masked:
        mask = torch.zeros(H, W, dtype=torch.bool); mask[:17] = True
    def scene_():
        s = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True, mesh_mask=mask)
        s._set("emitter.data", t(sc.light))
        return s
    with torch.no_grad():
        gt = render.render_w_brdf(scene_(), t(sc.albedo), t(sc.roughness), t(sc.metallic), None, spp)
    init = [t(sc.init_albedo), t(sc.init_roughness), t(sc.init_metallic)]
    for part in ("rm", "a"):
        fo = loop.FusedBrdfPhase(scene_(), gt, *init, optimize_part=part, spp=spp, fold=True)
        ge = loop.FusedBrdfPhase(scene_(), gt, *init, optimize_part=part, spp=spp, fold=False)
        out = []
        for it in range(12):
            fo.step(); ge.step()
            a, b = fo.stats[0].cpu().numpy(), ge.stats[0].cpu().numpy()
            out.append((it + 1, abs(a[0] / b[0] - 1), abs(a[1] / b[1] - 1), abs(a[2] / b[2] - 1), abs(a[3] / b[3] - 1), float((fo.p["roughness"] - ge.p["roughness"]).abs().max())))
        print(f"masked={masked} part={part}: iteration, |ratio|, |mse|, |l1|, |sr| relative differences, max |r| difference")
        for o in out:
            print("   %2d  %.1e  %.1e  %.1e  %.1e   %.1e" % o)
