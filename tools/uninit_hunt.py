"""Does the lazy none-mode loop read memory it has not written?  The same phase twice, the allocator's free blocks poisoned in between;
then again with single buffers cleared after construction, to find which one matters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import loop, render, synthetic  # noqa: E402

dev = torch.device("cuda:0")
H = W = 256
spp = 64
sc = synthetic.make_scene(3, H, W)
t = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev)


def poison(val):
    xs = [torch.full((64 << 20,), val, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    del xs


def run(clear=(), part="rm", iters=400, mask=False, val=float("nan")):
    poison(val)
    scene = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True)
    scene._set("emitter.data", t(sc.light))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, t(sc.albedo), t(sc.roughness), t(sc.metallic), None, spp)
    if mask:
        m = torch.zeros(H, W, dtype=torch.bool, device=dev)
        m[:40, :60] = True
        scene.set_mesh_mask(~m)
    ph = loop.FusedBrdfPhase(scene, gt, t(sc.init_albedo), t(sc.init_roughness), t(sc.init_metallic), optimize_part=part, spp=spp, patience=50, min_delta=1e-3)
    for name in clear:
        getattr(ph, name).zero_()
    ph.run(iters)
    torch.cuda.synchronize()
    return [ph.p[k].clone() for k in ("albedo", "roughness", "metallic")] + [ph.stats.clone(), ph.best_img.clone()]


def same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


for mask in (False, True):
    for part in ("rm", "a"):
        ref = run(part=part, mask=mask, val=0.0)
        for val in (float("nan"), 1e30, -3.0):
            r = run(part=part, mask=mask, val=val)
            print(f"mask={mask} part={part} poison={val}: {'same' if same(ref, r) else 'DIFFERENT'}", flush=True)
            if not same(ref, r):
                for c in ("ws", "lazy_fold", "jac", "_pred", "_best_img", "hist", "dcache"):
                    r2 = run(clear=(c,), part=part, mask=mask, val=val)
                    print("   cleared", c, "->", "same" if same(ref, r2) else "different", flush=True)
                break
