import sys, numpy as np, torch
sys.path.insert(0, ".")
from materialist_amd import loop, ops, render, synthetic
dev = torch.device("cuda:0")
_t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
H = W = 128; spp = 64
sc = synthetic.make_scene(4, H, W)
scene = render.load_estimated_mesh(_t(sc.depth), use_mesh_normal=True)
scene._set("emitter.data", _t(sc.light))
with torch.no_grad():
    gt = render.render_w_brdf(scene, _t(sc.albedo), _t(sc.roughness), _t(sc.metallic), None, spp)
init = [_t(x) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
for att in (False, True):
    ph = loop.FusedBrdfPhase(scene, gt, *init, optimize_part="rm", spp=spp, lazy=True, keep_grads=True, attached_sampling=att)
    pa, pr, pm = (ph.p[k].clone() for k in ("albedo", "roughness", "metallic"))
    ph.step()
    got = ph.g["roughness"].clone()
    ex = loop.FusedBrdfPhase(scene, gt, *init, optimize_part="rm", spp=spp, lazy=False, keep_grads=True)
    ex.step()
    pred = ops.shade_fwd(pa, pr, pm, ph.n, ph.light, spp, clamp_params=True).requires_grad_(True)
    lo, _, _, _ = loop._loss.brdf_loss(pred, gt, {}, {}, 0.1, ph.gt_srgb)
    (d_pred,) = torch.autograd.grad(lo, pred)
    refs = {a: ops.shade_bwd(pa.clamp(0, 1), pr.clamp(0.07, 1), pm.clamp(0, 1), ph.n, ph.light, d_pred.contiguous(), spp, attached=a)[1] for a in (False, True)}
    n = lambda x: float(x.norm())
    print("att", att, "|got|", n(got), "|exact fused d_r|", n(ex.g["roughness"]), "|ref det|", n(refs[False]), "|ref att|", n(refs[True]))
    print("   got vs exact fused (det):", n(got - ex.g["roughness"]) / n(ex.g["roughness"]), " got vs ref det:", n(got - refs[False]) / n(refs[False]), " got vs ref att:", n(got - refs[True]) / n(refs[True]),
          " exact fused vs ref det:", n(ex.g["roughness"] - refs[False]) / n(refs[False]))
    e = (got - refs[att]).abs() / torch.maximum(refs[att].abs(), refs[att].abs().mean())
    print("   max scaled err", float(e.max()), "p99", float(e.flatten().quantile(0.99)), "median", float(e.flatten().median()))
