import sys
sys.path.insert(0, '.')
import numpy as np, torch
from materialist_amd import loop, ops, render, synthetic
from materialist_amd import loss as _loss
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
H = W = 64; spp = 64
sc = synthetic.make_scene(5, H, W)
scene = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True)
scene._set("emitter.data", t(sc.light))
with torch.no_grad():
    gt = render.render_w_brdf(scene, t(sc.albedo), t(sc.roughness), t(sc.metallic), None, spp).clone()
a0, r0, m0 = t(sc.init_albedo), t(sc.init_roughness), t(sc.init_metallic)
for part in ("a", "rm", "arm"):
    a = (a0 + 0.05 * torch.randn_like(a0)).requires_grad_(True)
    r = (r0 + 0.05 * torch.randn_like(r0)).requires_grad_(True)
    m = (m0 + 0.05 * torch.randn_like(m0)).requires_grad_(True)
    ac, rc, mc = a.clamp(0, 1), r.clamp(0.07, 1), m.clamp(0, 1)
    pred = render.render_w_brdf(scene, ac, rc, mc, None, spp)
    keys = {"a": "albedo", "r": "roughness", "m": "metallic"}
    parts = {keys[c]: {"albedo": ac, "roughness": rc, "metallic": mc}[keys[c]] for c in part}
    total, mse, _, _ = _loss.brdf_loss(pred, gt, parts, {"albedo": a0, "roughness": r0, "metallic": m0}, 0.1)
    total.backward()
    # fused path
    n, light = scene.shading_normal().contiguous(), scene.light.detach().contiguous()
    dcache = ops.diffuse_cache(n, light, spp)
    jac = ops.plane9(a0)
    pred2 = ops.shade_fwd(a.detach(), r.detach(), m.detach(), n, light, spp, clamp_params=True, dcache=dcache, jac=jac)
    stats = ops.new_loss_stats(1, dev)
    gt_srgb = _loss.linear_to_srgb(gt).contiguous()
    ops.brdf_loss_stats(pred2, gt, gt_srgb, a.detach(), r.detach(), m.detach(), a0, r0, m0, 0.1, stats, None, optimize_part=part)
    g = [torch.empty_like(x) for x in (a0, r0, m0)]
    ops.brdf_loss_bwd_jac(a.detach(), r.detach(), m.detach(), jac, pred2, gt_srgb, stats, a0, r0, m0, 0.1, *g, optimize_part=part)
    print(part, "loss", float(total), float(stats[0, ops.STAT_LOSS]), "pred diff", float((pred2 - pred).abs().max()))
    for nm, x, y in (("a", a.grad, g[0]), ("r", r.grad, g[1]), ("m", m.grad, g[2])):
        if x is None: x = torch.zeros_like(y)
        print("   d_%s: |ref| %.3e |fused| %.3e  rel diff %.3e" % (nm, float(x.norm()), float(y.norm()), float((x - y).norm() / (x.norm() + 1e-30))))
