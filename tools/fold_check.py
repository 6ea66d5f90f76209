import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '.')
from materialist_amd import loop, ops, render, synthetic
dev = torch.device('cuda')
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
def setup(H, W, spp, image_id=0):
    sc = synthetic.make_scene(image_id, H, W)
    scene = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True)
    scene._set("emitter.data", t(sc.light))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, t(sc.albedo), t(sc.roughness), t(sc.metallic), None, spp)
    init = [t(x) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    return scene, gt, init
for (H, W) in ((96, 131), (256, 256)):
    scene, gt, init = setup(H, W, 64, 3)
    for part in ("rm", "a", "r", "m"):
        iters = 150
        fo = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=64, history_len=iters, fold=True, patience=50, min_delta=1e-3)
        ge = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=64, history_len=iters, fold=False, patience=50, min_delta=1e-3)
        assert fo.fold and not ge.fold
        nref_f = nref_g = 0
        for it in range(iters):
            fo.step(); ge.step()
            if it in (0, 1, 2, 10, 149):
                dp = float((fo.pred - ge.pred).abs().max() / ge.pred.abs().mean())
                print(part, (H, W), "it", it, "pred rel diff", dp, "mse", float(fo.stats[0, ops.STAT_MSE]), float(ge.stats[0, ops.STAT_MSE]))
            _, rf = ops.lazy_state_unpack(fo.lazy_state, fo.p["albedo"]); _, rg = ops.lazy_state_unpack(ge.lazy_state, ge.p["albedo"])
            nref_f += int(rf.sum()); nref_g += int(rg.sum())
        hf, hg = fo.history()[:, 0].cpu().numpy(), ge.history()[:, 0].cpu().numpy()
        print(part, (H, W), "hist max rel diff", np.abs(hf - hg).max() / hg.max(), "resampled", nref_f, nref_g)
        for k in ("albedo", "roughness", "metallic"):
            d = (fo.p[k] - ge.p[k]).abs()
            print("   ", k, "max", float(d.max()), "mean", float(d.mean()), "frac<5e-5", float((d < 5e-5).float().mean()))
            db = (fo.best[k] - ge.best[k]).abs()
            print("    best", k, float(db.max()))
        print("    best_img", float((fo.best_img - ge.best_img).abs().max()), "stats", (fo.stats - ge.stats).abs().max().item())
print("OK")
