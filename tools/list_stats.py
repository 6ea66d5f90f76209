"""How bursty are the lists of pixels that leave their models' intervals?  Per 512-pixel block and per 64-pixel wave segment, over iterations 100..400 of an 8 x 512^2 'rm' part."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from materialist_amd import loop, ops, render, synthetic
dev = torch.device('cuda')
B, H, W = 8, 512, 512
scs = [synthetic.make_scene(i, H, W) for i in range(B)]
st = lambda k: torch.from_numpy(np.stack([getattr(s, k) for s in scs])).to(dev)
scene = render.load_estimated_mesh(st("depth"), use_mesh_normal=True)
scene._set("emitter.data", st("light"))
with torch.no_grad():
    gt = render.render_w_brdf(scene, st("albedo"), st("roughness"), st("metallic"), None, 64)
ph = loop.FusedBrdfPhase(scene, gt, st("init_albedo"), st("init_roughness"), st("init_metallic"), optimize_part="rm", spp=64, fold=False)
ph.run(100)
blk, seg, wg = [], [], []
for it in range(300):
    ph.step()
    if it % 10 == 0:
        _, ref = ops.lazy_state_unpack(ph.lazy_state, ph.p["albedo"])
        r = ref.reshape(B, -1).cpu().numpy()
        blk.append(r.reshape(B, -1, 512).sum(-1).ravel())
        seg.append(r.reshape(B, -1, 64).sum(-1).ravel())
        wg.append(r.reshape(B, -1, 2048).sum(-1).ravel())
for name, x in (("512-px block", np.concatenate(blk)), ("64-px segment", np.concatenate(seg)), ("2048-px workgroup", np.concatenate(wg))):
    print(name, "mean %.3f" % x.mean(), "max", x.max(), "pcts 50/90/99/99.9", [float(np.percentile(x, p)) for p in (50, 90, 99, 99.9)],
          "frac>8 %.4f >16 %.4f >32 %.4f" % ((x > 8).mean(), (x > 16).mean(), (x > 32).mean()))
per_it = np.stack(wg).max(1); print("max per 2048-px workgroup per iteration: mean %.1f max %d" % (per_it.mean(), per_it.max()))
per_it = np.stack(seg).max(1); print("max per 64-px segment per iteration: mean %.1f max %d" % (per_it.mean(), per_it.max()))
per_it = np.stack(blk).max(1); print("max per 512-px block per iteration: mean %.1f max %d" % (per_it.mean(), per_it.max()))
