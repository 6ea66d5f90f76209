import sys, numpy as np, torch
sys.path.insert(0, ".")
from materialist_amd import loop, ops, render, synthetic
dev = torch.device("cuda:0")
_t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
H, W, spp, B = 64, 96, 16, 2
scs = [synthetic.make_scene(10 + i, H, W) for i in range(B)]
st = lambda k: _t(np.stack([getattr(s, k) for s in scs]))
scene_b = render.load_estimated_mesh(st("depth"), use_mesh_normal=True)
scene_b._set("emitter.data", st("light"))
with torch.no_grad():
    gt = render.render_w_brdf(scene_b, st("albedo"), st("roughness"), st("metallic"), None, spp)
init = [st(k) for k in ("init_albedo", "init_roughness", "init_metallic")]
def single(b):
    s1 = render.load_estimated_mesh(st("depth")[b], use_mesh_normal=True)
    s1._set("emitter.data", st("light")[b])
    return loop.FusedBrdfPhase(s1, gt[b], *[x[b] for x in init], optimize_part="rm", spp=spp, lazy=True)
fb = loop.FusedBrdfPhase(scene_b, gt, *init, optimize_part="rm", spp=spp, lazy=True)
f0, f0b = single(0), single(0)
for it in range(30):
    fb.step(); f0.step(); f0b.step()
    eq_ss = torch.equal(f0.p["roughness"], f0b.p["roughness"])
    eq_bs = torch.equal(fb.p["roughness"][0], f0.p["roughness"])
    eq_pred = torch.equal(fb.pred[0], f0.pred)
    eq_stats = torch.equal(fb.stats[0], f0.stats[0])
    _, rb = ops.lazy_state_unpack(fb.lazy_state, fb.p["albedo"]); _, r1 = ops.lazy_state_unpack(f0.lazy_state, f0.p["albedo"])
    print(it, "single==single", eq_ss, "batch==single r", eq_bs, "pred", eq_pred, "stats", eq_stats, "refreshed", int(rb[0].sum()), int(r1.sum()),
          (fb.stats[0]-f0.stats[0]).abs().max().item(), "dcache eq", torch.equal(fb.dcache[:,0], f0.dcache[:,0]))
    if not eq_bs and it > 3: break
