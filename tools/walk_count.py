"""Re-sampled fraction of the 8 x 512 x 512 'rm' part over iterations 300-1300 (every 10th), per library (MATPBR_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from materialist_amd import loop, ops, render, synthetic
dev = torch.device("cuda:0")
B, H, W, spp = 8, 512, 512, 64
scs = [synthetic.make_scene(i, H, W) for i in range(B)]
t = lambda xs: torch.from_numpy(np.stack(xs)).to(dev)
scene = render.load_estimated_mesh(t([s.depth for s in scs]), use_mesh_normal=True)
scene._set("emitter.data", t([s.light for s in scs]))
with torch.no_grad():
    gt = render.render_w_brdf(scene, t([s.albedo for s in scs]), t([s.roughness for s in scs]), t([s.metallic for s in scs]), None, spp)
init = [t([getattr(s, k) for s in scs]) for k in ("init_albedo", "init_roughness", "init_metallic")]
ph = loop.FusedBrdfPhase(scene, gt, *init, optimize_part="rm", spp=spp)
ph.run(300)
fr = []
for it in range(100):
    ph.run(10)
    _, ref = ops.lazy_state_unpack(ph.lazy_state, ph.p["albedo"])
    fr.append(ref.float().mean(dim=(1, 2)).cpu().numpy())
fr = np.array(fr)
print(os.environ.get("MATPBR_LIB", "product"), "re-sampled fraction, iterations 300-1300: mean over images %.5f; per image" % fr.mean(), np.round(fr.mean(0), 5).tolist())
