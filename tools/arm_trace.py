"""A part that moves all three maps (--opt_order arm, BASELINE configs[0]'s order) on the 8 x 512 x 512 shard, for rocprofv3 --kernel-trace --stats:
usage: arm_trace.py [groups = 2 | one] [part = arm]: one FusedBrdfPhase, or that many groups of images on streams of their own."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import loop, render, synthetic  # noqa: E402

dev = torch.device("cuda:0")
B, H, W, spp = 8, 512, 512, 64
part = sys.argv[2] if len(sys.argv) > 2 else "arm"
scs = [synthetic.make_scene(b, H, W) for b in range(B)]
st = lambda k: torch.from_numpy(np.stack([np.ascontiguousarray(getattr(s, k), dtype=np.float32) for s in scs])).to(dev)
scene = render.load_estimated_mesh(st("depth"), use_mesh_normal=True)
scene._set("emitter.data", st("light"))
with torch.no_grad():
    gt = render.render_w_brdf(scene, st("albedo"), st("roughness"), st("metallic"), None, spp).clone()
init = [st("init_albedo"), st("init_roughness"), st("init_metallic")]
import time

groups = 1 if len(sys.argv) > 1 and sys.argv[1] == "one" else int(sys.argv[1]) if len(sys.argv) > 1 else 2
ph = (loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp) if groups == 1 else
      loop.PipelinedBrdfPhase(scene, gt, *init, groups=groups, optimize_part=part, spp=spp))
ph.run(20)
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter()
    ph.run(300)
    torch.cuda.synchronize()                 # (the groups step on streams of their own: wall time around a full synchronisation)
    print(f"part {part!r}, {groups} group(s): {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per iteration of the shard", flush=True)
