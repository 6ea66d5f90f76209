"""hipEvent timing of the forward pass of the pos_mlp iteration alone (ArmMlpPhase.forward at 512 x 512): chain against layer by layer.
usage: chain_time.py [reps]   (MATPBR_LIB selects the library)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import posmlp, render, synthetic  # noqa: E402
from materialist_amd.armhead import ArmMlpPhase  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
H = W = 512
sc = synthetic.make_scene(0, H, W)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
scene = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True)
scene._set("emitter.data", t(sc.light))
gt = torch.rand(H, W, 3, device=dev)
a0, r0, m0 = (t(v).clamp(0, 1) for v in (sc.init_albedo, sc.init_roughness, sc.init_metallic))
start = torch.cat([a0.reshape(-1, 3), r0.reshape(-1, 1), m0.reshape(-1, 1)], -1).contiguous()
for chain in (True, False, True, False):
    ArmMlpPhase.FWD_CHAIN = chain
    torch.manual_seed(1)
    ph = ArmMlpPhase(scene, gt, posmlp.brdf_net("arm").to(dev), start, {"albedo": a0, "roughness": r0, "metallic": m0}, optimize_part="rm", spp=64)
    for _ in range(5):
        ph.forward()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ph.forward()
    e1.record()
    torch.cuda.synchronize()
    print(f"chain={chain}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per forward pass", flush=True)
    del ph
