"""A few launches of every 256-wide layer kernel of the pos_mlp iteration (round 5: two f16 pieces, three products; the bf16 and exact-f32
forms beside them) and of the forward chain, for rocprofv3 counter passes (tools/pmc_passes_r06.sh)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import ops, posmlp, render, synthetic  # noqa: E402
from materialist_amd.armhead import ArmMlpPhase  # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 512 * 512, 256, 256
x = torch.sin(torch.randn(M, 256, device=dev) * 3)
w = torch.randn(N, 256, device=dev) / 16
b = torch.randn(N, device=dev)
g = torch.randn(M, 256, device=dev) * 1e-6
s, c = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
gp, db = torch.empty(M, 256, device=dev), torch.empty(N, device=dev)
ws6, ws3, ws3t = ops.mlp_split_weights(w, N, K), ops.mlp_split_weights(w, N, K, f16=True), ops.mlp_split_weights(w, N, K, transposed=True, f16=True)
tmx = g.abs().view(M // 128, -1).amax(1).contiguous().view(torch.int32)
tmo = ops.mlp_tile_max(M, dev)
H = W = 512
sc = synthetic.make_scene(0, H, W)
t = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev)
scene = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True)
scene._set("emitter.data", t(sc.light))
a0, r0, m0 = (t(v).clamp(0, 1) for v in (sc.init_albedo, sc.init_roughness, sc.init_metallic))
start = torch.cat([a0.reshape(-1, 3), r0.reshape(-1, 1), m0.reshape(-1, 1)], -1).contiguous()
ph = ArmMlpPhase(scene, torch.rand(H, W, 3, device=dev), posmlp.brdf_net("arm").to(dev), start, {"albedo": a0, "roughness": r0, "metallic": m0}, optimize_part="rm", spp=64)
for _ in range(4):
    ops.mlp_layer_fwd_bx(x, ws3, b, s, None, N, K, 3)                            # mlp_nt_gx<sin, 3>
    ops.mlp_layer_bwd_input_blk(g, tmx, ws3t, s, gp, N, K, db, tmo)              # mlp_nt_gx<mul cos, 3>
    ops.mlp_layer_bwd_weight_blk(g, tmx, s, N, K)                                # mlp_wgrad_hx
    ph.forward()                                                                 # mlp_chain_fwd_kernel
    ops.mlp_layer_fwd_bx(x, ws6, b, s, None, N, K, 6)                            # round 4's forms
    ops.mlp_layer_bwd_input_bx(g, ws6, s, gp, N, K, db, 6, packed=True)
    ops.mlp_layer_bwd_weight_bx(g, x, N, K, 6)
    ops.mlp_layer_fwd(x, w, b, s, c, K)                                          # exact f32
torch.cuda.synchronize()
