"""Hot loop A alone (envmap PosMLP head + matpbr_env_phase_step, hipGraph replay) for rocprofv3 kernel traces.
usage: python tools/env_profile.py [steps] [graph|eager] [mlp|texels|envmlp|normal]
    mlp: loop.FusedEnvPhase (head, its backward and Adam as framework ops); texels: envhead.EnvTexelPhase; envmlp: envhead.EnvMlpPhase (the
    reference's parameterisation, every launch on the C ABI); normal: loop.NormalBrdfPhase (hot loop B, a part that moves the normal map)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import loop, posmlp, render, synthetic  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    graph = (sys.argv[2] != "eager") if len(sys.argv) > 2 else True
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(0, 512, 512)
    t = lambda x: torch.from_numpy(x).to(dev)
    s_env = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True)
    pr = render.traverse(s_env)
    pr["shape.bsdf.a"], pr["shape.bsdf.r"], pr["shape.bsdf.m"] = t(sc.albedo), t(sc.roughness), t(sc.metallic)
    s_gt = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True)
    s_gt._set("emitter.data", t(sc.light))
    with torch.no_grad():
        gt = render.render_w_brdf(s_gt, t(sc.albedo), t(sc.roughness), t(sc.metallic), None, 64)
    kind = sys.argv[3] if len(sys.argv) > 3 else "mlp"
    if kind == "normal":
        s_n = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=False)
        s_n._set("emitter.data", t(sc.light))
        geo = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True).shading_normal()
        gen = torch.Generator(device="cpu").manual_seed(1)
        n0 = torch.nn.functional.normalize(geo + 0.1 * torch.randn(geo.shape, generator=gen).to(dev), dim=-1).contiguous()
        ph = loop.NormalBrdfPhase(s_n, gt, t(sc.init_albedo), t(sc.init_roughness), t(sc.init_metallic), n0, optimize_part="armn", spp=64)
        enet = None
    elif kind == "envmlp":
        from materialist_amd.envhead import EnvMlpPhase
        ph = EnvMlpPhase(s_env, gt, posmlp.envmap_net().to(dev), torch.ones(512, 3, device=dev), spp=64, lr=1e-3, use_graph=graph)
        enet = None
    elif kind == "texels":
        from materialist_amd.envhead import EnvTexelPhase
        ph = EnvTexelPhase(s_env, gt, torch.zeros(16, 32, 3, device=dev, requires_grad=True), spp=64, lr=1e-3, use_graph=graph)
        enet = None
    else:
        ph = None
        enet = posmlp.envmap_net().to(dev)
    ones = torch.ones(512, 3, device=dev)
    ph = ph or loop.FusedEnvPhase(s_env, gt, lambda: enet(ones).reshape(16, 32, 3), loop.capturable_adam(enet.parameters(), 1e-3), spp=64,
                            use_graph=graph, keep_pred=False)
    for _ in range(10):
        ph.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ph.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"env iteration: {el / steps * 1e6:.1f} us ({steps / el:.0f} it/s), graph={graph}, mse {float(ph.stats[0, 1]):.5f}")


if __name__ == "__main__":
    main()
