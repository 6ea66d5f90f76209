#!/usr/bin/env python3
"""Record how the roughness of individual pixels moves during hot loop B (`--model_name none`, part 'rm'): the input of the
lazy re-sampling study (tools/lazy_sim.py).  Writes gpurun_out/r_traj_<scene>.npz with r[T, n] (clamped roughness of n pixels on
a regular sub-grid at every iteration), m[T, n], a[T, n, 3], their pixel indices and the per-iteration mse.

    python tools/record_r_traj.py [--iters 800] [--stride 8] [--scene synthetic|indoor2]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=800)
    ap.add_argument("--stride", type=int, default=8)
    ap.add_argument("--scene", default="synthetic")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    args = ap.parse_args()
    import torch

    from materialist_amd import loop, loss as _loss, render, synthetic

    dev = torch.device("cuda", 0)
    H = W = 512
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
    if args.scene == "synthetic":
        sc = synthetic.make_scene(0, H, W)
        scene = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True)
        scene._set("emitter.data", t(sc.light))
        with torch.no_grad():
            gt = render.render_w_brdf(scene, t(sc.albedo), t(sc.roughness), t(sc.metallic), None, 64)
        init = (t(sc.init_albedo), t(sc.init_roughness), t(sc.init_metallic))
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "indoor2.npz"))
        gt = _loss.srgb_to_linear(t(z["image_srgb_u8"].astype(np.float32) / 255))
        depth = z["depth_pred_f32"].astype(np.float32)
        depth = 2 * depth.max() - depth                                     # inverse_img_w_mi.py:713
        scene = render.load_estimated_mesh(t(depth), use_mesh_normal=True)
        scene._set("emitter.data", torch.ones(16, 32, 3, device=dev))       # :322 the first BRDF phase renders under ones
        init = (t(z["albedo_pred_f16"].astype(np.float32)).clamp(0, 1), torch.full((H, W, 1), 0.7, device=dev), torch.full((H, W, 1), 0.05, device=dev))
    ph = loop.FusedBrdfPhase(scene, gt, *init, optimize_part="rm", spp=64, patience=0)
    idx = (torch.arange(0, H, args.stride, device=dev)[:, None] * W + torch.arange(0, W, args.stride, device=dev)[None, :]).reshape(-1)
    T = args.iters
    rr = torch.empty((T, idx.numel()), device=dev)
    mm = torch.empty_like(rr)
    for it in range(T):
        rr[it] = ph.p["roughness"].reshape(-1)[idx].clamp(0.07, 1)
        mm[it] = ph.p["metallic"].reshape(-1)[idx].clamp(0, 1)
        ph.step()
    torch.cuda.synchronize()
    os.makedirs(args.out, exist_ok=True)
    nrm = scene.shading_normal().reshape(-1, 3)[idx].cpu().numpy()
    light = scene.light.detach().cpu().numpy()
    np.savez_compressed(os.path.join(args.out, f"r_traj_{args.scene}.npz"), r=rr.cpu().numpy(), m=mm.cpu().numpy(), idx=idx.cpu().numpy(),
                        a=ph.p["albedo"].reshape(-1, 3)[idx].clamp(0, 1).cpu().numpy(), n=nrm, light=light, mse=ph.history().cpu().numpy()[:, 0])
    d = (rr[1:] - rr[:-1]).abs()
    print(args.scene, "mean |dr| per iteration by 100-iteration block:", [round(float(d[k:k + 100].mean()), 6) for k in range(0, T - 1, 100)])
    print("max |dr|:", float(d.max()), " total path p50/p99:", np.percentile(d.sum(0).cpu().numpy(), [50, 99]))


if __name__ == "__main__":
    main()
