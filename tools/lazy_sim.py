#!/usr/bin/env python3
"""Lazy re-sampling on recorded roughness trajectories (CPU, fp64 oracle = the specification of the HIP kernels).

Replays r[t], m[t] of the pixels that tools/record_r_traj.py sampled from a real run of hot loop B through
oracle.lazy_fwd_lanes and compares every iteration with the exact render: worst |lazy - exact| / max(|exact|, mean|exact|)
over all pixels and iterations (the parity bar is 1e-3) and the fraction of pixels whose 20 GGX samples are walked again.

    python tools/lazy_sim.py gpurun_out/r_traj_synthetic.npz [--tol 1.0] [--iters 800]
    python tools/lazy_sim.py --model            # no recording: Adam-like drift model
"""
import argparse
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle.oracle import Oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("traj", nargs="?")
    ap.add_argument("--model", action="store_true")
    ap.add_argument("--tol", type=float, default=1.0)
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--size", type=int, default=512)
    args = ap.parse_args()
    o = Oracle(np.float64)
    H = W = args.size
    if args.model:
        from materialist_amd import synthetic

        sc = synthetic.make_scene(0, H, W)
        rng = np.random.default_rng(3)
        N, T = 4096, args.iters or 600
        idx = rng.integers(0, H * W, N)
        n = o.normals_from_depth(sc.depth.astype(np.float64)).reshape(-1, 3)[idx]
        a = sc.albedo.reshape(-1, 3)[idx].astype(np.float64)
        light = sc.light.astype(np.float64)
        # Adam-like motion: per-pixel drift direction with momentum-filtered noise, step <= lr(t)
        bias = rng.uniform(-1, 1, N)
        s = np.zeros(N)
        r = np.empty((T, N))
        m = np.empty((T, N))
        r[0], m[0] = 0.7, 0.05
        lr = 3e-4
        for t in range(1, T):
            if t % 100 == 0 and lr > 1.5e-4:
                lr *= 0.8
            s = 0.9 * s + 0.1 * (bias * np.exp(-t / 300) + rng.normal(0, 1.0, N))
            r[t] = np.clip(r[t - 1] + lr * np.tanh(2 * s), 0.07, 1)
            m[t] = np.clip(m[t - 1] + lr * np.tanh(2 * s[::-1]), 0, 1)
    else:
        z = np.load(args.traj)
        r, m, idx = z["r"].astype(np.float64), z["m"].astype(np.float64), z["idx"]
        a, n, light = z["a"].astype(np.float64), z["n"].astype(np.float64), z["light"].astype(np.float64)
        if light.ndim == 3 and light.shape[0] == 16:      # texel envmap: not recorded as SH -> cannot replay
            raise SystemExit("trajectory holds a texel light; record with SH light")
        T = min(args.iters or r.shape[0], r.shape[0])
        N = r.shape[1]
    wo = np.stack([o.view_dir(int(i // W), int(i % W), H, W) for i in idx])
    state = np.zeros((N, o.lazy_nstate()))
    worst, nref = [], []
    floor = None
    dr_err = []
    for t in range(T):
        ex = o.shade_fwd_lanes(a, r[t], m[t], n, wo, light, 64)
        if floor is None:
            floor = 0.5 * np.abs(ex).mean()
        lz, jac, ref = o.lazy_fwd_lanes(a, r[t], m[t], n, wo, light, state, 64, floor, tol=args.tol, force=(t == 0))
        e = np.abs(lz - ex) / np.maximum(np.abs(ex), np.abs(ex).mean())
        worst.append(e.max())
        nref.append(ref.mean())
        if t % 50 == 0:
            # d out / d r held at the reference point vs the exact detached derivative
            g = np.ones_like(ex)
            d_ex = o.shade_bwd_lanes(a, r[t], m[t], n, wo, light, g, 64)[1]
            d_lz = jac[:, 6:9].sum(1)
            dr_err.append(float(np.linalg.norm(d_lz - d_ex) / np.linalg.norm(d_ex)))
            print(f"t={t:4d} worst rel err {e.max():.2e} (p99.9 {np.percentile(e, 99.9):.1e})  refreshed {ref.mean():.3f}  "
                  f"median lo/hi {np.median(state[:, 1]):.4f}/{np.median(state[:, 2]):.4f} rho {np.median(state[:, 3]):.4f}  d_r rel-L2 err {dr_err[-1]:.1e}", flush=True)
    worst, nref = np.array(worst), np.array(nref)
    print(f"tol x{args.tol}: worst over {T} iterations {worst.max():.2e}; mean refresh fraction (t>=1) {nref[1:].mean():.4f}; "
          f"by 100-iteration block: {[round(float(nref[max(k, 1):k + 100].mean()), 4) for k in range(0, T, 100)]}")


if __name__ == "__main__":
    main()
