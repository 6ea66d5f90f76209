"""How far can the specular sums of a pixel be extrapolated in the roughness before the render leaves 1e-3?

Study behind the lazy re-sampling of hot loop B: out(r) of the production estimator on random lanes of a
synthetic scene (fp64 oracle), against
  T1: first-order Taylor around r0 with the true slope (central difference, h = 1e-3)
  Q3: the parabola through r0 - D, r0, r0 + D evaluated inside [r0 - D, r0 + D]
Error metric = the GPU parity bar: |approx - exact| / max(|exact|, mean|exact|), worst lane and percentiles.

    python tools/lazy_study.py [n_lanes]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from materialist_amd import synthetic  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    o = Oracle(np.float64)
    H = W = 512
    sc = synthetic.make_scene(0, H, W)
    rng = np.random.default_rng(7)
    ii, jj = rng.integers(0, H, N), rng.integers(0, W, N)
    n_img = o.normals_from_depth(sc.depth.astype(np.float64))
    n = n_img[ii, jj]
    wo = np.stack([o.view_dir(int(i), int(j), H, W) for i, j in zip(ii, jj)])
    a = sc.albedo[ii, jj].astype(np.float64)
    m = sc.metallic[ii, jj, 0].astype(np.float64)
    m[: N // 3] = 1.0   # a third of the lanes are full metals: the render is the specular lobe alone
    # tilt a third of the normals strongly (grazing views)
    t = rng.normal(size=(N, 3)) * 0.6
    t[: 2 * N // 3] *= 0.15
    n = n + t
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    light = sc.light.astype(np.float64)
    spp = 64

    def f(r):
        return o.shade_fwd_lanes(a, r, m, n, wo, light, spp)

    h = 1e-3
    for r0v in (0.08, 0.12, 0.2, 0.35, 0.5, 0.7, 0.9):
        r0 = np.full(N, r0v) + rng.uniform(-0.005, 0.005, N)
        f0 = f(r0)
        slope = (f(r0 + h) - f(r0 - h)) / (2 * h)
        scale = np.maximum(np.abs(f0), np.abs(f0).mean())
        print(f"r0 ~ {r0v}: mean out {np.abs(f0).mean():.3f}; relative slope p50/p99/max "
              + "/".join(f"{v:.2f}" for v in np.percentile(np.abs(slope) / scale, [50, 99, 100])))
        for d in (0.002, 0.004, 0.008, 0.016, 0.03):
            errs = []
            for sgn in (-1, 1):
                rr = np.clip(r0 + sgn * d, 0.07, 1.0)
                ex = f(rr)
                t1 = f0 + slope * (rr - r0)[:, None]
                errs.append(np.abs(t1 - ex) / np.maximum(np.abs(ex), np.abs(ex).mean()))
            e = np.maximum(*errs).max(axis=1)
            # parabola through r0-D, r0, r0+D with D = d, checked at +-D/2 (about where a parabola's interpolation error peaks)
            fm, fp = f(np.clip(r0 - d, 0.07, 1)), f(np.clip(r0 + d, 0.07, 1))
            eq = []
            for x in (-0.5, 0.5):
                rr = r0 + x * d
                ex = f(rr)
                q = f0 + (fp - fm) / 2 * x + (fp - 2 * f0 + fm) / 2 * x * x
                ok = (r0 - d >= 0.07) & (r0 + d <= 1.0)
                eq.append(np.where(ok[:, None], np.abs(q - ex) / np.maximum(np.abs(ex), np.abs(ex).mean()), 0.0))
            eq = np.maximum(*eq).max(axis=1)
            print(f"   d = {d:5.3f}  T1 err p50 {np.percentile(e, 50):.1e} p99 {np.percentile(e, 99):.1e} max {e.max():.1e}"
                  f"  frac>1e-3 {np.mean(e > 1e-3):.4f} | Q3(+-D/2) p99 {np.percentile(eq, 99):.1e} max {eq.max():.1e}")


if __name__ == "__main__":
    main()
