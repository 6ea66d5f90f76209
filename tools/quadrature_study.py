"""Quadrature study for the split estimator: closed-form-in-r diffuse lobe + GGX-sampled
specular lobe, against the reference-literal MIS estimator (oracle, spp 4096).  Host-only (numpy + the fp64 oracle).

    python tools/quadrature_study.py            # prints the PSNR table used in DESIGN.md
"""
from __future__ import annotations

import ctypes
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from materialist_amd import sh as _sh  # noqa: E402
from materialist_amd import synthetic  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def frame(n):
    sign = np.where(n[..., 2] >= 0, 1.0, -1.0)
    a = -1.0 / (sign + n[..., 2])
    b = n[..., 0] * n[..., 1] * a
    s = np.stack([1 + sign * n[..., 0] ** 2 * a, sign * b, -sign * n[..., 0]], -1)
    t = np.stack([b, sign + n[..., 1] ** 2 * a, -n[..., 1]], -1)
    return s, t


def vdc2(i):
    i = np.asarray(i, dtype=np.uint64)
    r = np.zeros(i.shape)
    f = 0.5
    while (i > 0).any():
        r += f * (i & 1)
        i >>= 1
        f *= 0.5
    return r


def rule_hammersley(n):
    m = 1
    while m < n:
        m <<= 1
    i = np.arange(n)
    return (i + 0.5) / n, vdc2(i) + 0.5 / m, np.full(n, 1.0 / n)


def rule_product(nu, nphi, gauss=True, twist=True):
    """nu points in u0 (Gauss-Legendre on [0,1] or midpoints) x nphi equally spaced azimuths (ring k rotated)."""
    if gauss:
        x, w = np.polynomial.legendre.leggauss(nu)
        u = 0.5 * (x + 1)
        wu = 0.5 * w
    else:
        u = (np.arange(nu) + 0.5) / nu
        wu = np.full(nu, 1.0 / nu)
    U0, U1, Wt = [], [], []
    for k in range(nu):
        off = (vdc2(np.array([k]))[0] if twist else 0.0) / nphi
        for j in range(nphi):
            U0.append(u[k]); U1.append((j + 0.5) / nphi + off); Wt.append(wu[k] / nphi)
    return np.array(U0), np.array(U1) % 1.0, np.array(Wt)


def sh_radiance(wi, light):
    return _sh.sh_basis(wi) @ light  # [...,3]


def split_estimator(a, r, m, n, wo, light, drule, srule):
    """a[N,3] r[N] m[N] n[N,3] unit, wo[N,3]; returns rgb[N,3]."""
    s, t = frame(n)
    NoV = np.maximum((n * wo).sum(-1), 0.0)
    po = (1 - NoV) ** 5
    # diffuse moments
    M = [np.zeros_like(a) for _ in range(5)]
    for u0, u1, w in zip(*drule):
        st, ct, ph = math.sqrt(u0), math.sqrt(1 - u0), 2 * math.pi * u1
        wi = s * (st * math.cos(ph)) + t * (st * math.sin(ph)) + n * ct
        L = sh_radiance(wi, light)
        u = 1 + (wi * wo).sum(-1, keepdims=True)
        p5 = (1 - ct) ** 5
        M[0] += w * L; M[1] += w * p5 * L; M[2] += w * u * L; M[3] += w * u * p5 * L; M[4] += w * u * u * p5 * L
    po_ = po[:, None]
    A0 = M[0] * (1 - 0.5 * po_) + M[1] * (0.25 * po_ - 0.5)
    A1 = po_ * M[2] + (1 - po_) * M[3]
    A2 = po_ * M[4]
    r_ = r[:, None]
    diff = a * (1 - m[:, None]) * (A0 + r_ * A1 + r_ * r_ * A2)
    # specular
    alpha2 = r ** 4
    k = (r + 1) ** 2 / 8
    g1v = 1 / (NoV * (1 - k) + k + 1e-6)
    vx, vy, vz = (s * wo).sum(-1), (t * wo).sum(-1), (n * wo).sum(-1)
    S0 = np.zeros_like(a); S1 = np.zeros_like(a)
    for u0, u1, w in zip(*srule):
        q = 1 / (1 + u0 * (alpha2 - 1))
        ct = np.sqrt(np.maximum((1 - u0) * q, 0)); st = np.sqrt(np.maximum(alpha2 * u0 * q, 0))
        ph = 2 * math.pi * u1
        hx, hy = st * math.cos(ph), st * math.sin(ph)
        d = hx * vx + hy * vy + ct * vz
        wlx, wly, wlz = 2 * d * hx - vx, 2 * d * hy - vy, 2 * d * ct - vz
        wi = s * wlx[:, None] + t * wly[:, None] + n * wlz[:, None]
        NoL = np.maximum(wlz, 0)
        g1l = 1 / (NoL * (1 - k) + k + 1e-6)
        ok = (d > 0) & (wlz > 0)
        wgt = np.where(ok, w * g1l * g1v * NoL * np.abs(d) / np.maximum(ct, 1e-20), 0.0)
        L = sh_radiance(wi, light)
        x5 = (1 - np.abs(d)) ** 5
        S0 += wgt[:, None] * L
        S1 += (wgt * x5)[:, None] * L
    C0 = 0.04 * (1 - m[:, None]) + m[:, None] * a
    return diff + C0 * S0 + (1 - C0) * S1


def psnr(x, ref):
    g = lambda v: np.clip(v, 0, None) ** (1 / 2.2)
    return -10 * math.log10(np.mean((g(x) - g(ref)) ** 2))


def lanes_case(kind, oracle):
    sc = synthetic.make_scene(0, 512, 512)
    nrm = oracle.normals_from_depth(sc.depth.astype(np.float64))
    ii, jj = np.meshgrid(np.arange(16, 512, 10)[:48], np.arange(16, 512, 10)[:48], indexing="ij")
    ii, jj = ii.ravel(), jj.ravel()
    a = sc.albedo[ii, jj].astype(np.float64)
    r = sc.roughness[ii, jj, 0].astype(np.float64)
    m = sc.metallic[ii, jj, 0].astype(np.float64)
    n = nrm[ii, jj]
    wo = np.stack([oracle.view_dir(int(i), int(j), 512, 512) for i, j in zip(ii, jj)])
    if kind == "stress":
        r = np.full_like(r, 0.1); m = np.ones_like(m)
    if kind == "r007":
        r = np.full_like(r, 0.07); m = np.ones_like(m)
    if kind == "tilt":   # strongly tilted normals (grazing views)
        rng = np.random.default_rng(5)
        n = n + rng.normal(size=n.shape) * 0.8
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        n = np.where(((n * wo).sum(-1) > 0.05)[:, None], n, nrm[ii, jj])
    return a, r, m, n, wo, sc.light.astype(np.float64)


def oracle_lanes(oracle, a, r, m, n, wo, light, spp, kind=1):
    """kind 1 = the reference-literal MIS estimator, 0 = the production estimator of the oracle"""
    return oracle.shade_fwd_lanes(a, r, m, n, wo, light, spp, kind=kind)


def main():
    oracle = Oracle(np.float64)
    cases = {k: lanes_case(k, oracle) for k in ("scene", "stress", "r007", "tilt")}
    refs = {k: oracle_lanes(oracle, *v, 4096) for k, v in cases.items()}
    refs2 = {k: oracle_lanes(oracle, *v, 8192) for k, v in cases.items()}
    print("reference self-consistency (spp 4096 vs 8192):", {k: round(psnr(refs[k], refs2[k]), 1) for k in cases})
    print("old MIS estimator:")
    for spp in (16, 32, 64, 128):
        print(f"  spp {spp:4d}", {k: round(psnr(oracle_lanes(oracle, *cases[k], spp), refs[k]), 1) for k in cases})
    drules = {"ham32": rule_hammersley(32), "ham64": rule_hammersley(64), "ham256": rule_hammersley(256), "gl4x8": rule_product(4, 8), "gl6x12": rule_product(6, 12)}
    srules = {"ham8": rule_hammersley(8), "ham16": rule_hammersley(16), "ham32": rule_hammersley(32), "ham64": rule_hammersley(64),
              "gl2x4": rule_product(2, 4), "gl2x6": rule_product(2, 6), "gl3x4": rule_product(3, 4), "gl3x5": rule_product(3, 5), "gl3x6": rule_product(3, 6), "gl4x4": rule_product(4, 4), "gl3x8": rule_product(3, 8),
              "gl4x6": rule_product(4, 6), "gl4x8": rule_product(4, 8), "gl5x8": rule_product(5, 8), "gl6x10": rule_product(6, 10),
              "mid4x8": rule_product(4, 8, gauss=False), "gl4x8nt": rule_product(4, 8, twist=False)}
    print("split estimator, diffuse rule sweep (specular ham64):")
    for dn, dr in drules.items():
        print(f"  {dn:8s}", {k: round(psnr(split_estimator(*cases[k], dr, srules['ham64']), refs[k]), 1) for k in cases})
    print("split estimator, specular rule sweep (diffuse ham256):")
    for sn, sr in srules.items():
        print(f"  {sn:8s} n={len(sr[0]):3d}", {k: round(psnr(split_estimator(*cases[k], drules['ham256'], sr), refs[k]), 1) for k in cases})


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def rule_product_t(nu, nphi, power=2.0, twist=True):
    """Gauss-Legendre in v with u0 = 1 - (1-v)^power: clusters nodes towards u0 -> 1 (grazing half vectors)."""
    x, w = np.polynomial.legendre.leggauss(nu)
    v = 0.5 * (x + 1); wv = 0.5 * w
    u = 1 - (1 - v) ** power
    wu = wv * power * (1 - v) ** (power - 1)
    U0, U1, Wt = [], [], []
    for k in range(nu):
        off = (vdc2(np.array([k]))[0] if twist else 0.0) / nphi
        for j in range(nphi):
            U0.append(u[k]); U1.append((j + 0.5) / nphi + off); Wt.append(wu[k] / nphi)
    return np.array(U0), np.array(U1) % 1.0, np.array(Wt)


def study2():
    oracle = Oracle(np.float64)
    cases = {k: lanes_case(k, oracle) for k in ("scene", "stress", "tilt")}
    refs = {k: oracle_lanes(oracle, *v, 4096) for k, v in cases.items()}
    d256 = rule_hammersley(256)
    s64 = rule_product(6, 10)
    print("diffuse rules (specular gl6x10):")
    for name, dr in {"gl2x4": rule_product(2, 4), "gl2x6": rule_product(2, 6), "gl3x4": rule_product(3, 4), "gl3x6": rule_product(3, 6), "gl3x8": rule_product(3, 8),
                     "gl4x6": rule_product(4, 6), "gl4x8": rule_product(4, 8), "ham16": rule_hammersley(16), "ham32": rule_hammersley(32)}.items():
        print(f"  {name:8s} n={len(dr[0]):3d}", {k: round(psnr(split_estimator(*cases[k], dr, s64), refs[k]), 1) for k in cases})
    print("specular rules with u0 = 1-(1-v)^p (diffuse ham256):")
    for p in (1.0, 1.5, 2.0, 3.0):
        for nu, nphi in ((3, 4), (4, 4), (4, 6), (5, 6)):
            sr = rule_product_t(nu, nphi, p)
            print(f"  p={p} {nu}x{nphi}", {k: round(psnr(split_estimator(*cases[k], d256, sr), refs[k]), 1) for k in cases})


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "2":
    study2()


def study3():
    """The production estimator as the oracle defines it (rules from spp), values and gradients, against MIS at spp 4096."""
    oracle = Oracle(np.float64)
    cases = {k: lanes_case(k, oracle) for k in ("scene", "stress", "r007", "tilt")}
    refs = {k: oracle_lanes(oracle, *v, 4096) for k, v in cases.items()}
    print("production estimator vs MIS(4096), PSNR dB:")
    for spp in (8, 16, 32, 64, 128):
        print(f"  spp {spp:4d} rules d/s = {len(oracle.rule(spp, 0)[0])}/{len(oracle.rule(spp, 1)[0])}",
              {k: round(psnr(oracle_lanes(oracle, *cases[k], spp, kind=0), refs[k]), 1) for k in cases},
              " MIS same spp:", {k: round(psnr(oracle_lanes(oracle, *cases[k], spp, kind=1), refs[k]), 1) for k in cases})
    rng = np.random.default_rng(0)
    print("gradients (d_out ~ N(0,1)), relative L2 error against MIS(8192) gradients:")
    for k, v in cases.items():
        g = rng.normal(size=v[0].shape)
        ref = oracle.shade_bwd_lanes(*v, g, 8192, kind=1)
        ref2 = oracle.shade_bwd_lanes(*v, g, 4096, kind=1)
        for spp in (64,):
            new = oracle.shade_bwd_lanes(*v, g, spp, kind=0)
            old = oracle.shade_bwd_lanes(*v, g, spp, kind=1)
            rel = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))
            names = ("d_a", "d_r", "d_m", "d_n", "d_light")
            print(f"  {k:7s} spp {spp}: new", {nm: f"{rel(x, y):.3g}" for nm, x, y in zip(names, new, ref)},
                  "\n                  MIS", {nm: f"{rel(x, y):.3g}" for nm, x, y in zip(names, old, ref)},
                  "\n            MIS(4096)", {nm: f"{rel(x, y):.3g}" for nm, x, y in zip(names, ref2, ref)})


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "3":
    study3()


def study4():
    """Specular rule shape (nu x nphi) against value and gradient accuracy."""
    oracle = Oracle(np.float64)
    cases = {k: lanes_case(k, oracle) for k in ("scene", "stress", "tilt")}
    rng = np.random.default_rng(0)
    gs = {k: rng.normal(size=v[0].shape) for k, v in cases.items()}
    refs = {k: oracle_lanes(oracle, *v, 4096) for k, v in cases.items()}
    grefs = {k: oracle.shade_bwd_lanes(*v, gs[k], 8192, kind=1) for k, v in cases.items()}
    rel = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))
    for nu, nphi in ((4, 4), (5, 3), (6, 3), (5, 4), (6, 4), (8, 2), (8, 3), (4, 6), (6, 6), (8, 4), (8, 8)):
        oracle.lib.oracle_rule_override(1, nu, nphi)
        row = {}
        for k, v in cases.items():
            val = psnr(oracle_lanes(oracle, *v, 64, kind=0), refs[k])
            g = oracle.shade_bwd_lanes(*v, gs[k], 64, kind=0)
            row[k] = f"{val:.1f}dB d_r {rel(g[1], grefs[k][1]):.3f} d_n {rel(g[3], grefs[k][3]):.3f} d_l {rel(g[4], grefs[k][4]):.3f}"
        print(f"  spec {nu}x{nphi} n={nu*nphi:3d}", row)
    oracle.lib.oracle_rule_override(1, 0, 0)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "4":
    study4()
