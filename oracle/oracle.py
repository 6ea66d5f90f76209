"""ctypes loader for the CPU oracle (oracle/matpbr_oracle.c).                       TEST INFRASTRUCTURE

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  The product
package (materialist_amd/) must never import it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
NSH = 25


def build(force: bool = False) -> None:
    """Compile both oracle libraries with gcc (idempotent)."""
    src = os.path.join(_HERE, "matpbr_oracle.c")
    libs = [os.path.join(_BUILD, f"liboracle_{t}.so") for t in ("f64", "f32")]
    stale = force or any((not os.path.exists(l)) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)


class Oracle:
    """Thin typed wrapper.  dtype float64 = the checker, float32 (+OpenMP) = the timed CPU port."""

    def __init__(self, dtype=np.float64):
        build()
        self.dtype = np.dtype(dtype)
        tag = "f64" if self.dtype == np.float64 else "f32"
        path = os.path.join(_BUILD, f"liboracle_{tag}.so")
        try:
            self.lib = ctypes.CDLL(path)
        except OSError:
            build(force=True)  # e.g. built on another host
            self.lib = ctypes.CDLL(path)
        self.creal = ctypes.c_double if tag == "f64" else ctypes.c_float
        assert self.lib.oracle_sizeof_real() == self.dtype.itemsize
        for name in ("oracle_D_GGX", "oracle_G1_GGX_Schlick", "oracle_fresnelSchlick"):
            fn = getattr(self.lib, name)
            fn.restype = self.creal
            fn.argtypes = [self.creal, self.creal]
        self.lib.oracle_G_Smith.restype = self.creal
        self.lib.oracle_G_Smith.argtypes = [self.creal] * 3
        self.lib.oracle_sh_K.restype = self.creal
        self.lib.oracle_sh_K.argtypes = [ctypes.c_int, ctypes.c_int]

    # -- helpers ------------------------------------------------------------------------------
    def _a(self, x):
        return np.ascontiguousarray(x, dtype=self.dtype)

    @staticmethod
    def _p(x):
        return None if x is None else x.ctypes.data_as(ctypes.c_void_p)

    def _r(self, x):
        return self.creal(float(x))

    # -- scalar functions (vectorised by a Python loop; grids are small) -----------------------
    def D_GGX(self, cos_h, eta):
        return np.vectorize(lambda c, e: self.lib.oracle_D_GGX(c, e))(cos_h, eta)

    def G1(self, nov, eta):
        return np.vectorize(lambda c, e: self.lib.oracle_G1_GGX_Schlick(c, e))(nov, eta)

    def G_Smith(self, nov, nol, eta):
        return np.vectorize(lambda v, l, e: self.lib.oracle_G_Smith(v, l, e))(nov, nol, eta)

    def fresnel(self, voh, f0):
        return np.vectorize(lambda v, f: self.lib.oracle_fresnelSchlick(v, f))(voh, f0)

    def sh_K(self):
        return np.array([self.lib.oracle_sh_K(l, m) for l in range(5) for m in range(-l, l + 1)])

    # -- batch functions: AoS [N,3] ------------------------------------------------------------
    def eval_brdf(self, wi, wo, n, a, r, m):
        wi, wo, n, a, r, m = map(self._a, (wi, wo, n, a, r, m))
        N = r.shape[0]
        f = np.empty((N, 3), self.dtype)
        pdf = np.empty(N, self.dtype)
        self.lib.oracle_eval_brdf_batch(ctypes.c_long(N), *map(self._p, (wi, wo, n, a, r, m, f, pdf)))
        return f, pdf

    def eval_brdf_grad(self, wi, wo, n, a, r, m, g):
        wi, wo, n, a, r, m, g = map(self._a, (wi, wo, n, a, r, m, g))
        N = r.shape[0]
        d_a = np.empty((N, 3), self.dtype)
        d_n = np.empty((N, 3), self.dtype)
        d_r = np.empty(N, self.dtype)
        d_m = np.empty(N, self.dtype)
        self.lib.oracle_eval_brdf_grad_batch(ctypes.c_long(N), *map(self._p, (wi, wo, n, a, r, m, g, d_a, d_r, d_m, d_n)))
        return d_a, d_r, d_m, d_n

    def sample_brdf(self, sample1, sample2, wo, n, a, r, m):
        sample1, sample2, wo, n, a, r, m = map(self._a, (sample1, sample2, wo, n, a, r, m))
        N = r.shape[0]
        wi = np.empty((N, 3), self.dtype)
        w = np.empty((N, 3), self.dtype)
        pdf = np.empty(N, self.dtype)
        self.lib.oracle_sample_brdf_batch(ctypes.c_long(N), *map(self._p, (sample1, sample2, wo, n, a, r, m, wi, pdf, w)))
        return wi, pdf, w

    def diffuse_sampler(self, u0, u1, n):
        n = self._a(n)
        wi = np.empty(3, self.dtype)
        self.lib.oracle_diffuse_sampler(self._r(u0), self._r(u1), self._p(n), self._p(wi))
        return wi

    def specular_sampler(self, u0, u1, rough, wo, n):
        n, wo = self._a(n), self._a(wo)
        wi = np.empty(3, self.dtype)
        self.lib.oracle_specular_sampler(self._r(u0), self._r(u1), self._r(rough), self._p(wo), self._p(n), self._p(wi))
        return wi

    def world_to_screen(self, p, fov_rad, aspect, near, far, width, height):
        p = self._a(p)
        out = np.empty(2, self.dtype)
        self.lib.oracle_world_to_screen(self._p(p), self._r(fov_rad), self._r(aspect), self._r(near), self._r(far),
                                        ctypes.c_int(width), ctypes.c_int(height), self._p(out))
        return out

    def pixel_to_world(self, i, j, depth, H, W, fov_deg=35.0):
        out = np.empty(3, self.dtype)
        self.lib.oracle_pixel_to_world(ctypes.c_int(i), ctypes.c_int(j), self._r(depth), ctypes.c_int(H), ctypes.c_int(W),
                                       self._r(fov_deg), self._p(out))
        return out

    def view_dir(self, i, j, H, W, fov_deg=35.0):
        out = np.empty(3, self.dtype)
        self.lib.oracle_view_dir(ctypes.c_int(i), ctypes.c_int(j), ctypes.c_int(H), ctypes.c_int(W), self._r(fov_deg), self._p(out))
        return out

    def sh_basis_angles(self, theta, phi):
        theta, phi = self._a(theta), self._a(phi)
        Y = np.empty((theta.shape[0], NSH), self.dtype)
        self.lib.oracle_sh_basis_batch(ctypes.c_long(theta.shape[0]), self._p(theta), self._p(phi), self._p(Y))
        return Y

    def sh_basis_dir(self, w):
        w = self._a(w)
        Y = np.empty((w.shape[0], NSH), self.dtype)
        self.lib.oracle_sh_basis_dir_batch(ctypes.c_long(w.shape[0]), self._p(w), self._p(Y))
        return Y

    def sample_table(self, spp):
        u = np.empty((spp // 2, 2), self.dtype)
        self.lib.oracle_sample_table(ctypes.c_int(spp), self._p(u))
        return u

    # -- image level ---------------------------------------------------------------------------
    # kind: 0 = the production estimator, 1 = the reference-literal MIS estimator (yardstick)
    def shade_fwd(self, a, r, m, n, light, spp, fov_deg=35.0, kind=0):
        a, r, m, n, light = map(self._a, (a, r, m, n, light))
        B, H, W = self._bhw(a)
        out = np.empty_like(a)
        self.lib.oracle_shade_fwd_kind(*map(self._p, (a, r, m, n, light, out)), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(B),
                                       ctypes.c_int(spp), self._r(fov_deg), ctypes.c_int(kind))
        return out

    def shade_fwd_win(self, a, r, m, n, light, spp, H, W, i0, j0, fov_deg=35.0, kind=0):
        """Render of the h x w window at (i0, j0) of an H x W image (maps are the window's)."""
        a, r, m, n, light = map(self._a, (a, r, m, n, light))
        B, h, w = self._bhw(a)
        out = np.empty_like(a)
        self.lib.oracle_shade_fwd_win(*map(self._p, (a, r, m, n, light, out)), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(B),
                                      ctypes.c_int(spp), self._r(fov_deg), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(i0),
                                      ctypes.c_int(j0), ctypes.c_int(kind))
        return out

    def shade_fwd_lanes(self, a, r, m, n, wo, light, spp, kind=0):
        """N lanes with explicit view directions wo[N,3] under one light[25,3]."""
        a, r, m, n, wo, light = map(self._a, (a, r, m, n, wo, light))
        out = np.empty_like(a)
        self.lib.oracle_shade_fwd_lanes(*map(self._p, (a, r, m, n, wo, light, out)), ctypes.c_long(a.shape[0]), ctypes.c_int(spp),
                                        ctypes.c_int(kind))
        return out

    def shade_bwd_lanes(self, a, r, m, n, wo, light, d_out, spp, kind=0):
        a, r, m, n, wo, light, d_out = map(self._a, (a, r, m, n, wo, light, d_out))
        d_a, d_n = np.empty_like(a), np.empty_like(n)
        d_r, d_m = np.empty_like(r), np.empty_like(m)
        d_l = np.empty_like(light)
        self.lib.oracle_shade_bwd_lanes(*map(self._p, (a, r, m, n, wo, light, d_out, d_a, d_r, d_m, d_n, d_l)), ctypes.c_long(a.shape[0]),
                                        ctypes.c_int(spp), ctypes.c_int(kind))
        return d_a, d_r, d_m, d_n, d_l

    def shade_fwd_frozen(self, a, r, m, n, n_s, r_s, light, spp, fov_deg=35.0):
        """Forward with sample directions / pdf frozen at (n_s, r_s): the function whose gradient shade_bwd returns."""
        a, r, m, n, n_s, r_s, light = map(self._a, (a, r, m, n, n_s, r_s, light))
        B, H, W = self._bhw(a)
        out = np.empty_like(a)
        self.lib.oracle_shade_fwd_frozen(*map(self._p, (a, r, m, n, n_s, r_s, light, out)), ctypes.c_int(H), ctypes.c_int(W),
                                         ctypes.c_int(B), ctypes.c_int(spp), self._r(fov_deg))
        return out

    def shade_bwd(self, a, r, m, n, light, d_out, spp, fov_deg=35.0, want_n=True, want_light=True, kind=0):
        a, r, m, n, light, d_out = map(self._a, (a, r, m, n, light, d_out))
        B, H, W = self._bhw(a)
        d_a = np.empty_like(a)
        d_r = np.empty_like(r)
        d_m = np.empty_like(m)
        d_n = np.empty_like(n) if want_n else None
        d_l = np.empty_like(light) if want_light else None
        self.lib.oracle_shade_bwd_kind(*map(self._p, (a, r, m, n, light, d_out, d_a, d_r, d_m, d_n, d_l)), ctypes.c_int(H),
                                       ctypes.c_int(W), ctypes.c_int(B), ctypes.c_int(spp), self._r(fov_deg), ctypes.c_int(kind))
        return d_a, d_r, d_m, d_n, d_l

    def diffuse_cache(self, n, light, spp, fov_deg=35.0):
        """A0, A1, A2 (rgb each) of the diffuse lobe per pixel: [..., 9]."""
        n, light = self._a(n), self._a(light)
        B, H, W = self._bhw(n)
        out = np.empty(n.shape[:-1] + (9,), self.dtype)
        self.lib.oracle_diffuse_cache(self._p(n), self._p(light), self._p(out), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(B),
                                      ctypes.c_int(spp), self._r(fov_deg))
        return out

    def shade_fwd_cached(self, a, r, m, n, light, dcache, spp, fov_deg=35.0):
        a, r, m, n, light, dcache = map(self._a, (a, r, m, n, light, dcache))
        B, H, W = self._bhw(a)
        out = np.empty_like(a)
        self.lib.oracle_shade_fwd_cached(*map(self._p, (a, r, m, n, light, dcache, out)), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(B),
                                         ctypes.c_int(spp), self._r(fov_deg))
        return out

    # -- lazy re-sampling (specification of csrc/matpbr_lazy.hpp) ---------------------------------
    def lazy_nstate(self):
        return int(self.lib.oracle_lazy_nstate())

    def lazy_fwd(self, a, r, m, n, light, state, spp, floor, tol=1.0, force=False, fov_deg=35.0):
        """One lazy forward of an image batch.  `state` [..., NSTATE] is updated IN PLACE (must be a contiguous array of this
        oracle's dtype); returns (out [...,3], jac [...,9], refreshed [...] int32)."""
        a, r, m, n, light = map(self._a, (a, r, m, n, light))
        B, H, W = self._bhw(a)
        assert state.dtype == self.dtype and state.flags.c_contiguous and state.shape == a.shape[:-1] + (self.lazy_nstate(),)
        out = np.empty_like(a)
        jac = np.empty(a.shape[:-1] + (9,), self.dtype)
        ref = np.empty(a.shape[:-1], np.int32)
        fl = self._a(np.broadcast_to(np.asarray(floor, dtype=self.dtype), (B,)))
        self.lib.oracle_lazy_fwd(*map(self._p, (a, r, m, n, light, state, out, jac, ref)), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(B),
                                 ctypes.c_int(spp), self._r(fov_deg), self._p(fl), self._r(tol), ctypes.c_int(1 if force else 0))
        return out, jac, ref

    def lazy_fwd_lanes(self, a, r, m, n, wo, light, state, spp, floor, tol=1.0, force=False):
        a, r, m, n, wo, light = map(self._a, (a, r, m, n, wo, light))
        N = a.shape[0]
        assert state.dtype == self.dtype and state.flags.c_contiguous and state.shape == (N, self.lazy_nstate())
        out = np.empty_like(a)
        jac = np.empty((N, 9), self.dtype)
        ref = np.empty(N, np.int32)
        self.lib.oracle_lazy_fwd_lanes(*map(self._p, (a, r, m, n, wo, light, state, out, jac, ref)), ctypes.c_long(N), ctypes.c_int(spp),
                                       self._r(floor), self._r(tol), ctypes.c_int(1 if force else 0))
        return out, jac, ref

    def shade_transfer(self, a, r, m, n, spp, fov_deg=35.0):
        """Per-pixel radiance transfer T[..., 25, 3]: render = sum_k light[k] * T[k]."""
        a, r, m, n = map(self._a, (a, r, m, n))
        B, H, W = self._bhw(a)
        T = np.empty(a.shape[:-1] + (NSH, 3), self.dtype)
        self.lib.oracle_shade_transfer(*map(self._p, (a, r, m, n, T)), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(B), ctypes.c_int(spp),
                                       self._r(fov_deg))
        return T

    def rule(self, spp, lobe):
        """(u0, u1, w) of the production quadrature rule of lobe 0 (diffuse) / 1 (specular)."""
        nmax = self.lib.oracle_rule_max()
        u0, u1, w = (np.empty(nmax, self.dtype) for _ in range(3))
        n = self.lib.oracle_rule(ctypes.c_int(spp), ctypes.c_int(lobe), self._p(u0), self._p(u1), self._p(w))
        return u0[:n].copy(), u1[:n].copy(), w[:n].copy()

    def normals_from_depth(self, depth, fov_deg=35.0):
        depth = self._a(depth)
        if depth.ndim == 2:
            B, (H, W) = 1, depth.shape
        else:
            B, H, W = depth.shape
        out = np.empty(depth.shape + (3,), self.dtype)
        self.lib.oracle_normals_from_depth(self._p(depth), self._p(out), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(B), self._r(fov_deg))
        return out

    @staticmethod
    def _bhw(a):
        if a.ndim == 3:
            return 1, a.shape[0], a.shape[1]
        return a.shape[0], a.shape[1], a.shape[2]
