"""Host-side spherical-harmonic utilities (numpy, float64): basis, envmap<->SH linear maps, y-rotation.

Convention = the reference's only SH definition, myutils/computeSH.py:13-68 (order 4, 25 real
coefficients, index l(l+1)+m, Condon-Shortley sign inside P_l^m), with directions mapped to angles as
myutils/envmap_utils.py:29-36 does for envmap lookups: theta = acos(y), phi = atan2(x, -z).
These are small one-off linear maps (25 x 512); the per-pixel SH work is in the HIP kernels.
"""
from __future__ import annotations

import math

import numpy as np

NSH = 25
SH_L = np.array([l for l in range(5) for _ in range(-l, l + 1)])
SH_M = np.array([m for l in range(5) for m in range(-l, l + 1)])


def sh_norm() -> np.ndarray:
    """N_k with Y_k = N_k * B_k(X,Y,Z); B_k are the raw polynomials of `sh_poly`."""
    K = lambda l, m: math.sqrt((2 * l + 1) * math.factorial(l - m) / math.factorial(l + m) / (4 * math.pi))
    s2 = math.sqrt(2.0)
    return np.array([
        K(0, 0),
        -s2 * K(1, 1), K(1, 0), -s2 * K(1, 1),
        6 * s2 * K(2, 2), -3 * s2 * K(2, 1), 0.5 * K(2, 0), -3 * s2 * K(2, 1), 3 * s2 * K(2, 2),
        -15 * s2 * K(3, 3), 30 * s2 * K(3, 2), -1.5 * s2 * K(3, 1), 0.5 * K(3, 0), -1.5 * s2 * K(3, 1), 15 * s2 * K(3, 2), -15 * s2 * K(3, 3),
        420 * s2 * K(4, 4), -105 * s2 * K(4, 3), 15 * s2 * K(4, 2), -2.5 * s2 * K(4, 1), 0.125 * K(4, 0), -2.5 * s2 * K(4, 1),
        7.5 * s2 * K(4, 2), -105 * s2 * K(4, 3), 105 * s2 * K(4, 4),
    ])


def sh_poly(w: np.ndarray) -> np.ndarray:
    """Raw basis polynomials B_k of world directions w[...,3] -> [...,25]; (X,Y,Z) = (-z, x, y)."""
    w = np.asarray(w, dtype=np.float64)
    X, Y, Z = -w[..., 2], w[..., 0], w[..., 1]
    z2, xy, yz, xz = Z * Z, X * Y, Y * Z, X * Z
    d = X * X - Y * Y
    t5, t7 = 5 * z2 - 1, 7 * z2 - 1
    s3, c3 = Y * (3 * X * X - Y * Y), X * (X * X - 3 * Y * Y)
    return np.stack([
        np.ones_like(X), Y, Z, X, xy, yz, 3 * z2 - 1, xz, d,
        s3, xy * Z, Y * t5, Z * (t5 - 2), X * t5, d * Z, c3,
        xy * d, s3 * Z, xy * t7, yz * (t7 - 2), (35 * z2 - 30) * z2 + 3, xz * (t7 - 2), d * t7, c3 * Z, d * d - 4 * xy * xy,
    ], axis=-1)


def sh_basis(w: np.ndarray) -> np.ndarray:
    return sh_poly(w) * sh_norm()


def envmap_directions(He: int, We: int) -> np.ndarray:
    """Unit direction of every texel centre of an equirectangular map, [He,We,3].
    Inverse of lookup_envmap (myutils/envmap_utils.py:29-36): col = (atan2(x,-z)/2pi * W) mod W,
    row = acos(y)/pi * H, evaluated at (col+.5, row+.5)."""
    theta = (np.arange(He) + 0.5) / He * math.pi
    phi = (np.arange(We) + 0.5) / We * 2 * math.pi
    T, P = np.meshgrid(theta, phi, indexing="ij")
    return np.stack([np.sin(T) * np.sin(P), np.cos(T), -np.sin(T) * np.cos(P)], -1)


def envmap_solid_angles(He: int, We: int) -> np.ndarray:
    edges = np.cos(np.arange(He + 1) / He * math.pi)
    return np.repeat((edges[:-1] - edges[1:])[:, None], We, 1) * (2 * math.pi / We)


def envmap_to_sh_matrix(He: int = 16, We: int = 32) -> np.ndarray:
    """P [25, He*We] with coef = P @ env.reshape(He*We, 3): quadrature of the SH projection integral."""
    Y = sh_basis(envmap_directions(He, We)).reshape(He * We, NSH)
    return (Y * envmap_solid_angles(He, We).reshape(-1, 1)).T.copy()


def sh_to_envmap_matrix(He: int = 16, We: int = 32) -> np.ndarray:
    """R [He*We, 25] with env = R @ coef: radiance of the SH light at every texel centre."""
    return sh_basis(envmap_directions(He, We)).reshape(He * We, NSH).copy()


def rotate_y_matrix(angle_rad: float) -> np.ndarray:
    """[25,25] matrix M with coef' = M @ coef for the light rotated by `angle_rad` about the +y (pole) axis,
    i.e. L'(theta, phi) = L(theta, phi - angle): per-band 2x2 rotations of the (m, -m) pairs.
    Counterpart of rotate_envmap's column roll (render_final.py:290-298)."""
    M = np.zeros((NSH, NSH))
    for l in range(5):
        M[l * (l + 1), l * (l + 1)] = 1.0
        for m in range(1, l + 1):
            c, s = math.cos(m * angle_rad), math.sin(m * angle_rad)
            ip, im = l * (l + 1) + m, l * (l + 1) - m
            # Y_m ~ cos(m phi), Y_-m ~ sin(m phi); cos(m(phi-a)) = c cos + s sin, sin(m(phi-a)) = c sin - s cos
            M[ip, ip], M[ip, im] = c, -s
            M[im, ip], M[im, im] = s, c
    return M
