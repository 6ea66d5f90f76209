"""Seeded synthetic scenes: depth, ground-truth material maps, SH light and an
initial guess.  Pure numpy/scipy on the host; nothing here needs the reference tree."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
from scipy.ndimage import gaussian_filter

from . import sh as _sh


@dataclass
class SyntheticScene:
    depth: np.ndarray       # [H,W]
    albedo: np.ndarray      # [H,W,3]
    roughness: np.ndarray   # [H,W,1]
    metallic: np.ndarray    # [H,W,1]
    light: np.ndarray       # [25,3] SH coefficients
    init_albedo: np.ndarray
    init_roughness: np.ndarray
    init_metallic: np.ndarray


def _lowpass(rng, H, W, C, sigma, lo, hi):
    x = rng.random((H, W, C))
    x = gaussian_filter(x, sigma=(sigma, sigma, 0), mode="wrap")
    x = (x - x.min()) / max(x.max() - x.min(), 1e-12)
    return lo + (hi - lo) * x


def make_light(rng) -> np.ndarray:
    coef = np.zeros((_sh.NSH, 3))
    coef[0] = math.sqrt(4 * math.pi)  # unit mean radiance
    for k in range(1, _sh.NSH):
        coef[k] = rng.normal(0.0, 0.3 / (_sh.SH_L[k] + 1) ** 2, 3) * math.sqrt(4 * math.pi)
    R = _sh.sh_to_envmap_matrix(32, 64)
    for _ in range(60):  # shrink the l>=1 bands until the reconstructed radiance is >= 0.02 everywhere
        if (R @ coef).min() >= 0.02:
            break
        coef[1:] *= 0.9
    return coef


def make_scene(image_id: int = 0, H: int = 512, W: int = 512) -> SyntheticScene:
    rng = np.random.default_rng(1234 + int(image_id))
    s = W / 512.0
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = np.full((H, W), 2.0)
    for _ in range(6):
        A = rng.uniform(-0.4, 0.4)
        sig = rng.uniform(40, 120) * s
        ci, cj = rng.uniform(0, H), rng.uniform(0, W)
        depth += A * np.exp(-((ii - ci) ** 2 + (jj - cj) ** 2) / (2 * sig * sig))
    albedo = _lowpass(rng, H, W, 3, 8 * s, 0.05, 0.95)
    rough = _lowpass(rng, H, W, 1, 8 * s, 0.07, 1.0)
    met = (_lowpass(rng, H, W, 1, 8 * s, 0.0, 1.0) > 0.6).astype(np.float64)
    met = gaussian_filter(met, sigma=(2 * s, 2 * s, 0), mode="wrap")
    light = make_light(rng)
    noise = gaussian_filter(rng.normal(0, 0.1, (H, W, 3)), sigma=(4 * s, 4 * s, 0), mode="wrap") * 4
    init_a = np.clip(albedo + noise, 0.0, 1.0)
    # the reference restarts roughness/metallic from constants (inverse_img_w_mi.py:183-188)
    init_r = np.full((H, W, 1), 0.7)
    init_m = np.full((H, W, 1), 0.05)
    f32 = lambda x: np.ascontiguousarray(x, dtype=np.float32)
    return SyntheticScene(f32(depth), f32(albedo), f32(rough), f32(met), f32(light), f32(init_a), f32(init_r), f32(init_m))
