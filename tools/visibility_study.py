"""How much of the residual between the build's direct-lighting render and Mitsuba's path-traced render of the reference's
optimised maps (indoor sample) is self-occlusion by the depth mesh?  Screen-space ray marching of the depth map, numpy only."""
import math, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from materialist_amd import sh
from materialist_amd.imageio_exr import read_exr
from materialist_amd.imageio_hdr import read_hdr
from oracle.oracle import Oracle

REF = "/root/reference/output_imgs/indoor"
g = lambda x: np.clip(x, 0, 1) ** (1 / 2.2)
psnr = lambda a, b: -10 * np.log10(np.mean((g(a) - g(b)) ** 2))
o = Oracle(np.float32)
a, r, m = (read_exr(f"{REF}/best_results/{k}.exr") for k in ("albedo", "roughness", "metallic"))
ref = read_exr(f"{REF}/best_results/rendered_img.exr")
depth = read_exr(f"{REF}/depthPred.exr")[..., 0]
depth = 2 * depth.max() - depth
n = o.normals_from_depth(depth)
env = read_hdr(f"{REF}/best_results/envmap.hdr")
coef = (sh.envmap_to_sh_matrix(16, 32) @ env.reshape(512, 3).astype(np.float64)).astype(np.float32)
img = o.shade_fwd(a, r, m, n, coef, 64)
print("baseline: raw %.2f dB, mean-matched %.2f dB" % (psnr(img, ref), psnr(img * ref.mean() / img.mean(), ref)))

H, W = depth.shape
f = 0.5 * W / math.tan(0.5 * math.radians(35.0))
cx, cy = 0.5 * (W - 1), 0.5 * (H - 1)
jj, ii = np.meshgrid(np.arange(W), np.arange(H))
# camera looks down -z; pixel ray p = ((j-cx)/f, -(i-cy)/f, -1) * depth (depth = z distance)
P = np.stack([(jj - cx) / f * depth, -(ii - cy) / f * depth, -depth], -1).astype(np.float32)

def visibility(dirs, steps=48, t_max=None, thick=None):
    """dirs [K,3] world directions -> V [K,H,W] in {0,1}: 1 if the ray from the surface point along dir leaves the depth mesh unoccluded."""
    t_max = t_max or 0.6 * float(depth.mean())
    thick = thick or 0.05 * float(depth.mean())
    V = np.ones((len(dirs), H, W), np.float32)
    ts = (np.arange(1, steps + 1) / steps) ** 1.5 * t_max
    for k, d in enumerate(dirs):
        occ = np.zeros((H, W), bool)
        for t in ts:
            q = P + 0.02 * t_max * n + t * d                      # offset along the normal against self-hits
            z = -q[..., 2]
            ok = z > 1e-3
            u = np.where(ok, q[..., 0] / np.maximum(z, 1e-3) * f + cx, -1)
            v = np.where(ok, -q[..., 1] / np.maximum(z, 1e-3) * f + cy, -1)
            ui, vi = np.round(u).astype(np.int64), np.round(v).astype(np.int64)
            inside = ok & (ui >= 0) & (ui < W) & (vi >= 0) & (vi < H)
            zs = np.where(inside, depth[np.clip(vi, 0, H - 1), np.clip(ui, 0, W - 1)], np.inf)
            hit = inside & (z > zs + 1e-3 * t_max) & (z < zs + thick + 2 * t * 0.1)
            occ |= hit
        V[k] = ~occ
    return V

# cosine-weighted irradiance with and without visibility, under the SH light, at every pixel: K directions on the sphere
K = 96
rng = np.random.default_rng(0)
# Fibonacci sphere
idx = np.arange(K) + 0.5
phi = np.pi * (1 + 5 ** 0.5) * idx
zc = 1 - 2 * idx / K
dirs = np.stack([np.sqrt(1 - zc ** 2) * np.cos(phi), zc, np.sqrt(1 - zc ** 2) * np.sin(phi)], -1).astype(np.float32)
L = np.maximum(sh.sh_basis(dirs) @ coef.astype(np.float64), 0.0).astype(np.float32)          # [K,3]
t0 = time.time()
V = visibility(dirs)
print("visibility: %.1f s, mean visible fraction over the upper hemisphere:" % (time.time() - t0), end=" ")
cosw = np.maximum(np.einsum("hwc,kc->khw", n, dirs), 0.0)                                     # [K,H,W]
print(float((V * cosw).sum() / cosw.sum()))
E0 = np.einsum("khw,kc->hwc", cosw, L)
E1 = np.einsum("khw,kc->hwc", cosw * V, L)
ao = E1 / np.maximum(E0, 1e-8)
for name, im in (("x AO(rgb irradiance ratio)", img * ao), ("x AO(scalar cosine-weighted visibility)", img * ((V * cosw).sum(0) / np.maximum(cosw.sum(0), 1e-8))[..., None])):
    print("%-45s raw %.2f dB, mean-matched %.2f dB" % (name, psnr(im, ref), psnr(im * ref.mean() / im.mean(), ref)))

