"""How much of the residual between the build's render and Mitsuba's render of the reference's optimised maps (indoor sample) is the
SH25 low-pass of the 16 x 32 envmap?  Direct lighting summed over the 512 texels themselves (texel radiance x BRDF x cosine x solid
angle, every texel upsampled to s x s sub-texels so that the narrow GGX lobes are resolved) against the SH25 light of the kernels.
Build container only (reads /root/reference sample outputs); numpy + the CPU oracle's N-lane BRDF."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd import sh  # noqa: E402
from materialist_amd.imageio_exr import read_exr  # noqa: E402
from materialist_amd.imageio_hdr import read_hdr  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

REF = "/root/reference/output_imgs/indoor"
g = lambda x: np.clip(x, 0, 1) ** (1 / 2.2)
psnr = lambda a, b: -10 * np.log10(np.mean((g(a) - g(b)) ** 2))


def main(step=2, sub=4):
    o = Oracle(np.float32)
    a, r, m = (read_exr(f"{REF}/best_results/{k}.exr") for k in ("albedo", "roughness", "metallic"))
    ref = read_exr(f"{REF}/best_results/rendered_img.exr")
    depth = read_exr(f"{REF}/depthPred.exr")[..., 0]
    depth = 2 * depth.max() - depth
    n = o.normals_from_depth(depth)
    env = read_hdr(f"{REF}/best_results/envmap.hdr").astype(np.float64)
    coef = (sh.envmap_to_sh_matrix(16, 32) @ env.reshape(512, 3)).astype(np.float32)
    sl = (slice(None, None, step), slice(None, None, step))                 # every step-th pixel: the statistics do not need all of them
    img = o.shade_fwd(a, r, m, n, coef, 64)[sl]
    refs = ref[sl]
    print("SH25 light (the kernels'):         raw %.2f dB, mean-matched %.2f dB" % (psnr(img, refs), psnr(img * refs.mean() / img.mean(), refs)))
    H, W = depth.shape
    f = 0.5 * W / math.tan(0.5 * math.radians(35.0))
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    p = np.stack([(jj - 0.5 * (W - 1)) / f, -(ii - 0.5 * (H - 1)) / f, -np.ones((H, W))], -1)
    wo = (-p / np.linalg.norm(p, axis=-1, keepdims=True)).astype(np.float32)[sl].reshape(-1, 3)
    A, R, M, N = a[sl].reshape(-1, 3), r[sl].reshape(-1), m[sl].reshape(-1), n[sl].reshape(-1, 3)
    He, We = 16 * sub, 32 * sub
    dirs = sh.envmap_directions(He, We).reshape(-1, 3).astype(np.float32)
    dom = sh.envmap_solid_angles(He, We).reshape(-1)
    for name, rad in (("texels (nearest)", np.repeat(np.repeat(env, sub, 0), sub, 1).reshape(-1, 3)),
                      ("SH25 reconstruction at the sub-texels", np.maximum(sh.sh_basis(dirs.astype(np.float64)) @ coef.astype(np.float64), -1e9))):
        out = np.zeros_like(A, dtype=np.float64)
        for t in range(dirs.shape[0]):
            wi = np.broadcast_to(dirs[t], wo.shape)
            fcos, _ = o.eval_brdf(wi, wo, N, A, R, M)                          # f * cos
            out += fcos.astype(np.float64) * rad[t] * dom[t]
        im = out.reshape(img.shape).astype(np.float32)
        print("%-34s raw %.2f dB, mean-matched %.2f dB   (vs the SH25 kernel render: %.2f dB)" %
              (name + ":", psnr(im, refs), psnr(im * refs.mean() / im.mean(), refs), psnr(im, img)))


if __name__ == "__main__":
    main()
