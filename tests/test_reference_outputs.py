"""Weak pins on the sample outputs the reference ships (`output_imgs/{indoor,jinjya}/`, SURVEY.md section 8c "weak pins").

These files exist only in the build container (`/root/reference`); the tests skip elsewhere.  They check
  * the PIZ decoder of `materialist_amd.imageio_exr` against an independent encoding of the same picture: the 8-bit PNG panels
    the reference wrote from the same tensors (`opt_env_img.png` = [gt | Mitsuba render | envmap], gamma 2.2; `gt_image.png`);
  * the build's render definition (DESIGN.md section 1, fp32 oracle) against **Mitsuba's own render** of the reference's optimised
    maps (`best_results/rendered_img.exr`), the only image-level evidence of `mi.render` available.  The residual (occlusion,
    inter-reflection, MC noise, stretched triangles at depth edges) is documented in DESIGN.md section 5; the threshold here
    only guards against convention regressions (camera, normals from depth, envmap -> SH, gamma).
"""
import os

import numpy as np
import pytest

REF = "/root/reference/output_imgs"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference sample outputs are only in the build container")


def _gamma(x):
    return np.clip(x, 0, 1) ** (1 / 2.2)


def _psnr(a, b):
    return -10 * np.log10(np.mean((_gamma(a) - _gamma(b)) ** 2))


def _png(path):
    from PIL import Image

    return np.asarray(Image.open(path).convert("RGB")).astype(np.float32) / 255


@pytest.mark.parametrize("scene", ["indoor", "jinjya"])
def test_piz_decode_matches_the_png_panels(scene):
    from materialist_amd.imageio_exr import read_exr

    d = f"{REF}/{scene}"
    rough = read_exr(f"{d}/best_results/roughness.exr")
    assert rough.shape == (512, 512, 1) and np.isfinite(rough).all()
    assert rough.min() == np.float32(0.07) and rough.max() == np.float32(1.0)      # the loop's clamp values, bit-exact (:420-423)
    if scene == "indoor":      # (jinjya's rendered_img.exr is not the render of its panel: 19.8 dB even after the best rescale)
        render = read_exr(f"{d}/best_results/rendered_img.exr")
        panel = _png(f"{d}/opt_env_img.png")[:, 512:1024]
        assert np.mean((_gamma(render) - panel) ** 2) < 10 ** (-3.3)                 # > 33 dB: 8-bit quantisation + last-vs-best epoch
    gt = read_exr(f"{d}/gt_image.exr")
    assert np.abs(_gamma(gt) - _png(f"{d}/opt_env_img.png")[:, :512]).max() < 1.5 / 255


def test_render_definition_against_mitsuba_render_of_the_reference_maps():
    from materialist_amd import sh
    from materialist_amd.imageio_exr import read_exr
    from materialist_amd.imageio_hdr import read_hdr
    from oracle.oracle import Oracle

    d = f"{REF}/indoor"
    o = Oracle(np.float32)
    a, r, m = (read_exr(f"{d}/best_results/{k}.exr") for k in ("albedo", "roughness", "metallic"))
    ref = read_exr(f"{d}/best_results/rendered_img.exr")
    depth = read_exr(f"{d}/depthPred.exr")[..., 0]
    depth = 2 * depth.max() - depth                                                  # inverse_img_w_mi.py:722
    n = o.normals_from_depth(depth)                                                  # config.json: use_mesh_normal = true
    env = read_hdr(f"{d}/best_results/envmap.hdr")
    coef = (sh.envmap_to_sh_matrix(16, 32) @ env.reshape(512, 3).astype(np.float64)).astype(np.float32)
    img = o.shade_fwd(a, r, m, n, coef, 64)
    raw = _psnr(img, ref)
    matched = _psnr(img * (ref.mean() / img.mean()), ref)                            # the BRDF loss is scale-free (:388-391)
    print(f"build render vs Mitsuba render, indoor: {raw:.2f} dB raw, {matched:.2f} dB mean-matched")
    assert raw > 20.5 and matched > 24.5
