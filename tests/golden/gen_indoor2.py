"""Builds tests/golden/indoor2.npz from DATA files the reference ships (run in the build container only):

  examples/indoor2.png                          the input photograph of the reference's own sample run (config.json: img_path)
  output_imgs/indoor/{albedoPred.exr, roughnessPred.png, metallicPred.png, depthPred.exr}
                                                MaterialNet's predictions for it (the trained weights are unreachable, SURVEY F3,
                                                so these files are the only way to start from the reference's initial guess)
  output_imgs/indoor/best_results/*             what the reference's optimisation (Mitsuba, pos_mlp, 'rm a', opt_env_from 2,
                                                spp 64) arrived at: maps, 16x32 envmap, its final render

Only pixels are stored (8-bit where the source is 8-bit, fp16 for float maps, fp32 for depth whose finite differences give the
normals); no reference code.  Used by tests/test_gpu_parity.py::test_real_image_* and tools/real_image.py.
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from materialist_amd.imageio_exr import read_exr  # noqa: E402
from materialist_amd.imageio_hdr import read_hdr  # noqa: E402

REF = "/root/reference"
OUT = f"{REF}/output_imgs/indoor"


def u8(path):
    return np.asarray(Image.open(path).convert("L"), dtype=np.uint8)


def q8(x):
    return (np.clip(x, 0, 1) * 255 + 0.5).astype(np.uint8)


def main():
    d = {
        "image_srgb_u8": np.asarray(Image.open(f"{REF}/examples/indoor2.png").convert("RGB"), dtype=np.uint8),
        "albedo_pred_f16": read_exr(f"{OUT}/albedoPred.exr").astype(np.float16),
        "roughness_pred_u8": u8(f"{OUT}/roughnessPred.png"),
        "metallic_pred_u8": u8(f"{OUT}/metallicPred.png"),
        "depth_pred_f32": read_exr(f"{OUT}/depthPred.exr")[..., 0].astype(np.float32),
        "ref_albedo_u8": q8(read_exr(f"{OUT}/best_results/albedo.exr")),
        "ref_roughness_u8": q8(read_exr(f"{OUT}/best_results/roughness.exr")[..., 0]),
        "ref_metallic_u8": q8(read_exr(f"{OUT}/best_results/metallic.exr")[..., 0]),
        "ref_render_f16": read_exr(f"{OUT}/best_results/rendered_img.exr").astype(np.float16),
        "ref_envmap_f32": read_hdr(f"{OUT}/best_results/envmap.hdr").astype(np.float32),
    }
    np.savez_compressed(os.path.join(HERE, "indoor2.npz"), **d)
    print({k: (v.shape, str(v.dtype)) for k, v in d.items()})


if __name__ == "__main__":
    main()
