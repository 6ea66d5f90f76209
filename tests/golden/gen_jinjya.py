"""Builds tests/golden/jinjya256.npz from DATA files of the reference's outdoor sample (`output_imgs/jinjya`, run in the build
container only): the photograph as the reference linearised it (gt_image.exr), MaterialNet's shipped predictions, the
`mesh_mask.png` that marks the sky (pixels without geometry), and what the reference's own run (`--model_name none --opt_order rm a
--opt_env_from 2 --opt_src a`) arrived at.  Everything is resampled to 256x256 with the pipeline's own resize so the fixture stays
small; only pixels are stored, no reference code.  Used by tests/test_gpu_parity.py::test_outdoor_sample_with_mesh_mask."""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from materialist_amd.imageio_exr import read_exr  # noqa: E402
from materialist_amd.imageio_hdr import read_hdr  # noqa: E402
from materialist_amd.pipeline import center_crop_and_resize  # noqa: E402

OUT = "/root/reference/output_imgs/jinjya"
S = 256


def rs(x):
    x = np.asarray(x, np.float32)
    y = center_crop_and_resize(x if x.ndim == 3 else x[..., None], (S, S))
    return y if x.ndim == 3 else y[..., 0]


def main():
    mask = np.asarray(Image.open(f"{OUT}/mesh_mask.png"))[..., 0] > 0
    d = {
        "gt_linear_f16": rs(read_exr(f"{OUT}/gt_image.exr")).astype(np.float16),
        "albedo_pred_f16": rs(read_exr(f"{OUT}/albedoPred.exr")).astype(np.float16),
        "roughness_pred_u8": (rs(np.asarray(Image.open(f"{OUT}/roughnessPred.png").convert("L"), np.float32) / 255) * 255 + 0.5).astype(np.uint8),
        "metallic_pred_u8": (rs(np.asarray(Image.open(f"{OUT}/metallicPred.png").convert("L"), np.float32) / 255) * 255 + 0.5).astype(np.uint8),
        "depth_pred_f32": rs(read_exr(f"{OUT}/depthPred.exr")[..., 0]).astype(np.float32),
        "mesh_mask": rs(mask.astype(np.float32)) > 0.5,
        "ref_render_f16": rs(read_exr(f"{OUT}/best_results/rendered_img.exr")).astype(np.float16),
        "ref_envmap_f32": read_hdr(f"{OUT}/best_results/envmap.hdr").astype(np.float32),
    }
    np.savez_compressed(os.path.join(HERE, "jinjya256.npz"), **d)
    print({k: (v.shape, str(v.dtype)) for k, v in d.items()}, "sky fraction", float(d["mesh_mask"].mean()))


if __name__ == "__main__":
    main()
