#!/usr/bin/env python3
"""BASELINE configs[1] on the reference's own sample run: examples/indoor2.png (512x512) from MaterialNet's shipped predictions,
`--model_name pos_mlp --opt_order rm a --opt_env_from 2 --opt_src a`, spp 64, on one MI355X; compares with what the
reference's Mitsuba-based run of the same command arrived at (tests/golden/indoor2.npz, see gen_indoor2.py).

    python tools/real_image.py [--model_name none] [--out gpurun_out/real_image]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def unpack_fixture(dst: str):
    from PIL import Image

    from materialist_amd.imageio_exr import write_exr

    z = np.load(os.path.join(ROOT, "tests", "golden", "indoor2.npz"))
    os.makedirs(dst, exist_ok=True)
    Image.fromarray(z["image_srgb_u8"]).save(os.path.join(dst, "indoor2.png"))
    write_exr(os.path.join(dst, "albedoPred.exr"), z["albedo_pred_f16"].astype(np.float32))
    H, W = z["depth_pred_f32"].shape
    up = np.zeros((H, W, 3), np.float32)
    up[..., 2] = 1
    write_exr(os.path.join(dst, "normalPred.exr"), up)            # geometric normals are used (use_mesh_normal); placeholder
    Image.fromarray(z["roughness_pred_u8"]).save(os.path.join(dst, "roughnessPred.png"))
    Image.fromarray(z["metallic_pred_u8"]).save(os.path.join(dst, "metallicPred.png"))
    write_exr(os.path.join(dst, "depthPred.exr"), z["depth_pred_f32"])
    return z


def unpack_jinjya(dst: str, out_dir: str):
    """The outdoor sample (tests/golden/jinjya256.npz, 256x256): linear photograph, predictions, and the sky mask, which goes
    where the reference keeps it: <output dir>/mesh_mask.png (inverse_img_w_mi.py:713-724)."""
    from PIL import Image

    from materialist_amd.imageio_exr import write_exr

    z = np.load(os.path.join(ROOT, "tests", "golden", "jinjya256.npz"))
    os.makedirs(dst, exist_ok=True)
    os.makedirs(out_dir, exist_ok=True)
    write_exr(os.path.join(dst, "jinjya.exr"), z["gt_linear_f16"].astype(np.float32))
    write_exr(os.path.join(dst, "albedoPred.exr"), z["albedo_pred_f16"].astype(np.float32))
    H, W = z["depth_pred_f32"].shape
    up = np.zeros((H, W, 3), np.float32)
    up[..., 2] = 1
    write_exr(os.path.join(dst, "normalPred.exr"), up)
    Image.fromarray(z["roughness_pred_u8"]).save(os.path.join(dst, "roughnessPred.png"))
    Image.fromarray(z["metallic_pred_u8"]).save(os.path.join(dst, "metallicPred.png"))
    write_exr(os.path.join(dst, "depthPred.exr"), z["depth_pred_f32"])
    Image.fromarray((z["mesh_mask"] * 255).astype(np.uint8)).save(os.path.join(out_dir, "mesh_mask.png"))
    return z


def run_jinjya(args):
    """`--model_name none --opt_order rm a --opt_env_from 2 --opt_src a` (the reference's config.json for this sample)."""
    import torch

    from materialist_amd.pipeline import inverse_image

    args.out = os.path.abspath(args.out)
    tmp = tempfile.mkdtemp(prefix="jinjya_")
    z = unpack_jinjya(tmp, os.path.join(args.out, "jinjya"))
    gt = z["gt_linear_f16"].astype(np.float32)
    lines = []
    t0 = time.time()
    res = inverse_image(os.path.join(tmp, "jinjya.exr"), "jinjya", opt_src="a", opt_order=args.opt_order, opt_env_from=args.opt_env_from,
                        save_path=args.out, model_name=args.model_name, size=256, spp=64, num_epochs=args.num_epochs, pred_dir=tmp,
                        log=lambda s: lines.append(s))
    torch.cuda.synchronize()
    mine = res["final_render"].detach().cpu().numpy()
    ref = z["ref_render_f16"].astype(np.float32)
    sky = z["mesh_mask"]
    return {"config": f"jinjya 256x256 (sky mask {sky.mean():.0%}), --model_name {args.model_name} --opt_order {' '.join(args.opt_order)} "
                      f"--opt_env_from {args.opt_env_from} --opt_src a, spp 64",
            "wall_s": round(time.time() - t0, 2),
            "psnr_vs_photo": {"this_build_final_render": round(psnr(mine * (gt.mean() / mine.mean()), gt), 2),
                              "this_build_ground_only": round(psnr((mine * (gt.mean() / mine.mean()))[~sky], gt[~sky]), 2),
                              "this_build_sky_only": round(psnr((mine * (gt.mean() / mine.mean()))[sky], gt[sky]), 2),
                              "reference_shipped_render": round(psnr(ref * (gt.mean() / ref.mean()), gt), 2)},
            "log": lines}


def psnr(a, b):
    g = lambda x: np.clip(x, 0, 1) ** (1 / 2.2)
    return float(-10 * np.log10(np.mean((g(a) - g(b)) ** 2)))


def run(args):
    import torch

    if args.model_name is None:
        args.model_name = "pos_mlp"

    from materialist_amd import loss as _loss
    from materialist_amd.pipeline import inverse_image

    tmp = tempfile.mkdtemp(prefix="indoor2_")
    z = unpack_fixture(tmp)
    gt = _loss.srgb_to_linear(torch.from_numpy(z["image_srgb_u8"].astype(np.float32) / 255)).numpy()
    lines, digests = [], []
    t0 = time.time()
    res = inverse_image(os.path.join(tmp, "indoor2.png"), "indoor2", opt_src="a", opt_order=args.opt_order, opt_env_from=args.opt_env_from,
                        save_path=args.out, model_name=args.model_name, size=512, spp=64, num_epochs=args.num_epochs, pred_dir=tmp,
                        log=lambda s: lines.append(s), digests=digests)
    torch.cuda.synchronize()
    wall = time.time() - t0
    mine = res["final_render"].detach().cpu().numpy()
    ratio = gt.mean() / mine.mean()
    ref = z["ref_render_f16"].astype(np.float32)
    f = lambda k: res[k].detach().cpu().numpy().reshape(512, 512, -1)
    ref_maps = {"albedo": z["ref_albedo_u8"] / 255.0, "roughness": z["ref_roughness_u8"][..., None] / 255.0, "metallic": z["ref_metallic_u8"][..., None] / 255.0}
    init = {"albedo": z["albedo_pred_f16"].astype(np.float32), "roughness": z["roughness_pred_u8"][..., None] / 255.0, "metallic": z["metallic_pred_u8"][..., None] / 255.0}
    out = {
        "config": f"indoor2.png 512x512, --model_name {args.model_name} --opt_order {' '.join(args.opt_order)} --opt_env_from {args.opt_env_from} --opt_src a, spp 64",
        "wall_s": round(wall, 2),
        "psnr_vs_photo": {"this_build_final_render": round(psnr(mine * ratio, gt), 2), "this_build_reported": round(res["psnr"], 2),
                          "reference_mitsuba_final_render": round(psnr(ref * (gt.mean() / ref.mean()), gt), 2),
                          "reference_mitsuba_final_render_unscaled": round(psnr(ref, gt), 2)},
        "psnr_build_render_vs_mitsuba_render": round(psnr(mine * (ref.mean() / mine.mean()), ref), 2),
        "mean_abs_map_difference": {k: {"final_vs_reference_final": round(float(np.abs(f(k) - ref_maps[k]).mean()), 4),
                                        "initial_vs_reference_final": round(float(np.abs(np.clip(init[k], 0.07 if k == "roughness" else 0, 1) - ref_maps[k]).mean()), 4)}
                                    for k in ("albedo", "roughness", "metallic")},
    }
    out["log"] = lines
    out["stage_digests"] = digests        # (stage, SHA-256 of the stage's tensors): compare with tests/golden/indoor2_digests.json to localise a run that differs
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--sample", default="indoor2", choices=["indoor2", "jinjya"])
    ap.add_argument("--model_name", default=None, help="default: pos_mlp for indoor2, none for jinjya (the reference's own configs)")
    ap.add_argument("--opt_order", nargs="+", default=["rm", "a"])
    ap.add_argument("--opt_env_from", type=int, default=2)
    ap.add_argument("--num_epochs", type=int, default=5000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "real_image"))
    return ap.parse_args(argv)


def main():
    args = parse()
    args.out = os.path.abspath(args.out)        # a relative --save_path would land under output_imgs/ (inverse_img_w_mi.py:82-104)
    if args.model_name is None:
        args.model_name = "pos_mlp" if args.sample == "indoor2" else "none"
    out = run(args) if args.sample == "indoor2" else run_jinjya(args)
    import torch

    pr = torch.cuda.get_device_properties(0)        # (the boxes of the pool are not all alike: which one this was)
    out["device"] = {"name": pr.name, "compute_units": pr.multi_processor_count, "memory_gb": round(pr.total_memory / 2 ** 30, 1)}
    print(json.dumps(out, indent=1))
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, f"real_image_{args.model_name}.json" if args.sample == "indoor2" else f"real_image_jinjya_{args.model_name}.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
