#!/usr/bin/env python3
"""Command line of the inverse-rendering pipeline, same flags as the reference's inverse_img_w_mi.py (:771-801) plus
`--size`, `--spp`, `--num_epochs`, `--pred_dir`.  Runs on libmatpbr.so (MI355X); see materialist_amd/pipeline.py."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Single-image inverse rendering on the matpbr HIP kernels")
    ap.add_argument("--img_inverse_path", type=str, required=True)
    ap.add_argument("--save_name", type=str, required=True)
    ap.add_argument("--opt_src", type=str, required=True, default="arm", help="'arm' subsets, or 'skip' to resume from best_results/")
    ap.add_argument("--opt_order", type=str, nargs="+", default=["arm"])
    ap.add_argument("--use_mask", action="store_true")
    ap.add_argument("--opt_env_from", type=int, default=0)
    ap.add_argument("--save_path", type=str, default=None)
    ap.add_argument("--model_name", type=str, default="pos_mlp", choices=["pos_mlp", "none"],
                    help="the reference parses and ignores this flag (it always runs pos_mlp, F4); here it is honoured, default pos_mlp")
    ap.add_argument("--size", type=int, default=512, help="render resolution (the reference hard-codes 512)")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--num_epochs", type=int, default=5000)
    ap.add_argument("--pred_dir", type=str, default=None, help="directory with *Pred.exr/png initial maps (MaterialNet output layout)")
    ap.add_argument("--matnet_weights", type=str, default=None, help="matnet_weights.pth (Lez/MatNet on the HF hub) for the MaterialNet initial guess")
    ap.add_argument("--geometry", type=str, default="mesh", choices=["mesh", "depth"],
                    help="per-pixel geometric normals: from the reference's mesh of the depth map, gap closing at depth edges included (mesh_recon.py, "
                         "default), or central differences of the depth map itself")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    from materialist_amd.pipeline import inverse_image

    res = inverse_image(args.img_inverse_path, args.save_name, args.opt_src, args.opt_order, args.use_mask, args.opt_env_from,
                        args.save_path, args.model_name, size=args.size, spp=args.spp, num_epochs=args.num_epochs, pred_dir=args.pred_dir,
                        matnet_weights=args.matnet_weights, geometry=args.geometry)
    print(f"done: PSNR {res['psnr']:.2f} dB, best loss_mse {res['best_loss']:.6f}, outputs in {res['output_dir']}")


if __name__ == "__main__":
    main()
