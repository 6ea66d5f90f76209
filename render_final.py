#!/usr/bin/env python3
"""Command line of the forward-only re-render, same flags as the reference's render_final.py (:420-449); `--mode rolling`
(the function the reference ships but never wires) is available.  See materialist_amd/relight.py."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args(argv=None):
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter, description="re-render an optimised scene under new lighting")
    ap.add_argument("--env_path", required=False, default=None, type=str)
    ap.add_argument("--save_name", required=True, type=str)
    ap.add_argument("--mode", required=True, type=str, help="real, rolling (oi = object insertion is not part of this build)")
    ap.add_argument("--input_path", required=False, default=None, type=str)
    ap.add_argument("--save_path", required=False, default=None, type=str)
    ap.add_argument("--frames", type=int, default=36)
    ap.add_argument("--rotation_step", type=float, default=10.0)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--edit_albedo", type=float, nargs=3, default=None, metavar=("DH", "DS", "DV"),
                    help="HSV shift of the albedo inside best_results/mask.png (the reference's `edit` dict, hard-wired to None in its CLI)")
    ap.add_argument("--edit_roughness", type=float, default=None, help="constant roughness inside the mask")
    ap.add_argument("--edit_metallic", type=float, default=None, help="constant metallic inside the mask")
    return ap.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    from materialist_amd import relight

    edit = {"albedo": a.edit_albedo, "roughness": a.edit_roughness, "metallic": a.edit_metallic}
    if a.mode == "real":
        print("Wrote file to", relight.render_real(a.save_name, a.env_path, a.input_path, a.save_path, a.spp, edit=edit))
    elif a.mode == "rolling":
        res = relight.render_rolling_envmap(a.save_name, a.env_path, a.frames, a.rotation_step, a.input_path, a.save_path, a.spp, edit=edit)
        print(f"Animation saved to {res['gif']}\nIndividual frames saved to {res['animation_dir']}")
    elif a.mode == "oi":
        raise NotImplementedError("object insertion (render_final.py:100-141,207-237) is not part of this build")
    else:
        raise ValueError("Invalid mode")


if __name__ == "__main__":
    main()
