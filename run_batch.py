#!/usr/bin/env python3
"""Inverse-render a list of images sharded over the GPUs of one node (BASELINE config 3).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run_batch.py --images a.png b.png ... \\
           --save_path out --opt_src arm --opt_order rm a --opt_env_from 2 [--model_name none|pos_mlp]
(single process / single GPU without the launcher).  `--synthetic K` optimises K seeded synthetic scenes instead of files."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", nargs="*", default=[])
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--save_path", type=str, default=None)
    ap.add_argument("--opt_src", type=str, default="arm")
    ap.add_argument("--opt_order", type=str, nargs="+", default=["arm"])
    ap.add_argument("--opt_env_from", type=int, default=0)
    ap.add_argument("--model_name", type=str, default="none", choices=["none", "pos_mlp"])
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--num_epochs", type=int, default=5000)
    a = ap.parse_args(argv)
    import numpy as np
    import torch

    from materialist_amd import batch, optimize, pipeline, render, synthetic

    rank, world, local = batch.init_distributed()
    dev = torch.device("cuda", local)
    paths = list(a.images) if a.images else [f"synthetic:{i}" for i in range(a.synthetic)]

    def process(i, path, cfg):
        if path.startswith("synthetic:"):
            sc = synthetic.make_scene(int(path.split(":")[1]), cfg["size"], cfg["size"])
            t = lambda x: torch.from_numpy(x).to(dev)
            scene = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True, device=dev)
            scene._set("emitter.data", t(sc.light))
            with torch.no_grad():
                gt = render.render_w_brdf(scene, t(sc.albedo), t(sc.roughness), t(sc.metallic), None, cfg["spp"]).clone()
            mat = {"albedo": t(sc.init_albedo), "roughness": t(sc.init_roughness), "metallic": t(sc.init_metallic), "gt_image": gt}
            scene = render.load_estimated_mesh(t(sc.depth), use_mesh_normal=True, device=dev)
            res = optimize.optimize_envmap_ARMN(scene, mat, optimize_order=cfg["opt_order"], spp=cfg["spp"], opt_env_from=cfg["opt_env_from"],
                                                opt_src=cfg["opt_src"], num_epochs=cfg["num_epochs"], model_name=cfg["model_name"])
        else:
            name = os.path.splitext(os.path.basename(path))[0]
            res = pipeline.inverse_image(path, name, cfg["opt_src"], cfg["opt_order"], False, cfg["opt_env_from"], cfg["save_path"], cfg["model_name"],
                                         size=cfg["size"], spp=cfg["spp"], num_epochs=cfg["num_epochs"], device=str(dev), log=lambda *_: None)
        return [res["best_loss"], res["psnr"], float(sum(max(t.epoch, 0) + 1 for t in res["trace"] if t.phase != "end"))]

    cfg = {"save_path": a.save_path, "opt_src": a.opt_src, "opt_order": a.opt_order, "opt_env_from": a.opt_env_from, "model_name": a.model_name,
           "size": a.size, "spp": a.spp, "num_epochs": a.num_epochs}
    rows = batch.run_batch(paths, cfg, process)
    if rank == 0:
        for r in rows:
            print(json.dumps({"image": r["path"], "rank": r["rank"], "best_loss_mse": r["values"][0], "psnr_db": r["values"][1], "iterations": int(r["values"][2])}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
