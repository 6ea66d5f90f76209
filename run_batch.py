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
    ap.add_argument("--model_name", type=str, default="pos_mlp", choices=["none", "pos_mlp"],
                    help="pos_mlp: the reference's behaviour (inverse_img_w_mi.py:782 always runs it); none: optimise the maps directly "
                         "(a rank's synthetic shard then runs as ONE batch in the kernels' batch dimension)")
    ap.add_argument("--use_mask", action="store_true",
                    help="inverse_img_w_mi.py:779: every photograph's <output dir>/best_results/mask.png (:702-711); per-image masked means")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--num_epochs", type=int, default=5000)
    ap.add_argument("--matnet_weights", type=str, default=None,
                    help="MaterialNet weights (the reference downloads Lez/MatNet matnet_weights.pth, inverse_img_w_mi.py:648-654): read by rank 0 "
                         "only and broadcast to the other ranks as one flat buffer")
    ap.add_argument("--pred_dir", type=str, default=None,
                    help="MaterialNet predictions in the reference's file layout (albedoPred.exr ...): <pred_dir>/<image name>/ or <pred_dir> itself")
    a = ap.parse_args(argv)
    import numpy as np
    import torch

    from materialist_amd import batch, optimize, pipeline, render, synthetic

    rank, world, local = batch.init_distributed()
    dev = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    paths = list(a.images) if a.images else [f"synthetic:{i}" for i in range(a.synthetic)]
    matnet = None
    if a.matnet_weights and a.images:            # weights: one reader, one broadcast
        from materialist_amd.dist import broadcast_state_dict
        from materialist_amd.materialnet import MaterialNet

        sd = torch.load(a.matnet_weights, map_location="cpu", weights_only=True) if rank == 0 else None
        matnet = MaterialNet()
        matnet.load_state_dict(broadcast_state_dict(sd, dev))
        matnet = matnet.to(dev).eval()

    def pred_dir_of(path):
        if not a.pred_dir:
            return None
        sub = os.path.join(a.pred_dir, os.path.splitext(os.path.basename(path))[0])
        return sub if os.path.isdir(sub) else a.pred_dir

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
            res = pipeline.inverse_image(path, name, cfg["opt_src"], cfg["opt_order"], cfg["use_mask"], cfg["opt_env_from"], cfg["save_path"], cfg["model_name"],
                                         size=cfg["size"], spp=cfg["spp"], num_epochs=cfg["num_epochs"], device=str(dev), log=lambda *_: None,
                                         matnet=matnet, pred_dir=pred_dir_of(path))
        return [res["best_loss"], res["psnr"], float(sum(max(t.epoch, 0) + 1 for t in res["trace"] if t.phase != "end"))]

    def process_shard(ids, shard_paths, cfg):
        """`--model_name none`: the rank's whole shard as one batch (B images in the kernels' batch dimension, per-image lights, SaveBest
        and EarlyStopping per image on the device) -- synthetic scenes, or photographs with their MaterialNet predictions."""
        if not shard_paths[0].startswith("synthetic:"):
            names = [os.path.splitext(os.path.basename(p))[0] for p in shard_paths]
            res = pipeline.inverse_images_batched(shard_paths, names, cfg["opt_src"], cfg["opt_order"], cfg["opt_env_from"], cfg["save_path"],
                                                  size=cfg["size"], spp=cfg["spp"], num_epochs=cfg["num_epochs"],
                                                  pred_dirs=[pred_dir_of(p) for p in shard_paths], device=str(dev), matnet=matnet, log=lambda *_: None,
                                                  use_mask=cfg["use_mask"])
            its = float(sum(max(t.epoch, 0) + 1 for t in res["trace"] if t.phase != "end"))
            return [[bl, ps, its] for bl, ps in zip(res["best_loss_per_image"], res["psnr_per_image"])]
        scs = [synthetic.make_scene(int(p.split(":")[1]), cfg["size"], cfg["size"]) for p in shard_paths]
        st = lambda k: torch.from_numpy(np.stack([getattr(s, k) for s in scs])).to(dev)
        depth = st("depth")
        scene = render.load_estimated_mesh(depth, use_mesh_normal=True, device=dev)
        scene._set("emitter.data", st("light"))
        with torch.no_grad():
            gt = render.render_w_brdf(scene, st("albedo"), st("roughness"), st("metallic"), None, cfg["spp"]).clone()
        mat = {"albedo": st("init_albedo"), "roughness": st("init_roughness"), "metallic": st("init_metallic"), "gt_image": gt}
        scene = render.load_estimated_mesh(depth, use_mesh_normal=True, device=dev)
        res = optimize.optimize_envmap_ARMN(scene, mat, optimize_order=cfg["opt_order"], spp=cfg["spp"], opt_env_from=cfg["opt_env_from"],
                                            opt_src=cfg["opt_src"], num_epochs=cfg["num_epochs"], model_name="none")
        its = float(sum(max(t.epoch, 0) + 1 for t in res["trace"] if t.phase != "end"))
        return [[bl, ps, its] for bl, ps in zip(res["best_loss_per_image"], res["psnr_per_image"])]

    cfg = {"save_path": a.save_path, "opt_src": a.opt_src, "opt_order": a.opt_order, "opt_env_from": a.opt_env_from, "model_name": a.model_name,
           "size": a.size, "spp": a.spp, "num_epochs": a.num_epochs, "use_mask": bool(a.use_mask)}
    # a rank's shard runs as one batch in --model_name none mode (synthetic scenes, or photographs; a photograph's mesh_mask.png and
    # the pixels its mesh leaves uncovered are per-image masks of the batch)
    batched = a.model_name == "none" and bool(paths) and "n" not in str(a.opt_order) and (
        all(p.startswith("synthetic:") for p in paths) or not any(p.startswith("synthetic:") for p in paths))
    rows = batch.run_batch(paths, cfg, process, process_shard=process_shard if batched else None)
    if rank == 0:
        for r in rows:
            v = r["values"] + [float("nan")] * 3
            print(json.dumps({"image": r["path"], "rank": r["rank"], "best_loss_mse": v[0], "psnr_db": v[1],
                              "iterations": int(v[2]) if v[2] == v[2] else None, "error": r["error"]}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
