#!/usr/bin/env python3
"""Run the sample photograph's inversion and print the SHA-256 of what it writes (final_envmap.hdr, best_results/*) + its PSNR as one JSON line.
    python tools/pipeline_hashes.py --model_name none --out /tmp/x [--num_epochs N] [--repeat K]
Used by tests/test_gpu_configs.py::test_the_pipeline_is_reproducible and by hand."""
import argparse
import glob
import hashlib
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hashes(out_dir: str):
    files = sorted(glob.glob(os.path.join(out_dir, "indoor2", "best_results", "*")) + glob.glob(os.path.join(out_dir, "indoor2", "final_envmap.hdr")))
    return {os.path.relpath(f, out_dir): hashlib.sha256(open(f, "rb").read()).hexdigest()[:16] for f in files}


def one(model_name: str, out_dir: str, num_epochs: int, seed: int):
    import torch

    spec = importlib.util.spec_from_file_location("real_image", os.path.join(ROOT, "tools", "real_image.py"))
    ri = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ri)
    torch.manual_seed(seed)
    args = ri.parse(["--sample", "indoor2", "--model_name", model_name, "--out", out_dir, "--num_epochs", str(num_epochs)])
    args.out = os.path.abspath(out_dir)
    res = ri.run(args)
    return {"psnr": res["psnr_vs_photo"]["this_build_final_render"], "hashes": hashes(args.out), "log": [l.split("] ")[-1] for l in res["log"]],
            "stage_digests": [list(d) for d in res["stage_digests"]]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_name", default="none")
    ap.add_argument("--out", required=True)
    ap.add_argument("--num_epochs", type=int, default=5000)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--write-golden", default=None, help="write run 0's stage digests, file hashes and PSNR to this JSON (tests/golden/indoor2_digests_<mode>.json)")
    a = ap.parse_args()
    runs = [one(a.model_name, os.path.join(a.out, f"run{k}"), a.num_epochs, a.seed) for k in range(a.repeat)]
    if a.write_golden:
        import torch

        with open(a.write_golden, "w") as f:
            json.dump({"model_name": a.model_name, "num_epochs": a.num_epochs, "seed": a.seed, "device": torch.cuda.get_device_properties(0).gcnArchName,
                       "psnr": runs[0]["psnr"], "stage_digests": runs[0]["stage_digests"], "hashes": runs[0]["hashes"]}, f, indent=1)
    print(json.dumps(runs))
