"""The env phase of the none-mode pipeline, run after run in one process: what does it hand over?"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("real_image", os.path.join(ROOT, "tools", "real_image.py"))
ri = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ri)
from materialist_amd import loop  # noqa: E402

log = []


def cs(t):
    t = t.detach().double().reshape(-1)
    return f"{float(t.sum()):.12g}"


orig_poll = loop.FusedEnvPhase.poll


def poll(self):
    out = orig_poll(self)
    if self.t in (100, 1000, 5000) and len(log) < 6:
        log.append((self.t, "best_env", cs(self.best_env), "head", cs(self.head().detach()), "best_mse", float(out["best_mse"][0]), "mse", float(out["mse"][0]),
                    "hist", cs(self.hist[: self.t])))
    return out


loop.FusedEnvPhase.poll = poll
for k in range(3):
    log.clear()
    args = ri.parse(["--sample", "indoor2", "--model_name", "none", "--out", "/tmp/ri", "--num_epochs", "5000"])
    ri.run(args)
    print("run", k, log[:3], flush=True)
