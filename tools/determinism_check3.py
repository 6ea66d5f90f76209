"""Where do the runs of one process part ways?  Checksums of what the first BRDF part is given, and of what it leaves, run after run."""
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
orig_init = loop.FusedBrdfPhase.__init__
orig_run = loop.FusedBrdfPhase.run


def cs(t):
    t = t.detach().double().reshape(-1)
    return f"{float(t.sum()):.17g}/{float((t * torch.arange(1, t.numel() + 1, device=t.device, dtype=torch.float64) % 977.0).sum()):.17g}"


def init(self, scene, gt, a, r, m, **kw):
    orig_init(self, scene, gt, a, r, m, **kw)
    if len(log) < 1:
        log.append(("in", cs(self.light)[:22], cs(self.n)[:18], cs(a)[:18], cs(r)[:14], cs(self.orig["roughness"])[:14], cs(self.stats)[:22], kw.get("patience"), kw.get("min_delta"),
                    None if kw.get("best_mse") is None else float(kw["best_mse"].reshape(-1)[0])))
        self._first = True


def run(self, n):
    orig_run(self, n)
    if getattr(self, "_first", False) and self.t in (100, 5000):
        log.append((self.t, [float(x) for x in self.hist[:3, 0]], cs(self.hist[: self.t])[:20]))


loop.FusedBrdfPhase.__init__ = init
loop.FusedBrdfPhase.run = run
for k in range(3):
    log.clear()
    args = ri.parse(["--sample", "indoor2", "--model_name", "none", "--out", "/tmp/ri", "--num_epochs", "5000"])
    out = ri.run(args)
    print("run", k, log[:3], flush=True)
