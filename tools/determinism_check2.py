"""One none-mode real-image run in this process; argv[1] = 1/0 rotate.  (Fresh-process determinism: run after another GPU job.)"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("real_image", os.path.join(ROOT, "tools", "real_image.py"))
ri = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ri)
from materialist_amd import loop  # noqa: E402

loop.FusedBrdfPhase.ROTATE_BEST = bool(int(sys.argv[1]))
args = ri.parse(["--sample", "indoor2", "--model_name", "none", "--out", "/tmp/ri"])
out = ri.run(args)
print("rotate", sys.argv[1], out["psnr_vs_photo"]["this_build_final_render"], [l.split("] ")[-1][:70] for l in out["log"][1:4]], flush=True)
