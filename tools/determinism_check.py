import sys, os, json, subprocess
sys.path.insert(0, "/root/repo")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import importlib.util
spec = importlib.util.spec_from_file_location("real_image", "tools/real_image.py")
ri = importlib.util.module_from_spec(spec); spec.loader.exec_module(ri)
from materialist_amd import loop
for rot in (False, True, True, True):
    loop.FusedBrdfPhase.ROTATE_BEST = rot
    args = ri.parse(["--sample", "indoor2", "--model_name", "none", "--out", "/tmp/ri"])
    args.out = "/tmp/ri"
    out = ri.run(args)
    print("rotate", rot, out["psnr_vs_photo"]["this_build_final_render"], [l.split("] ")[-1] for l in out["log"][1:]], flush=True)
