#!/bin/bash
# cycle stamps inside one wave of lazy_resample_kernel (a stamped library is built beside the product one and removed afterwards)
cd "$GRAFT_REPO_ROOT" || exit 1
cp materialist_amd/libmatpbr.so /tmp/lib_keep.so
python - <<'PY'
import subprocess, os
from materialist_amd import build as b
cmd = [b._hipcc(), *b.HIPCC_FLAGS, "-DMATPBR_RS_STAMPS", "-o", b.LIB_PATH, *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
subprocess.run(cmd, check=True)
PY
python - <<'PY'
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from materialist_amd import _lib, loop, render, synthetic
dev = torch.device("cuda:0")
for B in (8, 1):
    scs = [synthetic.make_scene(i, 512, 512) for i in range(B)]
    t = lambda f: (torch.stack([torch.as_tensor(f(s), dtype=torch.float32) for s in scs]) if B > 1 else torch.as_tensor(f(scs[0]), dtype=torch.float32)).to(dev)
    scene = render.load_estimated_mesh(t(lambda s: s.depth), use_mesh_normal=True)
    scene._set("emitter.data", t(lambda s: s.light))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, t(lambda s: s.albedo), t(lambda s: s.roughness), t(lambda s: s.metallic), None, 64)
    ph = loop.FusedBrdfPhase(scene, gt, t(lambda s: s.init_albedo), t(lambda s: s.init_roughness), t(lambda s: s.init_metallic), optimize_part="rm", spp=64)
    ph.run(600)
    torch.cuda.synchronize()
    out = np.zeros(8, dtype=np.uint64)
    fn = _lib.load().matpbr_debug_rs_stamps
    fn.argtypes = [ctypes.c_void_p]
    assert fn(out.ctypes.data) == 0
    s = out.astype(np.int64)
    print("B =", B, "cycles: top loads + scan", s[1] - s[0], "| search + list", s[2] - s[1], "| pixel loads + setup", s[3] - s[2], "| walk", s[4] - s[3], "| fold + stores", s[5] - s[4], "| total", s[5] - s[0])
PY
cp /tmp/lib_keep.so materialist_amd/libmatpbr.so
