#!/bin/bash
# cycle stamps inside one wave of lazy_pwalk_kernel (group 5, wave 0)
cd "$GRAFT_REPO_ROOT" || exit 1
cp materialist_amd/libmatpbr.so /tmp/lib_keep.so
python - <<'PY'
import subprocess, os
from materialist_amd import build as b
cmd = [b._hipcc(), *b.HIPCC_FLAGS, "-DMATPBR_RS_STAMPS", "-DMATPBR_RS_BLOCK=20", "-o", b.LIB_PATH, *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
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
    ph.run(300)
    fn = _lib.load().matpbr_debug_rs_stamps
    fn.argtypes = [ctypes.c_void_p]
    prev = None
    for rep in range(40):
        ph.run(1)
        torch.cuda.synchronize()
        out = np.zeros(8, dtype=np.uint64)
        assert fn(out.ctypes.data) == 0
        s = out.astype(np.int64)
        if prev is not None and s[6] != prev[6] and s[6] > s[0] > 0 and all(s[i + 1] >= s[i] for i in range(6)):
            d = lambda i, j: int(s[j] - s[i])
            print("B", B, "| counts+state", d(0, 1), "| tables+list", d(1, 2), "| pixel loads+setup", d(2, 3), "| walk", d(3, 4), "| fold+stores", d(4, 5), "| atomics/end", d(5, 6), "| total", d(0, 6))
        prev = s
PY
cp /tmp/lib_keep.so materialist_amd/libmatpbr.so
