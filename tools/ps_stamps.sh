#!/bin/bash
# cycle stamps inside one workgroup of lazy_pstep_kernel (a stamped library is built beside the product one and removed afterwards)
cd "$GRAFT_REPO_ROOT" || exit 1
cp materialist_amd/libmatpbr.so /tmp/lib_keep.so
python - <<'PY'
import subprocess, os
from materialist_amd import build as b
cmd = [b._hipcc(), *b.HIPCC_FLAGS, "-DMATPBR_PS_STAMPS", "-o", b.LIB_PATH, *[os.path.join(b.CSRC, s) for s in b.SOURCES]]
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
    fn = _lib.load().matpbr_debug_ps_stamps
    fn.argtypes = [ctypes.c_void_p]
    for rep in range(12):
        ph.run(7)
        torch.cuda.synchronize()
        out = np.zeros(16, dtype=np.uint64)
        assert fn(out.ctypes.data) == 0
        s = out.astype(np.int64)
        ghz = (s[8] - s[0]) / max(1, (s[13] - s[12]) * 10) 
        d = lambda i, j: int(s[j] - s[i])
        print("B", B, "T", s[15], "clk %.2f GHz" % ghz, "| first loads", d(0, 1), "| head", d(1, 2), "| stream", d(2, 3), "| barrier", d(3, 4),
              "| tables", d(4, 5) if s[15] else 0, "| walk", d(5, 6) if s[15] else 0, "| barrier", d(6, 7) if s[15] else 0, "| final", d(7, 8) if s[15] else d(4, 8), "| total", d(0, 8))
PY
cp /tmp/lib_keep.so materialist_amd/libmatpbr.so
