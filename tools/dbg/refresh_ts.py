import sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
from materialist_amd import ops, synthetic, _lib
dev = torch.device("cuda:0")
_t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
H = W = 512; spp = 64
sc = synthetic.make_scene(0, H, W)
a, r, m, light = _t(sc.albedo), _t(sc.roughness), _t(sc.metallic), _t(sc.light)
n = ops.normals_from_depth(_t(sc.depth))
dcache = ops.diffuse_cache(n, light, spp)
state = ops.lazy_state(a)
ops.shade_fwd_lazy(a, r, m, n, light, spp, dcache, state, force=True, floor=0.3)
lib = _lib.load()
dbg = torch.zeros(16, dtype=torch.int64, device=dev)
lib.matpbr_dbg_set.argtypes = [ctypes.c_void_p]
lib.matpbr_dbg_set(ctypes.c_void_p(dbg.data_ptr()))
g = torch.Generator(device=dev); g.manual_seed(1)
for frac in (0.0, 0.004, 0.02):
    for rep in range(3):
        mask = (torch.rand(r.shape, device=dev, generator=g) < frac).float()
        r2 = (r + 0.05 * mask * (1 if rep % 2 else -1)).clamp(0.07, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.shade_fwd_lazy(a, r2, m, n, light, spp, dcache, state, floor=0.3)
        e1.record(); torch.cuda.synchronize()
        d = dbg.cpu().numpy()
        k = int(d[15])
        print(f"frac {frac} rep {rep}: fwd+refresh {e0.elapsed_time(e1)*1e3:.1f} us; T={d[14]} stamps(us @100MHz): ", [round((d[j]-d[0])/100.0, 2) for j in range(k)])
