#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ by IMPORTING the reference's own
arithmetic (read-only, /root/reference) in the build container.

This script is test infrastructure.  It never travels with meaning to the GPU box (the reference
tree does not exist there); only the .npz fixtures it writes are used by the tests.

The reference needs `mitsuba`, `drjit`, `open3d`, `lovely_tensors`, `cv2`, which are not installable
here.  The shim below injects small torch-backed stand-ins into sys.modules so that the
reference's *arithmetic* (plain Python operators traced over array types) runs on torch.float64 CPU
tensors.  Layout used by the shim: vectors are SoA torch tensors of shape (C, N); scalars are (N,).

What each fixture pins (SURVEY.md §8c "golden vectors to commit"):
  brdf_scalar_grids.npz   D_GGX / G1_GGX_Schlick / G_Smith / fresnelSchlick   myutils/mi_plugin.py:60-97
  eval_brdf.npz           MatDiffBSDF.eval_brdf fwd + autograd grads           myutils/mi_plugin.py:1372-1427
  eval_brdf_kat.npz       App. C hand-picked known answers (same function)
  samplers.npz            mi_diffuse_sampler / mi_specular_sampler            myutils/mi_plugin.py:217-281
  sample_brdf.npz         MatDiffBSDF.sample_brdf (lobe choice + MC weight)    myutils/mi_plugin.py:1296-1341
  world_to_screen.npz     perspective_projection_matrix + mi_world_to_screen   myutils/mi_plugin.py:585-595,645-671
  sh.npz                  computeK, basis via projection(), reconstImageFromSH myutils/computeSH.py:13-68,165-240
  misc.npz                EarlyStopping / SaveBest decisions, gamma           myutils/misc.py:37-111,163-170
  misc_resize.npz         center_crop_and_resize                              myutils/misc.py:10-34
  posmlp.npz              PosMLP forward + gradients for fixed weights        mymodels/mlps.py:129-251
  materialnet.npz         MaterialNet forward for name-seeded weights, Resize sizes Material_net/dpt.py:175-217, util/transform.py:58-100
  envmaps.npz             decoded envmaps/0.hdr and output_imgs/*/best_results/envmap.hdr (data files)

[ext] caveat: `mi.Frame3f(n).to_world` is Mitsuba's `coordinate_system` (Duff et al. 2017 branchless
ONB).  Its source is not in the reference tree; the shim restates the published algorithm, so the
sampler fixtures pin the reference's sampler *given* that frame convention.
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("MATPBR_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

torch.set_default_dtype(torch.float64)


# --------------------------------------------------------------------------------------------------
# shim array type: torch tensor whose 1-D boolean masks index COLUMNS of a (C, N) SoA vector
# --------------------------------------------------------------------------------------------------
class V(torch.Tensor):
    @staticmethod
    def wrap(t):
        return t.as_subclass(V)

    def __getitem__(self, key):
        if isinstance(key, torch.Tensor) and key.dtype == torch.bool and key.ndim == 1 and self.ndim == 2:
            return torch.Tensor.__getitem__(self, (slice(None), key))
        return torch.Tensor.__getitem__(self, key)

    def __setitem__(self, key, value):
        if isinstance(key, torch.Tensor) and key.dtype == torch.bool and key.ndim == 1 and self.ndim == 2:
            if self.shape[1] == 1 and key.numel() != 1:  # lazily materialise Vector3f(0.0)
                self.data = self.data.expand(self.shape[0], key.numel()).clone()
            return torch.Tensor.__setitem__(self, (slice(None), key), value)
        return torch.Tensor.__setitem__(self, key, value)


def _t(x):
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(x, dtype=torch.get_default_dtype())


def _vec(*comps):
    if len(comps) == 1:
        c = _t(comps[0])
        if c.ndim == 0:
            return V.wrap(c.reshape(1, 1).expand(3, 1).clone())
        return V.wrap(c)
    comps = torch.broadcast_tensors(*[_t(c) for c in comps])
    return V.wrap(torch.stack(comps, 0))


class _TensorXf:
    """Stand-in for mi.TensorXf: .array is the flat storage, .shape the logical shape."""

    def __init__(self, data, shape=None):
        if shape is not None:
            data = torch.full(tuple(shape), float(data))
        self.t = _t(data)
        self.array = self.t.reshape(-1)
        self.shape = tuple(self.t.shape)


class _Frame3f:
    """[ext] Mitsuba 3 coordinate_system(n): Duff et al., 'Building an Orthonormal Basis, Revisited'."""

    def __init__(self, n):
        n = _t(n)
        sign = torch.where(n[2] >= 0, torch.ones_like(n[2]), -torch.ones_like(n[2]))
        a = -1.0 / (sign + n[2])
        b = n[0] * n[1] * a
        self.s = _vec(1.0 + sign * n[0] * n[0] * a, sign * b, -sign * n[0])
        self.t = _vec(b, sign + n[1] * n[1] * a, -n[1])
        self.n = V.wrap(n)

    def to_world(self, v):
        return V.wrap(self.s * v[0] + self.t * v[1] + self.n * v[2])


def _install_shim():
    mi = types.ModuleType("mitsuba")

    class BSDF:  # noqa: D401 - dummy base
        def __init__(self, props=None):
            pass

    mi.BSDF = BSDF
    mi.set_variant = lambda *a, **k: None
    mi.register_bsdf = lambda *a, **k: None
    mi.Float = lambda x: _t(x)
    mi.Int = lambda x: _t(x).long()
    mi.UInt32 = lambda x: x
    mi.Vector2f = lambda *c: _vec(*c)
    mi.Vector3f = lambda *c: _vec(*c)
    mi.Vector4f = lambda *c: _vec(*c)
    mi.Normal3f = mi.Vector3f
    mi.Color3f = mi.Vector3f
    mi.Matrix4f = lambda m: _t(np.asarray(m, dtype=np.float64))
    mi.TensorXf = _TensorXf
    mi.Frame3f = _Frame3f
    mi.BSDFFlags = types.SimpleNamespace(SpatiallyVarying=1, DiffuseReflection=2, FrontSide=4, BackSide=8)
    mi.ParamFlags = types.SimpleNamespace(Differentiable=1, NonDifferentiable=2)
    mi.util = types.SimpleNamespace(write_bitmap=lambda *a, **k: None)
    mi.Bitmap = lambda *a, **k: None

    dr = types.ModuleType("drjit")
    dr.floor = torch.floor
    dr.normalize = lambda v: V.wrap(v / torch.sqrt((v * v).sum(0)))
    dr.dot = lambda a, b: (a * b).sum(0).as_subclass(torch.Tensor)
    dr.maximum = lambda a, b: torch.maximum(_t(a), _t(b))
    dr.safe_sqrt = lambda x: torch.sqrt(torch.clamp(_t(x), min=0.0))
    dr.sin, dr.cos, dr.asin = torch.sin, torch.cos, torch.asin
    dr.isnan = torch.isnan
    dr.select = lambda m, a, b: torch.where(m, _t(a), _t(b))

    def gather(dtype, array, index):
        if dtype is mi.Float:
            return array[index]
        return V.wrap(array.reshape(-1, 3)[index].T)

    dr.gather = gather
    dr.wrap_ad = lambda **k: (lambda f: f)
    dr.set_flag = lambda *a, **k: None

    stubs = {"mitsuba": mi, "drjit": dr}
    for name in ("open3d", "lovely_tensors", "cv2", "imageio"):
        m = types.ModuleType(name)
        m.monkey_patch = lambda *a, **k: None
        stubs[name] = m
    sys.modules.update(stubs)
    if not hasattr(np, "math"):
        np.math = math  # computeSH.py uses np.math.factorial (removed in numpy 2)
    sys.path.insert(0, REF)
    return mi, dr


def _rand_unit(g, n, hemi_axis=None):
    v = torch.randn(3, n, generator=g)
    v = v / v.norm(dim=0, keepdim=True)
    if hemi_axis is not None:
        s = torch.sign((v * hemi_axis).sum(0))
        s[s == 0] = 1
        v = v * s
    return v


def _np(x):
    return np.ascontiguousarray(x.detach().as_subclass(torch.Tensor).numpy())


def main():
    sys.path.insert(0, os.path.normpath(os.path.join(OUT, "..", "..")))   # the build's own package (RGBE reader, name-seeded init)
    mi, dr = _install_shim()
    import myutils.mi_plugin as P  # noqa: E402  (reference module, read-only)
    import myutils.computeSH as SH  # noqa: E402
    import myutils.misc as M  # noqa: E402

    g = torch.Generator().manual_seed(20250629)

    # ---------------------------------------------------------------- scalar grids (a1-a3)
    r = torch.linspace(0.07, 1.0, 32)
    c = torch.linspace(0.0, 1.0, 33)
    R, C = torch.meshgrid(r, c, indexing="ij")
    F0 = torch.tensor([0.04, 0.5, 0.9])
    np.savez(
        os.path.join(OUT, "brdf_scalar_grids.npz"),
        r=_np(r), c=_np(c),
        D=_np(P.D_GGX(C, R)),
        G1=_np(P.G1_GGX_Schlick(C, R)),
        Gs=_np(P.G_Smith(C[None].expand(5, -1, -1), torch.linspace(0, 1, 5)[:, None, None].expand(-1, 32, 33), R[None].expand(5, -1, -1))),
        Gs_nol=_np(torch.linspace(0, 1, 5)),
        F0=_np(F0),
        Fr=_np(P.fresnelSchlick(c[None, :], F0[:, None])),
    )

    # ---------------------------------------------------------------- eval_brdf fwd + grads (a4)
    def run_eval(wi, wo, n, a, rr, m, use_mesh_normal=True):
        N = wi.shape[1]
        a = a.clone().requires_grad_(True)
        rr = rr.clone().requires_grad_(True)
        m = m.clone().requires_grad_(True)
        n_ = n.clone().requires_grad_(True)
        # texel gather: lane k reads texel k of an (N,1,C) "image" -> screen x = 0.5, y = k + 0.5
        # (flat_index = x + y * r.shape[0] with r.shape[0] = N ... see note below)
        self = types.SimpleNamespace(use_mesh_normal=use_mesh_normal)
        self.a = types.SimpleNamespace(array=a.T.reshape(-1), shape=(1, N, 3))
        self.r = types.SimpleNamespace(array=rr, shape=(1, N, 1))
        self.m = types.SimpleNamespace(array=m, shape=(1, N, 1))
        self.n = types.SimpleNamespace(array=n_.T.reshape(-1), shape=(1, N, 3))
        # flat_index = floor(sx) + floor(sy) * r.shape[0]; with shape[0] == 1 and sy = 0 the index is floor(sx)
        screen = _vec(torch.arange(N, dtype=torch.float64) + 0.5, torch.zeros(N))
        f, pdf = P.MatDiffBSDF.eval_brdf(self, V.wrap(wi), V.wrap(wo), V.wrap(n_), None, screen)
        f = f.as_subclass(torch.Tensor)
        pdf = pdf.as_subclass(torch.Tensor)
        ga, gr, gm, gn = torch.autograd.grad(f.sum(), [a, rr, m, n_], retain_graph=True, allow_unused=True)
        # per-channel gradients as well (needed for an RGB-weighted backward)
        per_ch = []
        for ch in range(3):
            per_ch.append(torch.autograd.grad(f[ch].sum(), [a, rr, m, n_], retain_graph=True, allow_unused=True))
        gpa, gpr, gpm, gpn = torch.autograd.grad(pdf.sum(), [a, rr, m, n_], allow_unused=True)
        z = lambda x, like: torch.zeros_like(like) if x is None else x
        out = dict(
            f=_np(f), pdf=_np(pdf), d_a=_np(ga), d_r=_np(gr), d_m=_np(gm), d_n=_np(z(gn, n)),
            dpdf_r=_np(z(gpr, rr)), dpdf_n=_np(z(gpn, n)),
        )
        for ch in range(3):
            out[f"d_r_ch{ch}"] = _np(per_ch[ch][1])
            out[f"d_m_ch{ch}"] = _np(per_ch[ch][2])
            out[f"d_n_ch{ch}"] = _np(z(per_ch[ch][3], n))
            out[f"d_a_ch{ch}"] = _np(per_ch[ch][0])
        return out

    N = 4096
    n = _rand_unit(g, N)
    wo = _rand_unit(g, N)
    wi = _rand_unit(g, N)
    # make ~80% of the lanes front-facing for both directions, keep the rest arbitrary (zero / clamp cases)
    front = torch.rand(N, generator=g) < 0.8
    flip_o = torch.sign((wo * n).sum(0)); flip_o[flip_o == 0] = 1
    flip_i = torch.sign((wi * n).sum(0)); flip_i[flip_i == 0] = 1
    wo = torch.where(front, wo * flip_o, wo)
    wi = torch.where(front, wi * flip_i, wi)
    a = torch.rand(3, N, generator=g)
    rr = 0.07 + 0.93 * torch.rand(N, generator=g)
    rr[:256] = 0.07 + 0.05 * torch.rand(256, generator=g)  # sharp lobes
    m = torch.rand(N, generator=g)
    m[::7] = 0.0
    m[3::11] = 1.0
    res = run_eval(wi, wo, n, a, rr, m, use_mesh_normal=True)
    np.savez(os.path.join(OUT, "eval_brdf.npz"), wi=_np(wi), wo=_np(wo), n=_np(n), a=_np(a), r=_np(rr), m=_np(m), **res)

    # App. C known answers
    def nrm(v):
        v = torch.tensor(v, dtype=torch.float64)
        return v / v.norm()

    kat = [
        (0.5, 0.0, (0.6, 0, 0.8), (0, 0, 1)),
        (0.5, 1.0, (0.6, 0, 0.8), (0, 0, 1)),
        (0.07, 0.3, (0.3, 0.2, 0.9), (-0.3, -0.1, 0.9)),
        (1.0, 0.5, (0, 0.8, 0.6), (0.5, 0, 0.5)),
        (0.3, 0.5, (0, 0.8, -0.1), (0.5, 0, 0.5)),
    ]
    K = len(kat)
    k_wi = torch.stack([nrm(k[2]) for k in kat], 1)
    k_wo = torch.stack([nrm(k[3]) for k in kat], 1)
    k_n = torch.tensor([[0.0, 0.0, 1.0]] * K).T.contiguous()
    k_a = torch.tensor([[0.8, 0.5, 0.2]] * K).T.contiguous()
    k_r = torch.tensor([k[0] for k in kat])
    k_m = torch.tensor([k[1] for k in kat])
    res = run_eval(k_wi, k_wo, k_n, k_a, k_r, k_m)
    np.savez(os.path.join(OUT, "eval_brdf_kat.npz"), wi=_np(k_wi), wo=_np(k_wo), n=_np(k_n), a=_np(k_a), r=_np(k_r), m=_np(k_m), **res)

    # ---------------------------------------------------------------- samplers (a5)
    u0, u1 = torch.meshgrid((torch.arange(8) + 0.5) / 8, (torch.arange(8) + 0.5) / 8, indexing="ij")
    u = torch.stack([u0.reshape(-1), u1.reshape(-1)], 0)  # (2, 64)
    S = u.shape[1]
    normals = torch.stack([nrm((0, 0, 1)), nrm((0.3, -0.2, 0.9)), nrm((0.5, 0.5, -0.7)), nrm((-0.9, 0.1, 0.1))], 1)
    views = torch.stack([nrm((0, 0, 1)), nrm((0.6, 0.1, 0.7)), nrm((0.2, 0.9, -0.3)), nrm((-0.5, 0.4, 0.6))], 1)
    rough = torch.tensor([0.07, 0.3, 1.0])
    diff = []
    spec = []
    for k in range(normals.shape[1]):
        nk = normals[:, k:k + 1].expand(3, S)
        diff.append(_np(P.mi_diffuse_sampler(V.wrap(u), V.wrap(nk))))
        row = []
        for rv in rough:
            vk = views[:, k:k + 1].expand(3, S)
            row.append(_np(P.mi_specular_sampler(V.wrap(u), rv.expand(S), V.wrap(vk), V.wrap(nk))))
        spec.append(np.stack(row))
    np.savez(os.path.join(OUT, "samplers.npz"), u=_np(u), normals=_np(normals), views=_np(views), rough=_np(rough),
             diffuse=np.stack(diff), specular=np.stack(spec))

    # ---------------------------------------------------------------- sample_brdf (lobe choice + weight)
    Ns = 1024
    s_n = _rand_unit(g, Ns)
    s_wo = _rand_unit(g, Ns, hemi_axis=s_n)
    s_a = torch.rand(3, Ns, generator=g)
    s_r = 0.07 + 0.93 * torch.rand(Ns, generator=g)
    s_m = torch.rand(Ns, generator=g)
    s1 = torch.rand(Ns, generator=g)
    s2 = torch.rand(2, Ns, generator=g)
    self = types.SimpleNamespace(use_mesh_normal=True)
    self.a = types.SimpleNamespace(array=s_a.T.reshape(-1), shape=(1, Ns, 3))
    self.r = types.SimpleNamespace(array=s_r, shape=(1, Ns, 1))
    self.m = types.SimpleNamespace(array=s_m, shape=(1, Ns, 1))
    self.n = types.SimpleNamespace(array=s_n.T.reshape(-1), shape=(1, Ns, 3))
    self.eval_brdf = lambda *args: P.MatDiffBSDF.eval_brdf(self, *args)
    screen = _vec(torch.arange(Ns, dtype=torch.float64) + 0.5, torch.zeros(Ns))
    wi_s, pdf_s, w_s = P.MatDiffBSDF.sample_brdf(self, s1, V.wrap(s2), V.wrap(s_wo), V.wrap(s_n), None, screen)
    np.savez(os.path.join(OUT, "sample_brdf.npz"), n=_np(s_n), wo=_np(s_wo), a=_np(s_a), r=_np(s_r), m=_np(s_m),
             sample1=_np(s1), sample2=_np(s2), wi=_np(wi_s), pdf=_np(pdf_s), weight=_np(w_s))

    # ---------------------------------------------------------------- world -> screen (a6)
    cam = json.load(open(os.path.join(REF, "myutils", "default_cam.json")))
    to_world = torch.tensor(cam["to_world"])[0]
    view = torch.inverse(to_world)
    W, H = cam["film.size"]
    fov = torch.deg2rad(torch.tensor(cam["x_fov"][0]))
    proj = P.perspective_projection_matrix(fov, W / H, cam["near_clip"], cam["far_clip"]).double()
    ii, jj = torch.meshgrid(torch.arange(0, 512, 37), torch.arange(0, 512, 41), indexing="ij")
    ii = ii.reshape(-1).double(); jj = jj.reshape(-1).double()
    depth = 1.5 + 2.0 * torch.rand(ii.numel(), generator=g)
    f_px = (W / 2) / math.tan(math.radians(35) / 2)
    cx = (W - 1) / 2
    pts = torch.stack([(jj - cx) / f_px * depth, -(ii - cx) / f_px * depth, -depth], 0)
    sc = P.mi_world_to_screen(V.wrap(pts), view, proj, W, H)
    np.savez(os.path.join(OUT, "world_to_screen.npz"), row=_np(ii), col=_np(jj), depth=_np(depth), points=_np(pts),
             screen=_np(sc), proj=_np(proj), view=_np(view), fov_deg=35.0, width=W, height=H)

    # ---------------------------------------------------------------- SH (a10)
    larr = np.array([0, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4], dtype=np.int32)
    marr = np.array([0, -1, 0, 1, -2, -1, 0, 1, 2, -3, -2, -1, 0, 1, 2, 3, -4, -3, -2, -1, 0, 1, 2, 3, 4], dtype=np.int32)
    Ksh = SH.computeK(larr.copy(), marr.copy())
    rng = np.random.default_rng(7)
    theta = np.concatenate([np.linspace(0, np.pi, 9), rng.uniform(0, np.pi, 55)])
    phi = np.concatenate([np.linspace(-np.pi, np.pi, 9), rng.uniform(-np.pi, np.pi, 55)])
    basis = np.stack([SH.projection(phi, theta, Ksh, np.eye(25)[:, k:k + 1])[:, 0] for k in range(25)], 1)  # (64, 25)
    coef = np.zeros((25, 3))
    coef[0] = 1.0
    coef[2] = (0.5, 0.2, 0.1)
    coef[3] = (0.1, 0.0, -0.2)
    img_c = SH.reconstImageFromSH(coef, 16, 32, isClip=False)
    coef_r = rng.normal(0, 0.3, (25, 3)) / (1 + larr[:, None]) ** 2
    coef_r[0] = 3.0
    img_r = SH.reconstImageFromSH(coef_r, 16, 32, isClip=False)
    np.savez(os.path.join(OUT, "sh.npz"), l=larr, m=marr, K=Ksh.astype(np.float64), theta=theta, phi=phi, basis=basis,
             coef_c=coef, img_c=img_c, coef_r=coef_r, img_r=img_r)

    # ---------------------------------------------------------------- misc (a11 gamma, a13 SaveBest/EarlyStopping)
    seqs = {
        "env": [1.0, 0.98, 0.985, 0.99, 0.97, 0.9699, 0.9698, 0.9697, 0.95, 0.96, 0.961, 0.962],
        "flat": [0.5] * 8,
        "rise": [0.3, 0.31, 0.32, 0.2, 0.25, 0.26, 0.27, 0.28],
    }
    es_out = {}
    for name, seq in seqs.items():
        for pat, delta in ((3, 0.01), (2, 0.025), (4, 0.001)):
            es = M.EarlyStopping(patience=pat, min_delta=delta)
            tr = []
            for v in seq:
                es(v)
                tr.append((es.counter, es.best_loss, float(es.early_stop)))
            es_out[f"es_{name}_{pat}_{delta}"] = np.array(tr)
        sb = M.SaveBest()
        keep = []
        for k, v in enumerate(seq):
            sb.update(v, torch.full((1,), float(k)), None, None, None, None, None)
            keep.append((sb.best_loss, float(sb.best_albedo[0])))
        es_out[f"sb_{name}"] = np.array(keep)
        es_out[f"seq_{name}"] = np.array(seq)
    x = torch.linspace(0, 2, 41)
    es_out["gamma_x"] = _np(x)
    es_out["linear_to_srgb"] = _np(M.linear_to_srgb(x))
    es_out["srgb_to_linear"] = _np(M.srgb_to_linear(x))
    np.savez(os.path.join(OUT, "misc.npz"), **es_out)

    # ---------------------------------------------------------------- PosMLP (f2, mymodels/mlps.py:129-251)
    import mymodels.mlps as ML  # noqa: E402

    pos = {}
    for tag, kw, npts, cin in (("env", dict(in_dims=5, out_dims=3, multires_view=2, output_type="envmap", color_ch=3), 512, 3),
                               ("arm", dict(in_dims=7, out_dims=5, multires_view=2, output_type="arm", color_ch=5), 576, 5),
                               ("armn", dict(in_dims=10, out_dims=8, multires_view=0, output_type="armn", color_ch=8), 576, 8)):
        torch.manual_seed(hash(tag) % 1000 if False else {"env": 1, "arm": 2, "armn": 3}[tag])
        net = ML.PosMLP(dims=[64] * 4, skip_connection=[1, 3], weight_norm=False, **kw)   # width 64 keeps the fixture small
        net.lin4.weight.data.normal_(0, 0.05)      # the zero-initialised last layer would make the test vacuous
        net.lin4.bias.data.normal_(0, 0.05)
        img_in = torch.rand(npts, cin, generator=g).requires_grad_(True)
        out = net(img_in)
        wgt = torch.randn(out.shape, generator=g)
        (out * wgt).sum().backward()
        for k_, v_ in net.state_dict().items():
            pos[f"{tag}.sd.{k_}"] = _np(v_)
        pos[f"{tag}.in"], pos[f"{tag}.out"], pos[f"{tag}.w"] = _np(img_in), _np(out), _np(wgt)
        pos[f"{tag}.d_in"] = _np(img_in.grad)
        pos[f"{tag}.d_lin0_w"] = _np(net.lin0.linear.weight.grad)
        pos[f"{tag}.d_lin4_b"] = _np(net.lin4.bias.grad)
    np.savez_compressed(os.path.join(OUT, "posmlp.npz"), **pos)

    # ---------------------------------------------------------------- MaterialNet (f3, Material_net/dpt.py:175-217)
    # 108 M parameters cannot be committed: every tensor of the state_dict is (re)initialised from its NAME
    # (materialist_amd.materialnet.init_from_names), so the test rebuilds the same weights and only inputs/outputs are stored.
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.Compose = lambda fns: (lambda s: [s := f(s) for f in fns][-1])
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt})
    sys.modules["cv2"].INTER_AREA, sys.modules["cv2"].INTER_CUBIC, sys.modules["cv2"].INTER_NEAREST = 3, 2, 0
    from Material_net.dpt import MaterialNet as RefNet  # noqa: E402
    from Material_net.util.transform import Resize as RefResize  # noqa: E402
    from materialist_amd.materialnet import init_from_names  # noqa: E402

    ref = RefNet(encoder="vitb", features=128, out_channels=[96, 192, 384, 768], use_bn=False, use_clstoken=False).double().eval()
    init_from_names(ref)
    x_in = torch.rand(1, 3, 70, 98, generator=g)
    with torch.no_grad():
        out_ref = ref(x_in)
        feats = ref.pretrained.get_intermediate_layers(x_in, [2, 5, 8, 11], return_class_token=True)
    rs = RefResize(width=518, height=518, resize_target=False, keep_aspect_ratio=True, ensure_multiple_of=14, resize_method="lower_bound",
                   image_interpolation_method=2)
    sizes_in = [(512, 512), (427, 423), (384, 512), (1024, 1024), (640, 480), (100, 300)]
    np.savez_compressed(os.path.join(OUT, "materialnet.npz"), x=_np(x_in), names=np.array(sorted(ref.state_dict().keys())),
                        n_params=sum(p.numel() for p in ref.parameters()),
                        feat3_patch=_np(feats[3][0]), feat0_cls=_np(feats[0][1]),
                        sizes_in=np.array(sizes_in), sizes_out=np.array([rs.get_size(w, h) for (w, h) in sizes_in]),
                        **{k: _np(v) for k, v in out_ref.items()})
    del ref

    # ---------------------------------------------------------------- center_crop_and_resize (pipeline head, misc.py:10-34)
    rr = np.random.default_rng(11)
    im_u8 = rr.integers(0, 256, (37, 53, 4), dtype=np.uint8)      # RGBA like examples/indoor1.png: alpha dropped, centre crop
    im_f32 = rr.random((41, 29, 3), dtype=np.float32)
    np.savez(os.path.join(OUT, "misc_resize.npz"), im_u8=im_u8, out_u8=M.center_crop_and_resize(im_u8, (16, 16)),
             im_f32=im_f32, out_f32=M.center_crop_and_resize(im_f32, (24, 24)))

    # ---------------------------------------------------------------- envmap data files (RGBE decode is ours)
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from materialist_amd.imageio_hdr import read_hdr  # build's own RGBE reader

    np.savez(
        os.path.join(OUT, "envmaps.npz"),
        env0=read_hdr(os.path.join(REF, "envmaps", "0.hdr")),
        indoor=read_hdr(os.path.join(REF, "output_imgs", "indoor", "best_results", "envmap.hdr")),
        jinjya=read_hdr(os.path.join(REF, "output_imgs", "jinjya", "best_results", "envmap.hdr")),
    )
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
