"""GPU parity tests: the HIP path (through the C ABI of libmatpbr.so) against the fp64 CPU oracle and the
committed golden vectors.  Tolerance: north_star asks for 1e-3 relative fp32; the assertion used is
|hip - oracle| <= RTOL * max(|oracle|, scale) with scale = mean |oracle| of the tensor, so that
near-zero entries of a gradient map are compared on the tensor's own scale."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

RTOL = 1e-3


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    torch.manual_seed(20250629)   # every test draws its random tensors from a fixed stream
    return torch.device("cuda:0")


def _t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)


def assert_close(got, ref, rtol=RTOL, what=""):
    got = got.detach().cpu().numpy().astype(np.float64) if hasattr(got, "detach") else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64).reshape(got.shape)
    scale = np.abs(ref).mean()
    err = np.abs(got - ref) / np.maximum(np.abs(ref), scale + 1e-30)
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    _report(what, err.max(), rtol)
    assert err.max() <= rtol, f"{what}: max scaled rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


def _report(what, measured, bound, floor=False):
    """MATPBR_TOLERANCE_REPORT=<file>: one line per comparison -- the test that made it (pytest's own id: the same wording in two tests can no
    longer collide), what was compared, the measured value beside its bound ("<=" a maximum, ">=" a floor) and whether it held (how the bounds
    in this file were set: DESIGN.md section 5)."""
    path = os.environ.get("MATPBR_TOLERANCE_REPORT")
    if path:
        test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::", 1)[-1]
        ok = float(measured) >= float(bound) if floor else float(measured) <= float(bound)
        with open(path, "a") as f:
            f.write(f"{test}\t{what}\t{float(measured):.3e}\t{'>=' if floor else '<='}\t{float(bound):.1e}\t{'ok' if ok else 'EXCEEDED'}\n")


def _scene_arrays(H, W, image_id=0, unit_random_normals=True):
    from materialist_amd import synthetic
    from oracle.oracle import Oracle

    sc = synthetic.make_scene(image_id, H, W)
    n = Oracle(np.float64).normals_from_depth(sc.depth.astype(np.float64))
    if unit_random_normals:  # perturb so that the shading frame varies strongly across pixels
        rng = np.random.default_rng(99 + image_id)
        n = n + 0.35 * rng.normal(size=n.shape)
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
    return sc, n.astype(np.float32)


# ------------------------------------------------------------------------------------------------ a4
@pytest.mark.parametrize("name", ["eval_brdf.npz", "eval_brdf_kat.npz"])
def test_eval_brdf_matches_reference_golden(golden_dir, name):
    from materialist_amd import ops

    dev = _cuda()
    g = np.load(os.path.join(golden_dir, name))
    wi, wo, n, a = (_t(g[k].T, dev) for k in ("wi", "wo", "n", "a"))
    r, m = _t(g["r"], dev), _t(g["m"], dev)
    f, pdf = ops.eval_brdf(wi, wo, n, a, r, m)
    assert_close(f, g["f"].T, what="f*cos")
    assert_close(pdf, g["pdf"], what="pdf")
    ones = torch.ones_like(f)
    d_a, d_r, d_m, d_n = ops.eval_brdf_bwd(wi, wo, n, a, r, m, ones)
    assert_close(d_a, g["d_a"].T, what="d_a")
    assert_close(d_m, g["d_m"], what="d_m")
    # d_r / d_n carry the GGX peak (D ~ 1e4 at r = 0.07).  NoH^2(alpha2-1)+1 is ill-conditioned there in fp32 arithmetic AND in
    # the fp32 inputs; the N-lane kernels form 1 - NoH^2 as |n x h|^2 for unit normals, which is neither: 1e-3 on every lane
    for got, ref, nm in ((d_r, g["d_r"], "d_r"), (d_n, g["d_n"].T, "d_n")):
        got = got.cpu().numpy()
        scale = np.abs(ref).mean()
        err = np.abs(got - ref) / np.maximum(np.abs(ref), scale)
        err_rows = err if err.ndim == 1 else err.max(1)
        assert err_rows.max() <= RTOL, f"{nm}: {err_rows.max():.3e} (r of the worst lane {g['r'][err_rows.argmax()]:.3f})"
    for ch in range(3):
        e = torch.zeros_like(f)
        e[:, ch] = 1.0
        d_a, d_r, d_m, d_n = ops.eval_brdf_bwd(wi, wo, n, a, r, m, e)
        assert_close(d_a, g[f"d_a_ch{ch}"].T, what=f"d_a ch{ch}")
        assert_close(d_m, g[f"d_m_ch{ch}"], what=f"d_m ch{ch}")


# ------------------------------------------------------------------------------------------------ a1-a3
def test_brdf_terms_match_reference_grids(golden_dir):
    from materialist_amd import ops

    dev = _cuda()
    g = np.load(os.path.join(golden_dir, "brdf_scalar_grids.npz"))
    R, C = np.meshgrid(g["r"], g["c"], indexing="ij")
    out = ops.brdf_terms(_t(C.reshape(-1), dev), _t(np.full(C.size, 0.5), dev), _t(R.reshape(-1), dev), _t(np.full(C.size, 0.04), dev)).cpu().numpy()
    # D_GGX near cos = 1 at small r is the ill-conditioned literal form (condition number 1/alpha^2): measured 8.5e-4 on the peak (round 5;
    # bound 2e-2 until then), an order below it elsewhere -- 1e-3 everywhere
    D = g["D"].reshape(-1)
    peak = (R.reshape(-1) < 0.15) & (C.reshape(-1) > 0.95)
    errD = np.abs(out[:, 0] - D) / np.maximum(np.abs(D), 1e-12)
    _report("D_GGX away from the peak", errD[~peak].max(), 1e-3)
    _report("D_GGX on the peak (r < 0.15, cos > 0.95)", errD[peak].max(), 1e-3)
    assert errD[~peak].max() < 1e-3 and errD[peak].max() < 1e-3, (errD[~peak].max(), errD[peak].max())
    np.testing.assert_allclose(out[:, 1], g["G1"].reshape(-1), rtol=1e-5)
    np.testing.assert_allclose(out[:, 2], g["Gs"][2].reshape(-1), rtol=1e-5)       # Gs_nol[2] = 0.5
    fr = ops.brdf_terms(_t(g["c"], dev), _t(g["c"], dev), _t(np.full(33, 0.5), dev), _t(np.full(33, 0.5), dev)).cpu().numpy()
    np.testing.assert_allclose(fr[:, 3], g["Fr"][1], rtol=1e-5)                     # F0 = 0.5


# ------------------------------------------------------------------------------------------------ a5
def test_sample_brdf_matches_reference_golden(golden_dir):
    from materialist_amd import ops

    dev = _cuda()
    g = np.load(os.path.join(golden_dir, "sample_brdf.npz"))
    wi, pdf, w = ops.sample_brdf(_t(g["sample1"], dev), _t(g["sample2"].T, dev), _t(g["wo"].T, dev), _t(g["n"].T, dev),
                                 _t(g["a"].T, dev), _t(g["r"], dev), _t(g["m"], dev))
    assert np.abs(wi.cpu().numpy() - g["wi"].T).max() < 2e-5
    assert_close(w, g["weight"].T, rtol=1e-4, what="MC weight")            # measured 6.2e-6
    assert_close(pdf, g["pdf"], rtol=1e-3, what="pdf")                     # measured 6.1e-5


def test_attached_sampling_derivative_matches_the_reference_autograd(golden_dir):
    """a5, the live reference's gradient convention at the plugin face: d/dr of sample_brdf THROUGH the sampled direction and the
    pdf (myutils/mi_plugin.py:227-230,1335-1341).  matpbr_sample_brdf_dr (forward-mode derivatives, fp32) against the reference's
    own torch autograd, recorded lane by lane in tests/golden/sample_brdf_grad.npz (1024 lanes, half cosine lobe, half GGX)."""
    from materialist_amd import ops

    dev = _cuda()
    g = np.load(os.path.join(golden_dir, "sample_brdf.npz"))
    gg = np.load(os.path.join(golden_dir, "sample_brdf_grad.npz"))
    d_wi, d_pdf, d_w = ops.sample_brdf_dr(_t(g["sample1"], dev), _t(g["sample2"].T, dev), _t(g["wo"].T, dev), _t(g["n"].T, dev),
                                          _t(g["a"].T, dev), _t(g["r"], dev), _t(g["m"], dev))
    d_wi, d_pdf, d_w = d_wi.cpu().numpy().astype(np.float64), d_pdf.cpu().numpy().astype(np.float64), d_w.cpu().numpy().astype(np.float64)
    ref_wi, ref_pdf, ref_w = gg["dwi_dr"].T, gg["dpdf_dr"], gg["dweight_dr"].T
    ok = (g["pdf"] > 2e-6) & (g["r"] > 0.0701) & (g["r"] < 0.9999)        # away from the pdf > 1e-6 mask (:1338) and the clamp ends of r
    assert ok.mean() > 0.9
    diffuse = g["sample1"] > 0.5
    assert np.abs(d_wi[diffuse]).max() == 0.0                              # only GGX-sampled directions move with r
    assert np.abs(d_wi[ok] - ref_wi[ok]).max() <= 1e-3 * max(np.abs(ref_wi[ok]).max(), 1.0)
    for got, ref, nm in ((d_pdf, ref_pdf, "d pdf / d r"), (d_w, ref_w, "d weight / d r")):
        scale = np.abs(ref[ok]).mean()
        err = np.abs(got[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), scale)
        _report(nm, err.max(), 1e-3)
        assert err.max() <= 1e-3, f"{nm}: {err.max():.3e}"               # measured: d pdf / d r 8.0e-4 (fp32 forward mode against the reference's fp32 autograd), d weight / d r 5.0e-5
    # and it is not the detached derivative: on GGX lanes the two differ visibly
    assert np.abs(ref_wi[ok & ~diffuse]).max() > 0.1


def test_samplers_match_reference_golden(golden_dir):
    from materialist_amd import ops

    dev = _cuda()
    g = np.load(os.path.join(golden_dir, "samplers.npz"))
    u = g["u"]
    S = u.shape[1]
    for k in range(g["normals"].shape[1]):
        n = _t(np.repeat(g["normals"][:, k][None], S, 0), dev)
        wo = _t(np.repeat(g["views"][:, k][None], S, 0), dev)
        a = torch.full((S, 3), 0.5, device=dev)
        m = torch.full((S,), 0.5, device=dev)
        wi, _, _ = ops.sample_brdf(torch.ones(S, device=dev), _t(u.T, dev), wo, n, a, torch.full((S,), 0.5, device=dev), m)
        assert np.abs(wi.cpu().numpy() - g["diffuse"][k].T).max() < 1e-5
        for ri, rv in enumerate(g["rough"]):
            wi, _, _ = ops.sample_brdf(torch.zeros(S, device=dev), _t(u.T, dev), wo, n, a, torch.full((S,), float(rv), device=dev), m)
            assert np.abs(wi.cpu().numpy() - g["specular"][k][ri].T).max() < 2e-5


# ------------------------------------------------------------------------------------------------ a10
def test_sh_eval_matches_oracle_and_reference_grid(golden_dir, oracle64):
    from materialist_amd import ops

    dev = _cuda()
    g = np.load(os.path.join(golden_dir, "sh.npz"))
    rows, cols = np.meshgrid(np.arange(16), np.arange(32), indexing="ij")
    th = np.pi * rows.reshape(-1) / 16
    ph = -np.pi + 2 * np.pi * cols.reshape(-1) / 32
    w = np.stack([np.sin(th) * np.sin(ph), np.cos(th), -np.sin(th) * np.cos(ph)], -1)
    L = ops.sh_eval(_t(w, dev), _t(g["coef_r"], dev))
    assert_close(L, g["img_r"].reshape(-1, 3), what="reconstImageFromSH grid")
    rng = np.random.default_rng(3)
    w = rng.normal(size=(1000, 3))
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    L = ops.sh_eval(_t(w, dev), _t(g["coef_r"], dev))
    assert_close(L, oracle64.sh_basis_dir(w) @ g["coef_r"], what="sh_eval")


# ------------------------------------------------------------------------------------------------ a9
def test_normals_from_depth(oracle64):
    from materialist_amd import ops, synthetic

    dev = _cuda()
    sc = synthetic.make_scene(3, 96, 80)
    n = ops.normals_from_depth(_t(sc.depth, dev))
    ref = oracle64.normals_from_depth(sc.depth.astype(np.float64))
    assert np.abs(n.cpu().numpy() - ref).max() < 2e-4
    nb = ops.normals_from_depth(_t(np.stack([sc.depth, sc.depth * 1.5]), dev))
    assert torch.equal(nb[0], n)


# ------------------------------------------------------------------------------------------------ a8
@pytest.mark.parametrize("H,W,spp", [(64, 64, 2), (48, 80, 8), (64, 64, 64), (33, 37, 128)])
def test_shade_fwd_matches_oracle(oracle64, H, W, spp):
    from materialist_amd import ops

    dev = _cuda()
    sc, n = _scene_arrays(H, W)
    ref = oracle64.shade_fwd(sc.albedo, sc.roughness, sc.metallic, n, sc.light, spp)
    out = ops.shade_fwd(_t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), _t(n, dev), _t(sc.light, dev), spp)
    assert_close(out, ref, what=f"shade_fwd {H}x{W} spp{spp}")


@pytest.mark.parametrize("spp", [2, 16, 64])
def test_shade_bwd_matches_oracle(oracle64, spp):
    from materialist_amd import ops

    dev = _cuda()
    H, W = 40, 56
    sc, n = _scene_arrays(H, W, image_id=1)
    rng = np.random.default_rng(5)
    d_out = rng.normal(size=(H, W, 3)).astype(np.float32)
    ref = oracle64.shade_bwd(sc.albedo, sc.roughness, sc.metallic, n, sc.light, d_out, spp)
    args = [_t(x, dev) for x in (sc.albedo, sc.roughness, sc.metallic, n, sc.light, d_out)]
    names = ("d_a", "d_r", "d_m", "d_n", "d_light")
    got_all = ops.shade_bwd(*args, spp, want_mat=True, want_n=True, want_light=True)
    for nm, got, rf in zip(names, got_all, ref):
        assert_close(got, rf, rtol=RTOL, what=f"{nm} spp{spp}")              # d_r, d_n measured <= 3.9e-5 (2e-3 until round 5)
    # every flag combination runs its own kernel instantiation: same numbers as the all-on variant
    for want_mat, want_n, want_light in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1)]:
        got = ops.shade_bwd(*args, spp, want_mat=bool(want_mat), want_n=bool(want_n), want_light=bool(want_light))
        want = (want_mat, want_mat, want_mat, want_n, want_light)
        for nm, g_, full, w_ in zip(names, got, got_all, want):
            if w_:
                assert_close(g_, full.cpu().numpy(), rtol=1e-5, what=f"{nm} flags {want_mat}{want_n}{want_light}")
            else:
                assert g_ is None


@pytest.mark.parametrize("spp", [64, 16])
def test_attached_sampling_flag_gives_the_exact_r_derivative_of_the_render(oracle64, spp):
    """a5 at image level, MATPBR_FLAG_ATTACHED_SAMPLING of matpbr_shade_bwd: d_r becomes the derivative of the rendered value
    through the GGX quadrature nodes (the live reference's convention, mi_plugin.py:227-230,1335-1341).  Checked against central
    differences of the fp64 oracle render in r (every pixel perturbed at once: pixels are independent); d_a and d_m are untouched
    by the flag; and the attached and the default (stop-gradient) d_r are two estimates of the same dI/dr: cosine similarity
    reported and bounded."""
    from materialist_amd import ops

    dev = _cuda()
    H, W = 40, 56
    sc, n = _scene_arrays(H, W, image_id=2)
    rng = np.random.default_rng(11)
    d_out = rng.normal(size=(H, W, 3)).astype(np.float32)
    rgh = np.clip(sc.roughness, 0.08, 0.98)
    h = 1e-5
    up = oracle64.shade_fwd(sc.albedo, rgh + h, sc.metallic, n, sc.light, spp)
    dn = oracle64.shade_fwd(sc.albedo, rgh - h, sc.metallic, n, sc.light, spp)
    ref = (((up - dn) / (2 * h)) * d_out).sum(-1, keepdims=True)
    args = [_t(x, dev) for x in (sc.albedo, rgh, sc.metallic, n, sc.light, d_out)]
    d_a0, d_r0, d_m0, _, _ = ops.shade_bwd(*args, spp)
    d_a1, d_r1, d_m1, _, _ = ops.shade_bwd(*args, spp, attached=True)
    assert torch.equal(d_a0, d_a1) and torch.equal(d_m0, d_m1)
    # measured 1.37e-3 at either sample count: one pixel whose GGX sample sits next to the pdf > 1e-6 mask of :1338, where the fp32 forward-mode
    # derivative and the fp64 oracle's differ in which side a sample falls; the 99.9th percentile is below 1e-4
    assert_close(d_r1, ref, rtol=2e-3, what=f"attached d_r spp{spp}")
    a_, d_ = d_r1.double().flatten(), d_r0.double().flatten()
    cos = float((a_ * d_).sum() / (a_.norm() * d_.norm()))
    print(f"attached vs detached d_r at spp {spp}: cosine {cos:.4f}, norm ratio {float(a_.norm() / d_.norm()):.4f}")
    assert 0.5 < cos < 0.99999                              # related, and not the same thing
    err_detached = np.abs(d_r0.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err_detached > 1e-2                              # the default is NOT the derivative of the rendered value
    # the operator face: render.ATTACHED_SAMPLING switches the autograd backward of render_w_brdf
    from materialist_amd import render

    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=False)
    scene._set("emitter.data", _t(sc.light, dev))
    r_t = _t(rgh, dev).requires_grad_(True)
    render.ATTACHED_SAMPLING = True
    try:
        out = render.render_w_brdf(scene, _t(sc.albedo, dev), r_t, _t(sc.metallic, dev), _t(n, dev), spp)
        out.backward(_t(d_out, dev))
    finally:
        render.ATTACHED_SAMPLING = False
    assert_close(r_t.grad, ref, rtol=2e-3, what="attached d_r through render_w_brdf")       # (the same pixel: 1.37e-3)


def test_extreme_materials_and_back_facing_normals(oracle64):
    """Edge cases the reference's known-answer vectors hold at lane level (App. C: back-facing / zero cases), here at image level:
    parameter maps on their clamp boundaries (roughness 0.07 / 1, metallic 0 / 1, albedo 0 / 1) and normals over the whole
    sphere, half of them facing away from the camera (n.wo < 0: NoV clamps to zero, :1391-1397, the lobes are sampled around n all the same)."""
    from materialist_amd import ops

    dev = _cuda()
    H, W, spp = 36, 44, 16
    rng = np.random.default_rng(21)
    sc, _ = _scene_arrays(H, W, image_id=3)
    a = rng.choice([0.0, 1.0, 0.5], size=(H, W, 3)).astype(np.float32)
    r = rng.choice([0.07, 1.0, 0.3], size=(H, W, 1)).astype(np.float32)
    m = rng.choice([0.0, 1.0], size=(H, W, 1)).astype(np.float32)
    n = rng.normal(size=(H, W, 3))
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    n = n.astype(np.float32)
    d_out = rng.normal(size=(H, W, 3)).astype(np.float32)
    ref_f = oracle64.shade_fwd(a, r, m, n, sc.light, spp)
    ref_b = oracle64.shade_bwd(a, r, m, n, sc.light, d_out, spp)
    assert ((n * np.array([0, 0, 1.0])).sum(-1) < 0).mean() > 0.3   # (wo is within 18 degrees of +z: these normals face away)
    args = [_t(x, dev) for x in (a, r, m, n, sc.light)]
    assert_close(ops.shade_fwd(*args, spp), ref_f, what="fwd, extreme maps")
    got = ops.shade_bwd(*args, _t(d_out, dev), spp, want_mat=True, want_n=True, want_light=True)
    for nm, g_, rf in zip(("d_a", "d_r", "d_m", "d_n", "d_light"), got, ref_b):
        assert_close(g_, rf, rtol=RTOL, what=f"{nm}, extreme maps")          # measured: d_r 2.0e-5, d_n 2.9e-4 (3e-3 until round 5)


def test_non_unit_normal_map_is_normalised(oracle64):
    from materialist_amd import ops

    dev = _cuda()
    H, W, spp = 24, 24, 8
    sc, n = _scene_arrays(H, W, image_id=2)
    scale = np.random.default_rng(1).uniform(0.3, 3.0, (H, W, 1)).astype(np.float32)
    d_out = np.ones((H, W, 3), np.float32)
    ref_f = oracle64.shade_fwd(sc.albedo, sc.roughness, sc.metallic, n * scale, sc.light, spp)
    ref_b = oracle64.shade_bwd(sc.albedo, sc.roughness, sc.metallic, n * scale, sc.light, d_out, spp)
    args = [_t(x, dev) for x in (sc.albedo, sc.roughness, sc.metallic, n * scale, sc.light)]
    assert_close(ops.shade_fwd(*args, spp), ref_f, what="fwd")
    got = ops.shade_bwd(*args, _t(d_out, dev), spp, want_mat=True, want_n=True, want_light=False)
    assert_close(got[3], ref_b[3], rtol=RTOL, what="d_n through normalisation")


def test_autograd_operator_face(oracle64):
    """render_w_brdf / render_envmap (inverse_img_w_mi.py:59-80) are differentiable in every tensor argument."""
    from materialist_amd import render, sh

    dev = _cuda()
    H, W, spp = 32, 32, 16
    sc, n = _scene_arrays(H, W, image_id=4, unit_random_normals=False)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=False)
    a, r, m, nn = (_t(x, dev).requires_grad_(True) for x in (sc.albedo, sc.roughness, sc.metallic, n))
    env = (torch.rand(16, 32, 3, device=dev) + 0.2).requires_grad_(True)
    img0 = render.render_envmap(scene, env, spp)          # materials = MatDiffBSDF defaults (0.5)
    assert img0.shape == (H, W, 3)
    img = render.render_w_brdf(scene, a, r, m, nn, spp)   # light persists in the scene from the previous call
    g_out = torch.randn_like(img)
    (img * g_out).sum().backward()
    coef = (sh.envmap_to_sh_matrix(16, 32) @ env.detach().cpu().numpy().reshape(512, 3).astype(np.float64))
    ref_img = oracle64.shade_fwd(sc.albedo, sc.roughness, sc.metallic, n, coef, spp)
    assert_close(img, ref_img, what="render_w_brdf")
    d_a, d_r, d_m, d_n, d_l = oracle64.shade_bwd(sc.albedo, sc.roughness, sc.metallic, n, coef, g_out.cpu().numpy(), spp)
    assert_close(a.grad, d_a, what="a.grad")
    assert_close(r.grad, d_r, rtol=RTOL, what="r.grad")
    assert_close(m.grad, d_m, what="m.grad")
    assert_close(nn.grad, d_n, rtol=RTOL, what="n.grad")
    d_env = (sh.envmap_to_sh_matrix(16, 32).T @ d_l).reshape(16, 32, 3)
    assert_close(env.grad, d_env, what="envmap.grad")
    # use_mesh_normal=True shades with the geometric normal, not the n map (F10)
    scene2 = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene2._set("emitter.data", env.detach())
    img2 = render.render_w_brdf(scene2, a.detach(), r.detach(), m.detach(), None, spp)
    geo = oracle64.normals_from_depth(sc.depth.astype(np.float64))
    assert_close(img2, oracle64.shade_fwd(sc.albedo, sc.roughness, sc.metallic, geo, coef, spp), what="mesh-normal render")


def test_operator_face_keeps_its_models_between_calls_and_rebuilds_them_when_the_light_changes():
    """render_w_brdf called again and again under one light (the reference's BRDF loop, inverse_img_w_mi.py:384-386): from the second call
    with the same (light, normals) on, the scene renders from cached per-pixel models and differentiates through their jac planes -- within
    1e-3 of the exact render and of the exact d_a / d_m, 3e-3 of the exact d_r (worst pixel); a new light (another tensor, or the same tensor modified in place)
    invalidates the cache, and the renders follow the new light."""
    from materialist_amd import ops, render

    dev = _cuda()
    H, W, spp = 64, 96, 64
    sc, n = _scene_arrays(H, W, image_id=6)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    nrm = scene.shading_normal()
    L1 = _t(sc.light, dev)
    L2 = (L1 * 0.7 + 0.05 * torch.randn_like(L1)).contiguous()
    L2[0] = L1[0]
    scene._set("emitter.data", L1)
    a0, r0, m0 = (_t(x, dev) for x in (sc.albedo, sc.roughness, sc.metallic))

    def call(light, step):
        a, r, m = (x.clone().requires_grad_(True) for x in (a0, (r0 + 2e-4 * step).clamp(0.07, 1.0), m0))
        img = render.render_w_brdf(scene, a, r, m, None, spp)
        g_out = torch.ones_like(img)
        img.backward(g_out)
        jac = ops.plane9(a0)
        exact = ops.shade_fwd(a.detach(), r.detach(), m.detach(), nrm, light, spp, jac=jac)
        d_a, d_r, d_m = ops.shade_bwd_jac(a.detach(), r.detach(), m.detach(), jac, g_out)
        e_img = float(((img.detach() - exact).abs() / torch.maximum(exact.abs(), exact.abs().mean())).max())
        err = lambda got, ref: float(((got - ref.reshape(got.shape)).abs() / torch.maximum(ref.abs(), ref.abs().mean()).reshape(got.shape)).max())
        # d out / d r of a model is the detached derivative to FIRST order in r - r_ref (round 5; half-precision value, e5m2 slope, as in the
        # fused loops): measured 1.1e-3 on the worst pixel (3e-2 was the bound of the zeroth-order models), bound 3e-3; d_a and d_m come from
        # fp32 P and S0 - S1: measured 1.3e-4, bound 1e-3
        _report("operator-face cache: render", e_img, 1e-3)
        _report("operator-face cache: d_a, d_m", max(err(a.grad, d_a), err(m.grad, d_m)), 1e-3)
        _report("operator-face cache: d_r (worst pixel)", err(r.grad, d_r), 3e-3)
        return e_img, max(err(a.grad, d_a), err(m.grad, d_m), err(r.grad, d_r) / 3.0)

    for step in range(4):                       # call 0: a light seen for the first time (direct); call 1 builds the cache; 2, 3 render from it
        e_img, e_g = call(L1, step)
        assert e_img < 1e-3 and e_g < 1e-3, (step, e_img, e_g)
    assert scene.cache_builds == 1
    scene._set("emitter.data", L2)              # another light: the models of L1 must not be used
    for step in range(3):
        e_img, e_g = call(L2, step)
        assert e_img < 1e-3 and e_g < 1e-3, ("L2", step, e_img, e_g)
    assert scene.cache_builds == 2
    L2.mul_(1.25)                               # the same tensor, modified in place (an optimiser step): a new version
    for step in range(3):
        e_img, e_g = call(L2, step)
        assert e_img < 1e-3 and e_g < 1e-3, ("L2 in place", step, e_img, e_g)
    assert scene.cache_builds == 3
    # a call without gradients under a cached light: the exact render (GGX samples + cached diffuse coefficients)
    with torch.no_grad():
        img = render.render_w_brdf(scene, a0, r0, m0, None, spp)
    exact = ops.shade_fwd(a0, r0, m0, nrm, L2, spp)
    assert float(((img - exact).abs() / torch.maximum(exact.abs(), exact.abs().mean())).max()) < 2e-5


@pytest.mark.parametrize("B,part", [(1, "rm"), (1, "arm"), (3, "a"), (3, "rm")])
def test_fused_brdf_loss_node_is_the_torch_composition(B, part):
    """a11 on the drop-in face (round 5): `loss.brdf_loss` of CUDA tensors is ONE autograd node over matpbr_brdf_loss_stats and
    matpbr_brdf_loss_dpred; against the torch composition of inverse_img_w_mi.py:388-418 (the same function with `loss.FUSED = False`): the
    loss, loss_mse, the ratio, pred_srgb, and the gradients with respect to the render and to every regularised map, single image and batch."""
    from materialist_amd import loss as L

    dev = _cuda()
    torch.manual_seed(31)
    H, W = 48, 80
    shp = (H, W) if B == 1 else (B, H, W)
    gt = torch.rand(shp + (3,), device=dev) * 0.8 + 0.05
    pred0 = (gt * (0.6 + 0.5 * torch.rand(shp + (3,), device=dev))).contiguous()
    keys = {"a": ("albedo", 3), "r": ("roughness", 1), "m": ("metallic", 1)}
    raw = {keys[c][0]: torch.rand(shp + (keys[c][1],), device=dev) for c in part}
    orig = {k: (v + 0.1 * torch.randn_like(v)).clamp(0, 1) for k, v in raw.items()}
    res = {}
    for fused in (False, True):
        L.FUSED = fused
        try:
            pred = pred0.clone().requires_grad_(True)
            ps = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
            parts = {k: (v.clamp(0.07, 1) if k == "roughness" else v.clamp(0, 1)) for k, v in ps.items()}
            loss, mse, pred_srgb, ratio = L.brdf_loss(pred * 1.0, gt, parts, orig, 0.1)
            (loss * 1.7).backward()
            res[fused] = (loss.detach(), mse.detach(), pred_srgb.detach(), ratio.detach() if torch.is_tensor(ratio) else ratio, pred.grad, {k: v.grad for k, v in ps.items()})
        finally:
            L.FUSED = True
    (l0, m0, s0, r0, g0, gp0), (l1, m1, s1, r1, g1, gp1) = res[False], res[True]
    assert float(l1) == pytest.approx(float(l0), rel=2e-5)
    assert torch.allclose(m1.reshape(-1), m0.reshape(-1), rtol=2e-5) and torch.allclose(torch.as_tensor(r1).reshape(-1), torch.as_tensor(r0).reshape(-1).to(dev), rtol=1e-5)
    assert (s1 - s0).abs().max().item() <= 2e-6
    assert (g1 - g0).abs().max().item() <= 2e-4 * g0.abs().max().item()
    for k in gp0:
        assert (gp1[k] - gp0[k]).abs().max().item() <= 1e-6 * max(gp0[k].abs().max().item(), 1e-30) + 1e-12, k


def test_operator_face_cache_is_keyed_on_tensor_objects_not_addresses():
    """a batch with ONE shared [25,3] light is rendered through an expanded (copied) light, so the cache cannot be keyed on the copy's
    address.  Two `_set("emitter.data", ...)` without a render in between, lights that are freed and re-allocated (the caching allocator hands
    the same address to the next one), a light written through its raw storage + `invalidate_cache()`: every render is the render of the
    light that is set."""
    import gc

    from materialist_amd import ops, render

    dev = _cuda()
    B, H, W, spp = 2, 32, 64, 16
    scs = [_scene_arrays(H, W, image_id=i)[0] for i in (3, 4)]
    depth = torch.stack([_t(s.depth, dev) for s in scs])
    scene = render.load_estimated_mesh(depth, use_mesh_normal=True)
    nrm = scene.shading_normal()
    a0, r0, m0 = (torch.stack([_t(getattr(s, k), dev) for s in scs]) for k in ("albedo", "roughness", "metallic"))
    base = _t(scs[0].light, dev)

    def check(light):
        a, r, m = (x.clone().requires_grad_(True) for x in (a0, r0, m0))
        img = render.render_w_brdf(scene, a, r, m, None, spp)
        img.sum().backward()
        exact = ops.shade_fwd(a0, r0, m0, nrm, light.unsqueeze(0).expand(B, -1, -1).contiguous(), spp)
        assert float(((img.detach() - exact).abs() / torch.maximum(exact.abs(), exact.abs().mean())).max()) < 1e-3

    for k in range(6):                                   # each light lives for three renders (build on the second), then is dropped
        light = (base * (0.5 + 0.3 * k)).contiguous()
        scene._set("emitter.data", light)
        for _ in range(3):
            check(light)
        del light
        gc.collect()
    builds = scene.cache_builds
    assert builds == 6
    l1, l2 = (base * 0.4).contiguous(), (base * 1.7).contiguous()
    scene._set("emitter.data", l1)
    scene._set("emitter.data", l2)                       # no render in between
    for _ in range(3):
        check(l2)
    # a write through the raw storage (what this library's kernels do): no version counter moves, the caller says so
    l2.data.mul_(0.5)
    scene.invalidate_cache()
    for _ in range(3):
        check(l2)


# ------------------------------------------------------------------------- size-independent properties @ 512^2
def test_full_size_properties():
    from materialist_amd import ops, synthetic

    dev = _cuda()
    H = W = 512
    spp = 64
    sc = synthetic.make_scene(0, H, W)
    sc2 = synthetic.make_scene(1, H, W)
    n = ops.normals_from_depth(_t(sc.depth, dev))
    a, r, m = _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev)
    l1, l2 = _t(sc.light, dev), _t(sc2.light, dev)
    o1 = ops.shade_fwd(a, r, m, n, l1, spp)
    o2 = ops.shade_fwd(a, r, m, n, l2, spp)
    o12 = ops.shade_fwd(a, r, m, n, 0.5 * l1 + 2.0 * l2, spp)
    assert torch.isfinite(o1).all() and (o1 >= -1e-4).all()
    # linear in the light
    assert (o12 - (0.5 * o1 + 2.0 * o2)).abs().max() <= 1e-4 * o12.abs().max()
    # deterministic: bit-identical on repeat
    assert torch.equal(o1, ops.shade_fwd(a, r, m, n, l1, spp))
    # batching: images of a batch match their stand-alone renders bit for bit
    nb = ops.normals_from_depth(_t(np.stack([sc.depth, sc2.depth]), dev))
    ab, rb, mb = (_t(np.stack([x, y]), dev) for x, y in ((sc.albedo, sc2.albedo), (sc.roughness, sc2.roughness), (sc.metallic, sc2.metallic)))
    ob = ops.shade_fwd(ab, rb, mb, nb, torch.stack([l1, l2]), spp)
    assert torch.equal(ob[0], o1)
    # backward: <d_out, J dl> == <J^T d_out, dl> for the (linear) light argument, and reproducible light gradient
    d_out = torch.randn_like(o1)
    g1 = ops.shade_bwd(a, r, m, n, l1, d_out, spp, want_mat=True, want_light=True)
    g2 = ops.shade_bwd(a, r, m, n, l1, d_out, spp, want_mat=True, want_light=True)
    assert all(torch.equal(x, y) for x, y in zip(g1, g2) if x is not None)
    lhs = (d_out.double() * o2.double()).sum()
    rhs = (g1[4].double() * l2.double()).sum()
    assert abs(lhs - rhs) <= 1e-4 * abs(lhs)
    # albedo gradient of an albedo-linear function: <d_a, a> + (a-independent part) -- check via finite difference on a scalar scale
    eps = 1e-2
    op = ops.shade_fwd(a * (1 + eps), r, m, n, l1, spp)
    om = ops.shade_fwd(a * (1 - eps), r, m, n, l1, spp)
    fd = ((op.double() - om.double()) * d_out.double()).sum() / (2 * eps)
    an = (g1[0].double() * a.double()).sum()
    assert abs(fd - an) <= 2e-3 * abs(fd)


def test_error_behaviour():
    from materialist_amd import ops
    from materialist_amd._lib import MatpbrError

    dev = _cuda()
    a = torch.rand(8, 8, 3, device=dev)
    r = torch.rand(8, 8, 1, device=dev)
    n = torch.zeros(8, 8, 3, device=dev)
    n[..., 2] = 1
    light = torch.zeros(25, 3, device=dev)
    with pytest.raises(ValueError):
        ops.shade_fwd(a, r, r, n, light, 3)
    with pytest.raises(ValueError):
        ops.shade_fwd(a, r, r, n, light, 130)
    with pytest.raises(MatpbrError):
        ops.shade_fwd(a.cpu(), r, r, n, light, 8)
    with pytest.raises(TypeError):
        ops.shade_fwd(a.double(), r, r, n, light, 8)


# ------------------------------------------------------------------------------- a11 + optimiser: fused hot loop B
def test_fused_brdf_phase_matches_torch_composition():
    """FusedBrdfPhase (HIP loss statistics + fused loss backward + HIP Adam) against BrdfPhase (same step composed from
    torch ops around the autograd render): same loss numbers, same parameters after several iterations."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H = W = 64
    spp = 16
    sc = synthetic.make_scene(5, H, W)
    depth = _t(sc.depth, dev)
    light = _t(sc.light, dev)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    # push a few parameters outside the clamp range so that the gating is exercised
    init[0][:4] = 1.2
    init[1][4:8] = 0.01
    init[2][8:12] = -0.3

    def make_scene():
        s = render.load_estimated_mesh(depth, use_mesh_normal=True)
        s._set("emitter.data", light)
        return s

    with torch.no_grad():
        gt = render.render_w_brdf(make_scene(), _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    ref = loop.BrdfPhase(make_scene(), gt, *init, None, optimize_part="arm", spp=spp)
    fused = loop.FusedBrdfPhase(make_scene(), gt, *init, spp=spp, lazy=False)   # the kernels that walk every sample; lazy path: test_gpu_lazy.py
    for it in range(5):
        mse_ref = ref.step()
        fused.step()
        st = fused.stats[0].cpu().numpy()
        assert st[ops.STAT_MSE] == pytest.approx(float(mse_ref), rel=2e-4), f"iteration {it}"
        assert st[ops.STAT_LOSS] == pytest.approx(float(ref.last["loss"]), rel=2e-4)
    assert float(fused.stats[0, ops.STAT_BEST]) == pytest.approx(float(ref.saver.best_loss), rel=2e-4)
    for k in ("albedo", "roughness", "metallic"):
        # Adam normalises the step: 5 iterations move a parameter by at most 5*lr = 1.5e-3; compare on that scale
        diff = (fused.p[k] - ref.params[k].detach()).abs().max().item()
        assert diff < 3e-5, f"{k}: {diff}"
        assert_close(fused.best[k], ref.saver.best[k].cpu().numpy(), rtol=1e-4, what=f"best {k}")
    assert_close(fused.best_img, ref.saver.best["rendered_img"].cpu().numpy(), rtol=1e-3, what="best render")
    assert fused.history().shape == (5, 1)
    assert fused.poll()["iters"].tolist() == [5]


def test_fused_phases_shade_with_a_fixed_predicted_normal_map():
    """`--opt_order 'rm a n'` makes the whole run shade with the predicted normal map (inverse_img_w_mi.py:335-340,751-756); its parts
    WITHOUT 'n' leave that map alone, so they run the fused phases (the map is `scene.shading_normal()`): the exact fused step against
    BrdfPhase under the same map, and the lazy folded step against the exact one."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H = W = 64
    spp = 16
    sc = synthetic.make_scene(6, H, W)
    depth, light = _t(sc.depth, dev), _t(sc.light, dev)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    geo = render.load_estimated_mesh(depth, use_mesh_normal=True).shading_normal()
    g = torch.Generator(device="cpu").manual_seed(4)
    nmap = torch.nn.functional.normalize(geo + 0.25 * torch.randn(geo.shape, generator=g).to(dev), dim=-1).contiguous()
    assert float((nmap * geo).sum(-1).min()) < 0.99               # not the geometric normals

    def make_scene():
        s = render.load_estimated_mesh(depth, use_mesh_normal=False)
        s._set("emitter.data", light)
        s._set("shape.bsdf.n", nmap)
        return s

    assert torch.equal(make_scene().shading_normal(), nmap)
    with torch.no_grad():
        gt = render.render_w_brdf(make_scene(), _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), nmap, spp)
        gt_geo = render.render_w_brdf(render.load_estimated_mesh(depth, use_mesh_normal=True), _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    ref = loop.BrdfPhase(make_scene(), gt, *init, nmap, optimize_part="rm", spp=spp)
    assert ref.opt_keys == ["roughness", "metallic"]
    fused = loop.FusedBrdfPhase(make_scene(), gt, *init, optimize_part="rm", spp=spp, lazy=False)
    lazy = loop.FusedBrdfPhase(make_scene(), gt, *init, optimize_part="rm", spp=spp)
    assert lazy.fold
    monkey_loss = loop._loss
    for it in range(5):
        # the reference side as the independent statement of :371-432: the torch composition of the loss, and every render_w_brdf call walking
        # the samples (with its cache the operator face renders from per-pixel models from its second call on, and a pixel whose gradient is of
        # the size of the models' error may take an Adam step the other way: measured 9.4e-5 = 0.3 lr on one pixel)
        monkey_loss.FUSED, keep_cache = False, render.OPERATOR_CACHE
        render.OPERATOR_CACHE = "none"
        try:
            mse_ref = ref.step()
        finally:
            monkey_loss.FUSED, render.OPERATOR_CACHE = True, keep_cache
        fused.step()
        lazy.step()
        st = fused.stats[0].cpu().numpy()
        assert st[ops.STAT_MSE] == pytest.approx(float(mse_ref), rel=2e-4), f"iteration {it}"
        assert st[ops.STAT_LOSS] == pytest.approx(float(ref.last["loss"]), rel=2e-4)
        assert float(lazy.stats[0, ops.STAT_MSE]) == pytest.approx(float(mse_ref), rel=2e-3)
    for k in ("roughness", "metallic"):
        assert (fused.p[k] - ref.params[k].detach()).abs().max().item() < 3e-5, k
        assert (lazy.p[k] - ref.params[k].detach()).abs().max().item() < 3e-4, k     # Adam on a gradient good to ~1 %: a fraction of 5 lr
    assert torch.equal(fused.p["albedo"], init[0]) and torch.equal(lazy.p["albedo"], init[0])
    with pytest.raises(NotImplementedError):
        loop.FusedBrdfPhase(make_scene(), gt, *init, optimize_part="rmn", spp=spp)
    assert float((gt - gt_geo).abs().max()) > 1e-3                 # the map matters: the phases above did not shade with the geometry's normals


@pytest.mark.parametrize("part", ["n", "armn", "rn"])
def test_normal_phase_matches_the_torch_composition(part):
    """NormalBrdfPhase (a part that moves the normal map, launch by launch on the C ABI, SaveBest / EarlyStopping on the device) against
    BrdfPhase (autograd render, torch losses, torch Adam): the same losses, the same parameters after several Adam steps (Adam normalises
    the step: five iterations move a parameter by at most 5 lr = 1.5e-3; compared on that scale), the same snapshot."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H = W = 64
    spp = 16
    sc = synthetic.make_scene(7, H, W)
    depth, light = _t(sc.depth, dev), _t(sc.light, dev)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    init[0][:4] = 1.2                                               # outside the clamp range: the gating is exercised
    init[1][4:8] = 0.01
    geo = render.load_estimated_mesh(depth, use_mesh_normal=True).shading_normal()
    g = torch.Generator(device="cpu").manual_seed(5)
    n_true = torch.nn.functional.normalize(geo + 0.2 * torch.randn(geo.shape, generator=g).to(dev), dim=-1).contiguous()
    n_init = (1.7 * torch.nn.functional.normalize(geo + 0.1 * torch.randn(geo.shape, generator=g).to(dev), dim=-1)).contiguous()   # not unit length
    n_orig = torch.nn.functional.normalize(geo, dim=-1).contiguous()

    def make_scene():
        s = render.load_estimated_mesh(depth, use_mesh_normal=False)
        s._set("emitter.data", light)
        return s

    with torch.no_grad():
        gt = render.render_w_brdf(make_scene(), _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), n_true, spp)
    originals = {"albedo": init[0] * 0.9, "roughness": init[1] * 0.9, "metallic": init[2] * 0.9, "normal": n_orig}
    ref = loop.BrdfPhase(make_scene(), gt, *init, n_init, optimize_part=part, spp=spp, originals=originals)
    ph = loop.NormalBrdfPhase(make_scene(), gt, *init, n_init, optimize_part=part, spp=spp, originals=originals)
    for it in range(5):
        mse_ref = ref.step()
        ph.step()
        st = ph.stats[0].cpu().numpy()
        assert st[ops.STAT_MSE] == pytest.approx(float(mse_ref), rel=5e-4), f"iteration {it}"
        assert float(ph.loss()[0]) == pytest.approx(float(ref.last["loss"]), rel=5e-4), f"iteration {it}"
    assert float(ph.stats[0, ops.STAT_BEST]) == pytest.approx(float(ref.saver.best_loss), rel=5e-4)
    keys = {"a": "albedo", "r": "roughness", "m": "metallic", "n": "normal"}
    for ch, k in keys.items():
        if ch in part:
            diff = (ph.p[k] - ref.params[k].detach().reshape(ph.p[k].shape)).abs().max().item()
            assert diff < 5e-5, f"{k}: {diff}"
            assert (ph.p[k] - (n_init if k == "normal" else init["arm".index(ch)]).reshape(ph.p[k].shape)).abs().max().item() > 1e-3   # and it moved
        else:
            assert torch.equal(ph.p[k], init["arm".index(ch)].reshape(ph.p[k].shape))
        assert_close(ph.best[k], ref.saver.best[k].reshape(ph.best[k].shape).cpu().numpy(), rtol=1e-4, what=f"best {k}")
    assert (ph.best["normal"].norm(dim=-1) - 1).abs().max().item() < 1e-5
    assert_close(ph.best_img, ref.saver.best["rendered_img"].cpu().numpy(), rtol=1e-3, what="best render")
    assert ph.history().shape == (5, 1) and ph.poll()["iters"].tolist() == [5]
    # EarlyStopping on the device: the firing iteration still updates, the ones enqueued behind it change nothing
    es = loop.NormalBrdfPhase(make_scene(), gt, *init, n_init, optimize_part=part, spp=spp, originals=originals, patience=2, min_delta=0.5)
    es.run(3)
    snap = {k: v.clone() for k, v in es.p.items()}
    assert es.poll()["stopped"].tolist() == [True] and es.poll()["iters"].tolist() == [3]
    es.run(4)
    assert es.poll()["iters"].tolist() == [3] and all(torch.equal(es.p[k], snap[k]) for k in snap)
    with pytest.raises(NotImplementedError):
        loop.NormalBrdfPhase(make_scene(), gt, *init, n_init, optimize_part="rm", spp=spp)


def test_normal_phase_on_a_batch_and_on_a_ragged_image():
    """NormalBrdfPhase on a batch of images = the same images alone, bit for bit (per-image statistics, per-image EarlyStopping), and an
    image whose pixel count is no multiple of the workgroup size (50 x 70) against BrdfPhase."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    spp = 8
    H, W = 40, 48
    scs = [synthetic.make_scene(20 + i, H, W) for i in range(2)]
    st = lambda f: torch.stack([_t(f(s), dev) for s in scs])
    depth, light = st(lambda s: s.depth), st(lambda s: s.light)
    init = [st(lambda s: s.init_albedo), st(lambda s: s.init_roughness), st(lambda s: s.init_metallic)]
    geo = render.load_estimated_mesh(depth, use_mesh_normal=True).shading_normal()
    gen = torch.Generator(device="cpu").manual_seed(8)
    n_init = torch.nn.functional.normalize(geo + 0.15 * torch.randn(geo.shape, generator=gen).to(dev), dim=-1).contiguous()

    def make_scene(sel=None):
        s = render.load_estimated_mesh(depth if sel is None else depth[sel], use_mesh_normal=False)
        s._set("emitter.data", light if sel is None else light[sel])
        return s

    with torch.no_grad():
        gt = render.render_w_brdf(make_scene(), st(lambda s: s.albedo), st(lambda s: s.roughness), st(lambda s: s.metallic), geo, spp)
    both = loop.NormalBrdfPhase(make_scene(), gt, *init, n_init, optimize_part="rmn", spp=spp, patience=3, min_delta=0.2)
    both.run(8)
    info = both.poll()
    for i in range(2):
        one = loop.NormalBrdfPhase(make_scene(i), gt[i], *[x[i] for x in init], n_init[i], optimize_part="rmn", spp=spp, patience=3, min_delta=0.2)
        one.run(8)
        for k in ("roughness", "metallic", "normal"):
            assert torch.equal(one.p[k], both.p[k][i]), (i, k)
            assert torch.equal(one.best[k], both.best[k][i]), (i, k)
        assert torch.equal(one.history()[:, 0], both.history()[:, i])
        assert one.poll()["iters"].tolist() == [int(info["iters"][i])]
    # a ragged image
    H2, W2 = 50, 70
    sc = synthetic.make_scene(31, H2, W2)
    depth2, light2 = _t(sc.depth, dev), _t(sc.light, dev)
    init2 = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    geo2 = render.load_estimated_mesh(depth2, use_mesh_normal=True).shading_normal()
    n2 = torch.nn.functional.normalize(geo2 + 0.15 * torch.randn(geo2.shape, generator=gen).to(dev), dim=-1).contiguous()

    def scene2():
        s = render.load_estimated_mesh(depth2, use_mesh_normal=False)
        s._set("emitter.data", light2)
        return s

    with torch.no_grad():
        gt2 = render.render_w_brdf(scene2(), _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), geo2, spp)
    # (the anchor of the normal regulariser away from the start map: where normalize(n) equals its anchor up to rounding, sign(n - n0) is the
    # sign of that rounding -- in the reference too -- and Adam turns it into a step of +- lr)
    orig2 = {"albedo": init2[0], "roughness": init2[1], "metallic": init2[2], "normal": geo2}
    ref = loop.BrdfPhase(scene2(), gt2, *init2, n2, optimize_part="an", spp=spp, originals=orig2)
    ph = loop.NormalBrdfPhase(scene2(), gt2, *init2, n2, optimize_part="an", spp=spp, originals=orig2)
    for it in range(3):
        mse_ref = ref.step()
        ph.step()
        assert float(ph.stats[0, ops.STAT_MSE]) == pytest.approx(float(mse_ref), rel=5e-4), it
        assert float(ph.loss()[0]) == pytest.approx(float(ref.last["loss"]), rel=5e-4), it
    for k in ("albedo", "normal"):
        assert (ph.p[k] - ref.params[k].detach()).abs().max().item() < 5e-5, k
        assert torch.isfinite(ph.p[k]).all()


def test_fused_phase_parts_and_device_early_stopping():
    """optimize_part masks (only the part's maps move, only its regularisers count) and the on-device EarlyStopping:
    identical stop iteration to the host state machine fed with the recorded losses, nothing changes after the stop."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H = W = 48
    spp = 8
    sc = synthetic.make_scene(6, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    # part 'rm': albedo must not move; reference composition with the same part
    ref = loop.BrdfPhase(scene, gt, *init, None, optimize_part="rm", spp=spp)
    fused = loop.FusedBrdfPhase(scene, gt, *init, optimize_part="rm", spp=spp, lazy=False)
    for _ in range(3):
        mse_ref = ref.step()
        fused.step()
        assert float(fused.stats[0, ops.STAT_LOSS]) == pytest.approx(float(ref.last["loss"]), rel=2e-4)
    assert torch.equal(fused.p["albedo"], init[0])
    assert (fused.p["roughness"] - ref.params["roughness"].detach()).abs().max().item() < 3e-5
    assert float(fused.stats[0, ops.STAT_LA]) == 0.0 and float(fused.stats[0, ops.STAT_LR]) > 0.0
    # early stopping: huge min_delta -> every iteration after the first is a miss -> stops after 1 + patience iterations
    es = loop.FusedBrdfPhase(scene, gt, *init, optimize_part="arm", spp=spp, patience=4, min_delta=0.5, lazy=False)
    es.run(12)
    info = es.poll()
    assert info["stopped"].tolist() == [True] and info["iters"].tolist() == [5]
    host = loop.EarlyStopping(patience=4, min_delta=0.5)
    hist = es.history()[:, 0].cpu().tolist()
    stop_at = None
    for i, v in enumerate(hist[:5]):
        host(v)
        if host.early_stop:
            stop_at = i + 1
            break
    assert stop_at == 5
    assert all(v == 0.0 for v in hist[5:])            # iterations after the stop did not execute
    frozen = {k: v.clone() for k, v in es.p.items()}
    es.run(3)
    assert all(torch.equal(frozen[k], es.p[k]) for k in frozen)
    # batch of two images with independent stop flags: image 1's target equals its initial render -> flat loss -> stops early
    scb = render.load_estimated_mesh(_t(np.stack([sc.depth, sc.depth]), dev), use_mesh_normal=True)
    scb._set("emitter.data", _t(np.stack([sc.light, sc.light]), dev))
    initb = [torch.stack([x, x]).contiguous() for x in init]
    with torch.no_grad():
        pred0 = render.render_w_brdf(scb, initb[0].clamp(0, 1), initb[1].clamp(0.07, 1), initb[2].clamp(0, 1), None, spp)
    gtb = torch.stack([gt, pred0[1]]).contiguous()
    fb = loop.FusedBrdfPhase(scb, gtb, *initb, optimize_part="arm", spp=spp, patience=3, min_delta=0.3, lazy=False)
    fb.run(10)
    ib = fb.poll()
    assert ib["iters"][0].item() >= ib["iters"][1].item() and ib["stopped"][1].item()


@pytest.mark.parametrize("part", ["a", "am", "m"])
def test_fused_phase_with_fixed_roughness_reuses_the_specular_sums_bit_exactly(part):
    """Parts that leave the roughness alone ('a' of --opt_order 'rm a'): after the first iteration the fused step combines the
    specular sums kept from it instead of walking the samples.  Same fused operations in the same order: every map, the render and
    the statistics are bit-identical to the phase that walks the samples every iteration; and the torch composition agrees."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H, W, spp = 40, 56, 16
    sc = synthetic.make_scene(4, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    # (lazy=False: the phases that walk samples; the default phase renders every part from the per-pixel models, tests/test_gpu_lazy.py)
    cached = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp, lazy=False)
    walked = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp, lazy=False)
    assert cached.s1cache is not None
    walked._ph.s1cache = None                              # the same phase without the cache: samples walked every iteration
    ref = loop.BrdfPhase(scene, gt, *init, None, optimize_part=part, spp=spp)
    for it in range(6):
        cached.step()
        walked.step()
        ref.step()
        assert torch.equal(cached.pred, walked.pred), it
        assert torch.equal(cached.stats, walked.stats), it
        assert float(cached.stats[0, ops.STAT_LOSS]) == pytest.approx(float(ref.last["loss"]), rel=2e-4), it
    for k in cached.p:
        assert torch.equal(cached.p[k], walked.p[k]), k
    assert torch.equal(cached.p["roughness"], init[1])
    assert torch.equal(cached.best_img, walked.best_img)


def test_cached_forward_is_bit_identical_to_the_render():
    """matpbr_shade_fwd_keep / matpbr_shade_fwd_cached on the piecewise face: new albedo and metallic maps under the planes kept from
    a render with the same roughness, normals and light give the render's bits (with and without clamping, a batch of two, an odd
    pixel count)."""
    from materialist_amd import ops, synthetic

    dev = _cuda()
    H, W, spp = 33, 37, 32
    sc = [synthetic.make_scene(i, H, W) for i in (1, 2)]
    a, r, m, light = (_t(np.stack([getattr(s, k) for s in sc]), dev) for k in ("albedo", "roughness", "metallic", "light"))
    n = ops.normals_from_depth(_t(np.stack([s.depth for s in sc]), dev))
    dcache = ops.diffuse_cache(n, light, spp)
    jac, s1 = ops.plane9(a), torch.empty(3, 2, H, W, device=dev)
    for clamp in (False, True):
        ops.shade_fwd(a, r, m, n, light, spp, clamp_params=clamp, dcache=dcache, jac=jac, s1=s1)
        torch.manual_seed(int(clamp))
        a2 = (a + 0.3 * torch.randn_like(a)) if clamp else (a * torch.rand_like(a))
        m2 = (m + 0.3 * torch.randn_like(m)) if clamp else torch.rand_like(m)
        full = ops.shade_fwd(a2, r, m2, n, light, spp, clamp_params=clamp, dcache=dcache, jac=ops.plane9(a))
        fast = ops.shade_fwd_cached(a2, m2, jac, s1, clamp_params=clamp)
        assert torch.equal(full, fast), clamp


def test_adam_step_matches_torch():
    from materialist_amd import ops

    dev = _cuda()
    torch.manual_seed(0)
    p0 = torch.randn(10007, device=dev)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=3e-4)
    q, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for t in range(1, 6):
        g = torch.randn_like(p0) * (10.0 ** (t - 3))
        p.grad = g.clone()
        opt.step()
        ops.adam_step(q, g, m, v, 3e-4, t)
    assert (q - p.detach()).abs().max().item() < 1e-6


def test_optimize_envmap_ARMN_smoke():
    """The full alternating schedule on a small synthetic image: the trace has the expected shape and the fit improves."""
    from materialist_amd import loss, optimize, render, synthetic

    dev = _cuda()
    H = W = 48
    spp = 8
    sc = synthetic.make_scene(7, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp).clone()
    mat = {"albedo": _t(sc.init_albedo, dev), "roughness": _t(sc.init_roughness, dev), "metallic": _t(sc.init_metallic, dev),
           "gt_image": gt}
    scene0 = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    with torch.no_grad():
        first = render.render_w_brdf(scene0, mat["albedo"], mat["roughness"], mat["metallic"], None, spp)   # white unit light
        psnr0 = float(loss.psnr(first * (gt.mean() / first.mean()), gt))
    out = optimize.optimize_envmap_ARMN(scene0, mat, optimize_order=("rm", "a"), spp=spp, opt_env_from=0, opt_src="arm", num_epochs=60,
                                        sync_every=20)
    phases = [(t.loop, t.phase, t.part) for t in out["trace"]]
    assert phases[:3] == [(1, "env", ""), (1, "brdf", "rm"), (1, "brdf", "a")]
    assert out["trace"][2].stop == "skip 'a' in loop 1"
    assert phases[-1][1] == "end"
    # 60 epochs at the reference learning rates (1e-3 / 3e-4) only start the fit; it must move in the right direction
    assert out["psnr"] > psnr0 + 0.3, (psnr0, out["psnr"])
    assert out["albedo"].shape == (H, W, 3) and out["envmap"].shape == (16, 32, 3)


def test_env_texel_phase_matches_the_framework_composition():
    """Hot loop A of `--model_name none` (texels through a softplus) entirely on the C ABI, `envhead.EnvTexelPhase`, against
    `loop.FusedEnvPhase` whose head, head backward and Adam are framework ops: parameters, history, statistics and the best envmap over
    iterations that cross a learning-rate change and the hipGraph capture; the caller's tensor is updated by sync_params()."""
    from materialist_amd import loop, ops, render, synthetic
    from materialist_amd.envhead import EnvTexelPhase

    dev = _cuda()
    H = W = 64
    spp = 16
    sc = synthetic.make_scene(8, H, W)

    def make_scene():
        s = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
        p = render.traverse(s)
        p["shape.bsdf.a"], p["shape.bsdf.r"], p["shape.bsdf.m"] = _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev)
        return s

    with torch.no_grad():
        gt = render.render_envmap(make_scene(), _t(sc.light, dev), spp).clone()
    torch.manual_seed(2)
    raw0 = torch.randn(16, 32, 3, device=dev) * 0.3
    raw_a = raw0.clone().requires_grad_(True)
    opt_a = torch.optim.Adam([raw_a], lr=1e-2)
    ref = loop.FusedEnvPhase(make_scene(), gt, lambda: torch.nn.functional.softplus(raw_a), opt_a, spp=spp, patience=50, min_delta=1e-3)
    raw_b = raw0.clone().requires_grad_(True)
    tex = EnvTexelPhase(make_scene(), gt, raw_b, spp=spp, lr=1e-2, patience=50, min_delta=1e-3, use_graph=True)
    for it in range(12):
        if it == 7:
            loop.set_lr(opt_a, 4e-3)
            tex.set_lr(4e-3)
        ref.step()
        tex.step()
    assert tex._graph is not None
    tex.sync_params()
    # Adam turns a rounding-level difference of a near-zero gradient (texels the image barely sees) into a step of the order of lr
    assert (raw_a.detach() - raw_b.detach()).abs().max().item() < 1e-4 and (raw_a.detach() - raw_b.detach()).abs().mean().item() < 1e-6
    assert torch.allclose(tex.history(), ref.history(), rtol=2e-5)
    assert torch.allclose(tex.stats[:, : ops.STAT_BEST + 1], ref.stats[:, : ops.STAT_BEST + 1], rtol=2e-5, atol=1e-8)
    assert (tex.best_env - ref.best_env).abs().max().item() < 1e-4 and (tex.best_env - ref.best_env).abs().mean().item() < 1e-6   # softplus of the parameters above
    assert tex.poll()["iters"].tolist() == [12]
    # the fused tail (three launches per iteration: matpbr_env_texel_phase_step) against the seven launches it replaces: the same operations in
    # the same order -- parameters, Adam moments, statistics, history and the best envmap bit for bit, with EarlyStopping armed and firing --
    # and the same again with the iterations between two polls replayed as one unrolled graph (step_many, what optimize.env_phase_runner calls)
    runs = {}
    for fused, many in ((True, False), (False, False), (True, True)):
        EnvTexelPhase.FUSED_TAIL = fused
        try:
            raw = raw0.clone().requires_grad_(True)
            ph = EnvTexelPhase(make_scene(), gt, raw, spp=spp, lr=1e-2, patience=3, min_delta=0.2, use_graph=True)
        finally:
            EnvTexelPhase.FUSED_TAIL = True
        assert ph.fused_tail == fused
        if many:
            for _ in range(4):
                ph.step()
            ph.step_many(5)
            ph.set_lr(3e-3)
            ph.step_many(7)
            ph.step_many(7)
            ph.step_many(7)
            assert ph.t == 30 and sorted(ph._unrolled) == [5, 7]
        else:
            for it in range(30):
                if it == 9:
                    ph.set_lr(3e-3)
                ph.step()
        ph.sync_params()
        runs[fused, many] = (raw.detach().clone(), ph.adam_m.clone(), ph.adam_v.clone(), ph.stats.clone(), ph.history().clone(), ph.best_env.clone(), ph.poll())
    for other in ((False, False), (True, True)):
        for a, b in zip(runs[True, False][:6], runs[other][:6]):
            assert torch.equal(a, b)
        assert runs[True, False][6]["stopped"].tolist() == [True] and runs[True, False][6]["iters"].tolist() == runs[other][6]["iters"].tolist()
    assert runs[True, False][6]["iters"].tolist()[0] < 30


@pytest.mark.parametrize("env_size,hw", [((32, 32), (40, 56)), ((8, 16), (64, 64)), ((16, 32), (96, 128)), ((16, 32), (1024, 1024))])
def test_env_texel_tail_on_other_envmap_and_image_sizes(env_size, hw):
    """The one-workgroup tail of hot loop A (matpbr_env_texel_phase_step) beyond the 16 x 32 texels and the image sizes of the other tests: 1024
    texels (more parameters than its up-front requests cover), 128 texels, few partial rows (a 40 x 56 image: 9 tiles), and 1024 x 1024 (BASELINE
    configs[3]: 1024 partial rows, four chunks of them per slice of the fold) -- against the seven launches it replaces, bit for bit, with a
    learning-rate change and EarlyStopping firing."""
    from materialist_amd import render, synthetic
    from materialist_amd.envhead import EnvTexelPhase

    dev = _cuda()
    H, W = hw
    spp = 8
    sc = synthetic.make_scene(12, H, W)

    def make_scene():
        s = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
        p = render.traverse(s)
        p["shape.bsdf.a"], p["shape.bsdf.r"], p["shape.bsdf.m"] = _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev)
        return s

    with torch.no_grad():
        gt = render.render_envmap(make_scene(), _t(sc.light, dev), spp).clone()
    torch.manual_seed(5)
    raw0 = torch.randn(*env_size, 3, device=dev) * 0.3
    runs = {}
    for fused in (True, False):
        EnvTexelPhase.FUSED_TAIL = fused
        try:
            raw = raw0.clone().requires_grad_(True)
            ph = EnvTexelPhase(make_scene(), gt, raw, spp=spp, lr=1e-2, patience=3, min_delta=0.2, use_graph=True)
        finally:
            EnvTexelPhase.FUSED_TAIL = True
        for it in range(4):
            ph.step()
        ph.set_lr(3e-3)
        ph.step_many(6)
        ph.step_many(6)
        ph.sync_params()
        runs[fused] = (raw.detach().clone(), ph.adam_m.clone(), ph.adam_v.clone(), ph.stats.clone(), ph.history().clone(), ph.best_env.clone(), ph.g.clone(),
                       ph.poll())
    for a, b in zip(runs[True][:7], runs[False][:7]):
        assert torch.equal(a, b)
    assert torch.isfinite(runs[True][0]).all() and (runs[True][0] - raw0).abs().max().item() > 1e-3
    assert runs[True][7]["iters"].tolist() == runs[False][7]["iters"].tolist()


def test_fused_env_phase_matches_torch_composition():
    """FusedEnvPhase (one matpbr_env_phase_step per iteration) against EnvPhase (autograd render + torch loss): texel light
    through softplus + SH projection, Adam in torch on both sides."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H = W = 48
    spp = 8
    sc = synthetic.make_scene(8, H, W)

    def make_scene():
        s = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
        p = render.traverse(s)
        p["shape.bsdf.a"], p["shape.bsdf.r"], p["shape.bsdf.m"] = _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev)
        return s

    s_gt = make_scene()
    with torch.no_grad():
        gt = render.render_envmap(s_gt, _t(sc.light, dev), spp).clone()
    torch.manual_seed(1)
    raw0 = torch.randn(16, 32, 3, device=dev) * 0.3
    # reference composition: parameters -> softplus -> render_envmap (projection inside the scene) -> env_loss -> backward
    raw_a = raw0.clone().requires_grad_(True)
    opt_a = torch.optim.Adam([raw_a], lr=1e-2)
    sa = make_scene()
    raw_b = raw0.clone().requires_grad_(True)
    opt_b = torch.optim.Adam([raw_b], lr=1e-2)
    fused = loop.FusedEnvPhase(make_scene(), gt, lambda: torch.nn.functional.softplus(raw_b), opt_b, spp=spp)
    for it in range(4):
        pred = render.render_envmap(sa, torch.nn.functional.softplus(raw_a), spp)
        total, mse, l1 = loop._loss.env_loss(pred, gt)
        total.backward()
        opt_a.step()
        opt_a.zero_grad()
        fused.step()
        st = fused.stats[0].cpu().numpy()
        assert st[ops.STAT_MSE] == pytest.approx(float(mse.detach()), rel=2e-4), it
        assert st[ops.STAT_LOSS] == pytest.approx(float(total.detach()), rel=2e-4), it
    assert (raw_a - raw_b).abs().max().item() < 2e-4       # 4 Adam steps of 1e-2
    assert fused.poll()["iters"].tolist() == [4]
    assert fused.best_env.shape == (16, 32, 3)
    # the same iterations replayed from a hipGraph (3 eager + captured): identical trajectory
    raw_c = raw0.clone().requires_grad_(True)
    opt_c = loop.capturable_adam([raw_c], 1e-2)
    gph = loop.FusedEnvPhase(make_scene(), gt, lambda: torch.nn.functional.softplus(raw_c), opt_c, spp=spp, use_graph=True)
    raw_d = raw0.clone().requires_grad_(True)
    eag = loop.FusedEnvPhase(make_scene(), gt, lambda: torch.nn.functional.softplus(raw_d), loop.capturable_adam([raw_d], 1e-2), spp=spp)
    for it in range(8):
        if it == 5:
            loop.set_lr(opt_c, 5e-3)
            loop.set_lr(eag.opt, 5e-3)
        gph.step()
        eag.step()
    assert gph._graph is not None
    assert gph.poll()["iters"].tolist() == [8]
    assert (raw_c - raw_d).abs().max().item() < 1e-6
    assert torch.allclose(gph.history(), eag.history(), rtol=1e-5)
    assert_close(fused.pred, pred.detach().cpu().numpy(), rtol=1e-3, what="last render")


def test_inverse_image_writes_the_reference_output_layout(tmp_path):
    """f1: the pipeline head + writers produce output_imgs/<name>/ as SURVEY.md App. D lists it (the mp4 files are GIFs)."""
    from PIL import Image

    from materialist_amd import pipeline

    _cuda()
    rng = np.random.default_rng(4)
    img = (rng.random((40, 56, 3)) * 255).astype(np.uint8)
    src = str(tmp_path / "in.png")
    Image.fromarray(img).save(src)
    res = pipeline.inverse_image(src, "case", opt_src="arm", opt_order=["rm", "a"], opt_env_from=0, save_path=str(tmp_path), size=32, spp=8,
                                 num_epochs=12, sync_every=6, log=lambda *_: None, frame_interval=0.0, model_name="none")
    out = res["output_dir"]
    assert out == str(tmp_path / "case")
    for name in ("albedoPred.exr", "normalPred.exr", "roughnessPred.png", "metallicPred.png", "depthPred.exr", "gt_image.exr", "gt_image.png",
                 "config.json", "env.png", "final_envmap.hdr", "opt_env_img.png", "env_optimization.mp4", "mat_optimization.mp4", "case.ply"):
        assert os.path.exists(os.path.join(out, name)), name
    assert sorted(os.listdir(os.path.join(out, "best_results"))) == ["albedo.exr", "envmap.hdr", "metallic.exr", "normal.exr",
                                                                     "rendered_img.exr", "roughness.exr"]
    assert len(os.listdir(os.path.join(out, "env_frames"))) >= 2 and len(os.listdir(os.path.join(out, "mat_frames"))) >= 2
    import json

    cfg = json.load(open(os.path.join(out, "config.json")))
    assert set(cfg) == {"img_path", "save_name", "opt_src", "opt_order", "use_mask", "opt_env_from", "model_name", "timestamp", "image_size",
                        "spp", "output_type", "use_mesh_normal"}
    assert Image.open(os.path.join(out, "opt_env_img.png")).size == (96, 32)        # three panels side by side
    # resume path (--opt_src skip, :737-749) reads what was written
    res2 = pipeline.inverse_image(src, "case", opt_src="skip", opt_order=["skip"], save_path=str(tmp_path), size=32, spp=8, num_epochs=5,
                                  sync_every=5, log=lambda *_: None, model_name="none")
    assert res2["trace"][-1].stop == "skip"
    # 'n' in --opt_order: shade with (and optimise) the normal map (inverse_img_w_mi.py:751-758,378-379)
    res3 = pipeline.inverse_image(src, "case_mn", opt_src="arm", opt_order=["arm", "n"], save_path=str(tmp_path), size=32, spp=8, num_epochs=6,
                                  sync_every=6, log=lambda *_: None, model_name="none")
    assert [t.part for t in res3["trace"] if t.phase == "brdf" and t.loop == 1] == ["arm", "n"]
    assert res3["normal"] is not None and abs(float(res3["normal"].norm(dim=-1).mean()) - 1.0) < 1e-4
    assert json.load(open(os.path.join(res3["output_dir"], "config.json")))["use_mesh_normal"] is False
    # f4: re-render the optimised scene under its own envmap and as a rolling animation (render_final.py)
    from materialist_amd import relight

    png = relight.render_real("case", None, input_path=str(tmp_path), save_path=str(tmp_path), spp=8)
    assert png.endswith("mi_case_envmap_.png") and os.path.exists(png) and os.path.exists(png[:-4] + ".exr")
    roll = relight.render_rolling_envmap("case", None, frames=10, rotation_step=36.0, input_path=str(tmp_path), save_path=str(tmp_path), spp=8)
    assert len(roll["frames"]) == 10 and os.path.basename(roll["frames"][3]) == "frame_0003.png"
    assert os.path.exists(roll["gif"]) and os.path.basename(roll["gif"]) == "rolling_envmap_case_envmap.gif"
    assert os.path.exists(roll["mp4"]) and os.path.basename(roll["mp4"]) == "rolling_envmap_case_envmap.mp4"      # render_final.py:405-409


@pytest.mark.parametrize("part", ["armn", "rmn"])
def test_armn_network_launch_by_launch_is_the_autograd_phase(part):
    """the eight-output 'armn' network (10 inputs: raw coordinates + 8 channels, hidden layers 246 / 256 / 246 / 256; mymodels/mlps.py:236-244,
    inverse_img_w_mi.py:167-172) launch by launch on the C ABI -- `armhead.MlpEngine` inside `loop.PosMlpNormalPhase`: the first layer on the thin-K
    kernel, the 256-wide layers on two f16 pieces forward and backward, the 8-column output layer, AdamW with SaveBest's weight
    snapshot on the flat buffer; the head (tanh, residual, straight-through clamps, normalize) and its backward as element-wise passes -- against
    the same phase with the network under autograd and torch.optim.AdamW (`ENGINE = False`): the gradients that reach every parameter in the
    first iteration, the losses of five iterations, the weights after them and SaveBest's copy."""
    import copy

    from materialist_amd import loop, posmlp, render, synthetic

    dev = _cuda()
    H = W = 96                                                       # 9216 points: whole 128-row tiles, above the layer kernels' minimum
    spp = 8
    sc = synthetic.make_scene(9, H, W)
    depth, light = _t(sc.depth, dev), _t(sc.light, dev)
    geo = render.load_estimated_mesh(depth, use_mesh_normal=True).shading_normal()
    gen = torch.Generator(device="cpu").manual_seed(6)
    n_true = torch.nn.functional.normalize(geo + 0.2 * torch.randn(geo.shape, generator=gen).to(dev), dim=-1).contiguous()

    def make_scene():
        s = render.load_estimated_mesh(depth, use_mesh_normal=False)
        s._set("emitter.data", light)
        return s

    with torch.no_grad():
        gt = render.render_w_brdf(make_scene(), _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), n_true, spp)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    start = torch.cat([init[0].reshape(-1, 3), init[1].reshape(-1, 1), init[2].reshape(-1, 1), geo.reshape(-1, 3)], dim=-1).contiguous()
    fixed = {"albedo": init[0], "roughness": init[1], "metallic": init[2], "normal": geo}
    torch.manual_seed(4)
    net_a = posmlp.brdf_net("armn").to(dev)
    net_a.lin4.weight.data.normal_(0, 0.02)                          # (the reference starts the output layer at zero: every hidden gradient would be zero)
    net_b = copy.deepcopy(net_a)
    runs = {}
    for engine, net in ((True, net_a), (False, net_b)):
        loop.PosMlpNormalPhase.ENGINE = engine
        try:
            ph = loop.PosMlpNormalPhase(make_scene(), gt, net, start, fixed, optimize_part=part, spp=spp, saver=loop.DeviceSaveBest())
            assert (ph.engine is not None) == engine
            first = []
            if engine:
                mses = [float(ph.step())]
                first = [(gw[:, :wp_k].clone(), gb[:bp.numel()].clone()) for (gw, gb), (wp, bp), wp_k in
                         zip(ph.engine.gviews, ph.engine.views, [getattr(net, f"lin{l}").linear.weight.shape[1] if l < 4 else net.lin4.weight.shape[1] for l in range(5)])]
                first = [t for pair in first for t in pair]
            else:
                opt_step = ph.opt.step

                def capture(*a, **kw):
                    if not first:
                        first.extend(p_.grad.detach().clone() for p_ in net.parameters())
                    return opt_step(*a, **kw)

                ph.opt.step = capture
                mses = [float(ph.step())]
            mses += [float(ph.step()) for _ in range(4)]
            if engine:      # Adam's bias corrections count the statistics row's iterations: the two counters move together
                from materialist_amd import ops as _ops

                assert float(ph.engine.hyper[1]) == 5.0 == float(ph.stats[0, _ops.STAT_ITERS])
        finally:
            loop.PosMlpNormalPhase.ENGINE = True
        runs[engine] = (mses, {k: v.detach().clone() for k, v in net.state_dict().items()}, float(ph.saver.best_loss),
                        {k: v.clone() for k, v in ph.best_weights.items()}, first, {k: v.clone() for k, v in ph.saver.best.items()})
    names = [n for n, _ in net_a.named_parameters()]
    assert len(runs[True][4]) == len(runs[False][4]) == len(names)
    errs = {name: ((ga - gb).norm().item() / (gb.norm().item() + 1e-30)) for ga, gb, name in zip(runs[True][4], runs[False][4], names)}
    print("first-iteration gradients, engine against autograd, relative:", {k: f"{v:.1e}" for k, v in errs.items()})
    for ga, gb, name in zip(runs[True][4], runs[False][4], names):        # f32-accurate products on both sides (two f16 pieces / three bf16 pieces)
        assert ga.shape == gb.shape, name
    # f32-accurate products on both sides; what separates them is the reference's own construction: a saturated output's straight-through value
    # (clamp(u) + u) - u is 1 to ROUNDING, the outer clamp of :493-496 passes its gradient only if that rounding fell at or below 1, and two
    # forwards that agree to 1e-7 disagree on a share of those entries (the autograd phase on the HIP layer kernels against the same phase on
    # torch.mm differs by the same 3.7e-3 in the first layer, 5e-4 in the last: tools history in EXPERIMENTS.md)
    assert max(errs.values()) <= 8e-3, errs
    for a, b in zip(runs[True][0], runs[False][0]):
        assert a == pytest.approx(b, rel=5e-4)
    assert runs[True][2] == pytest.approx(runs[False][2], rel=5e-4)
    for k, v in runs[True][1].items():                                    # (AdamW's first steps are lr * sign(g): entries whose gradient is rounding move either way)
        diff = (v - runs[False][1][k]).abs()
        assert diff.median().item() < 5e-6 and (diff > 1e-4).float().mean().item() < 0.15, (k, diff.max().item(), (diff > 1e-4).float().mean().item())
    for k, v in runs[True][3].items():
        diff = (v - runs[False][3][k]).abs()
        assert diff.median().item() < 5e-6, k
    for k in ("albedo", "roughness", "metallic", "normal"):
        assert torch.allclose(runs[True][5][k], runs[False][5][k], atol=2e-3), k


@pytest.mark.parametrize("model_name", ["none", "pos_mlp"])
def test_use_mask_keeps_roughness_and_metallic_uniform_inside_the_mask(tmp_path, model_name):
    """--use_mask (inverse_img_w_mi.py:379-381,509-511,702-711): best_results/mask.png marks one material region."""
    from PIL import Image

    from materialist_amd import pipeline
    from materialist_amd.imageio_exr import read_exr

    _cuda()
    rng = np.random.default_rng(7)
    src = str(tmp_path / "in.png")
    Image.fromarray((rng.random((32, 32, 3)) * 255).astype(np.uint8)).save(src)
    os.makedirs(tmp_path / "case" / "best_results")
    mk = np.zeros((32, 32, 3), np.uint8)
    mk[8:20, 4:28] = 255
    Image.fromarray(mk).save(str(tmp_path / "case" / "best_results" / "mask.png"))
    pipeline.inverse_image(src, "case", opt_src="arm", opt_order=["rm", "a"], use_mask=True, opt_env_from=0, save_path=str(tmp_path), size=32,
                           spp=8, num_epochs=6, sync_every=3, log=lambda *_: None, model_name=model_name)
    for name in ("roughness", "metallic"):
        x = read_exr(str(tmp_path / "case" / "best_results" / f"{name}.exr"))[..., 0]
        inside, outside = x[8:20, 4:28], np.concatenate([x[:8].ravel(), x[20:].ravel()])
        assert inside.max() - inside.min() < 1e-6, name
        assert outside.max() - outside.min() > 1e-4 or name == "metallic"      # the flat prior's metallic starts uniform


def test_mesh_mask_pixels_show_the_environment_and_feed_the_light(tmp_path):
    """mesh_mask.png (inverse_img_w_mi.py:713-724): pixels without geometry see the emitter along the camera ray; linear in the
    light, no material gradient; the drivers route such scenes through the operator face in both model modes."""
    from PIL import Image

    from materialist_amd import ops, pipeline, render, sh, synthetic

    dev = _cuda()
    H, W, spp = 40, 56, 8
    sc = synthetic.make_scene(12, H, W)
    mask = torch.zeros(H, W, dtype=torch.bool)
    mask[:11] = True
    mask[20:24, 30:40] = True
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True, mesh_mask=mask)
    plain = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    a = _t(sc.albedo, dev).requires_grad_(True)
    light = _t(sc.light, dev).requires_grad_(True)
    for s_ in (scene, plain):
        s_._set("emitter.data", light)
    img = render.render_w_brdf(scene, a, _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    ref = render.render_w_brdf(plain, a.detach(), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    m = mask.to(dev)
    assert torch.equal(img[~m], ref[~m])
    # camera rays of the masked pixels, the pinhole of the kernels (wo = -ray)
    f = (W / 2.0) / np.tan(np.radians(35.0) / 2.0)
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    ray = np.stack([(jj - (W - 1) / 2) / f, -(ii - (H - 1) / 2) / f, -np.ones_like(ii, dtype=np.float64)], -1)
    ray /= np.linalg.norm(ray, axis=-1, keepdims=True)
    expect = sh.sh_basis(ray[mask.numpy()]) @ sc.light.astype(np.float64)
    assert np.abs(img[m].detach().cpu().numpy() - expect).max() < 1e-5
    via_kernel = ops.sh_eval(_t(ray[mask.numpy()], dev), light.detach())
    assert (via_kernel - img[m].detach()).abs().max().item() < 1e-5
    img.sum().backward()
    assert float(a.grad[m].abs().max()) == 0.0 and float(a.grad[~m].abs().max()) > 0
    # forward-only relighting shows the same background
    from materialist_amd.relight import Relighter

    rl = Relighter({"albedo": a.detach(), "roughness": _t(sc.roughness, dev), "metallic": _t(sc.metallic, dev)}, scene.geo_normal, spp, mesh_mask=mask)
    fr = rl.frames(sc.light[None])[0]
    assert (fr - img.detach()).abs().max().item() <= 2e-5 * float(img.detach().abs().max())
    # drivers: a run with mesh_mask.png in the output directory, both model modes
    rng = np.random.default_rng(5)
    src = str(tmp_path / "in.png")
    Image.fromarray((rng.random((32, 32, 3)) * 255).astype(np.uint8)).save(src)
    for model_name in ("none", "pos_mlp"):
        os.makedirs(tmp_path / model_name, exist_ok=True)
        mk = np.zeros((32, 32), np.uint8)
        mk[:9] = 255
        Image.fromarray(mk).save(str(tmp_path / model_name / "mesh_mask.png"))
        lines = []
        res = pipeline.inverse_image(src, model_name, opt_src="arm", opt_order=["rm", "a"], opt_env_from=0, save_path=str(tmp_path), size=32,
                                     spp=8, num_epochs=5, sync_every=5, log=lines.append, model_name=model_name)
        assert any("see the environment directly" in ln for ln in lines)
        assert np.isfinite(res["psnr"]) and os.path.exists(os.path.join(res["output_dir"], "best_results", "envmap.hdr"))


def test_pos_mlp_with_predicted_normals_runs_and_improves(tmp_path):
    """'n' in --opt_order under --model_name pos_mlp: output_type 'armn' (inverse_img_w_mi.py:165-172,493-506), the MLP predicts the
    normal map too; the loop must lower the loss and write best_results/normal.exr with unit normals."""
    from PIL import Image

    from materialist_amd import pipeline
    from materialist_amd.imageio_exr import read_exr

    _cuda()
    torch.manual_seed(3)
    rng = np.random.default_rng(11)
    src = str(tmp_path / "in.png")
    Image.fromarray((rng.random((32, 32, 3)) * 255).astype(np.uint8)).save(src)
    lines = []
    res = pipeline.inverse_image(src, "case", opt_src="arm", opt_order=["armn"], opt_env_from=0, save_path=str(tmp_path), size=32, spp=8,
                                 num_epochs=25, sync_every=5, log=lines.append, model_name="pos_mlp")
    assert any("armn" in ln for ln in lines)
    n = read_exr(str(tmp_path / "case" / "best_results" / "normal.exr"))
    assert np.abs(np.linalg.norm(n, axis=-1) - 1).max() < 1e-4
    cfg = __import__("json").load(open(tmp_path / "case" / "config.json"))
    assert cfg["output_type"] == "armn" and cfg["use_mesh_normal"] is False
    assert res["best_loss"] < 0.2 and np.isfinite(res["psnr"])


def test_none_mode_with_n_in_the_order_runs_its_other_parts_fused(tmp_path):
    """--opt_order 'rm n' under --model_name none: the run shades with the predicted normal map throughout (use_mesh_normal False); the 'rm'
    part leaves it alone and runs the fused phase, the 'n' part NormalBrdfPhase (no autograd in either); normal.exr holds unit normals."""
    from PIL import Image

    from materialist_amd import pipeline
    from materialist_amd.imageio_exr import read_exr

    _cuda()
    torch.manual_seed(3)
    rng = np.random.default_rng(12)
    src = str(tmp_path / "in.png")
    Image.fromarray((rng.random((32, 32, 3)) * 255).astype(np.uint8)).save(src)
    lines = []
    res = pipeline.inverse_image(src, "case", opt_src="arm", opt_order=["rm", "n"], opt_env_from=0, save_path=str(tmp_path), size=32, spp=8,
                                 num_epochs=12, sync_every=4, log=lines.append, model_name="none")
    rm = [ln for ln in lines if "part 'rm'" in ln]
    nn = [ln for ln in lines if "part 'n'" in ln]
    assert rm and all("with normals" not in ln and "normal map" not in ln for ln in rm), lines
    assert nn and all("normal map, on the device" in ln for ln in nn), lines      # NormalBrdfPhase, not the autograd composition
    cfg = __import__("json").load(open(tmp_path / "case" / "config.json"))
    assert cfg["use_mesh_normal"] is False
    n = read_exr(str(tmp_path / "case" / "best_results" / "normal.exr"))
    assert np.abs(np.linalg.norm(n, axis=-1) - 1).max() < 1e-4
    assert np.isfinite(res["psnr"]) and res["best_loss"] < 0.5


@pytest.mark.parametrize("part", ["armn", "rmn"])
def test_pos_mlp_normal_phase_on_the_c_abi_matches_the_autograd_composition(part):
    """PosMlpNormalPhase (output_type 'armn': the net predicts the normal map too) with the render, the loss statistics, SaveBest's decision and
    the gradients of the maps on the C ABI against the same phase through the operator face and the torch-composed loss: same losses, same
    weights after several AdamW steps, same snapshot."""
    import copy

    from materialist_amd import loop, ops, posmlp, render, synthetic

    dev = _cuda()
    H = W = 48
    spp = 8
    sc = synthetic.make_scene(9, H, W)
    depth, light = _t(sc.depth, dev), _t(sc.light, dev)
    geo = render.load_estimated_mesh(depth, use_mesh_normal=True).shading_normal()
    gen = torch.Generator(device="cpu").manual_seed(6)
    n_true = torch.nn.functional.normalize(geo + 0.2 * torch.randn(geo.shape, generator=gen).to(dev), dim=-1).contiguous()

    def make_scene():
        s = render.load_estimated_mesh(depth, use_mesh_normal=False)
        s._set("emitter.data", light)
        return s

    with torch.no_grad():
        gt = render.render_w_brdf(make_scene(), _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), n_true, spp)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    start = torch.cat([init[0].reshape(-1, 3), init[1].reshape(-1, 1), init[2].reshape(-1, 1), geo.reshape(-1, 3)], dim=-1).contiguous()
    fixed = {"albedo": init[0], "roughness": init[1], "metallic": init[2], "normal": geo}
    torch.manual_seed(4)
    net_a = posmlp.brdf_net("armn").to(dev)
    net_b = copy.deepcopy(net_a)
    runs = {}
    for device_loss, net in ((True, net_a), (False, net_b)):
        loop.PosMlpNormalPhase.DEVICE_LOSS = device_loss
        try:
            ph = loop.PosMlpNormalPhase(make_scene(), gt, net, start, fixed, optimize_part=part, spp=spp, saver=loop.DeviceSaveBest())
            first_grads, opt_step = [], ph.opt.step

            def capture(*a, **kw):                                   # the gradients of the first iteration, as the optimiser sees them
                if not first_grads:
                    first_grads.extend(p_.grad.detach().clone() for p_ in net.parameters())
                return opt_step(*a, **kw)

            ph.opt.step = capture
            mses = [float(ph.step()) for _ in range(4)]
        finally:
            loop.PosMlpNormalPhase.DEVICE_LOSS = True
        assert hasattr(ph, "stats") == device_loss                   # which path ran
        runs[device_loss] = (mses, {k: v.detach().clone() for k, v in net.state_dict().items()}, {k: v.clone() for k, v in ph.saver.best.items()},
                             float(ph.saver.best_loss), {k: v.clone() for k, v in ph.best_weights.items()}, first_grads)
    for a, b in zip(runs[True][0], runs[False][0]):
        assert a == pytest.approx(b, rel=5e-4)
    assert runs[True][3] == pytest.approx(runs[False][3], rel=5e-4)
    # the same gradients reach the network: every parameter tensor to 1e-3 of its norm
    for ga, gb, (name, _) in zip(runs[True][5], runs[False][5], net_a.named_parameters()):
        assert (ga - gb).norm().item() <= 1e-3 * gb.norm().item() + 1e-12, name
    # ... and the weights after four AdamW steps of 3e-4 agree wherever the gradient is more than rounding (AdamW's first steps are lr * sign(g):
    # an entry whose gradient is zero to rounding -- the first layer's weights on near-constant inputs -- moves by +- lr in either run)
    for k, v in runs[True][1].items():
        diff = (v - runs[False][1][k]).abs()
        assert diff.median().item() < 2e-6 and (diff > 3e-5).float().mean().item() < 0.15, (k, diff.max().item(), (diff > 3e-5).float().mean().item())
    for k in ("albedo", "roughness", "metallic", "normal"):
        assert_close(runs[True][2][k].reshape(-1), runs[False][2][k].reshape(-1).cpu().numpy(), rtol=1e-3, what=f"best {k}")   # (the maps of weights that differ as above)
    assert_close(runs[True][2]["rendered_img"], runs[False][2]["rendered_img"].cpu().numpy(), rtol=1e-3, what="best render")


def test_pos_mlp_phase_matches_torch_composition():
    """f2 in the loop: PosMlpBrdfPhase (maps from the residual MLP, render/loss/backward in libmatpbr.so, gradients handed back
    to torch) against the same iteration composed from torch ops around the autograd render (inverse_img_w_mi.py:493-554)."""
    import copy

    from materialist_amd import loop, ops, posmlp, render, synthetic

    dev = _cuda()
    H = W = 32          # 1024 points: a square image for PosMLP
    spp = 8
    sc = synthetic.make_scene(9, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp).clone()
    a0, r0, m0 = _t(sc.init_albedo, dev), _t(sc.init_roughness, dev), _t(sc.init_metallic, dev)
    start_arm = torch.cat([a0.reshape(-1, 3), r0.reshape(-1, 1), m0.reshape(-1, 1)], -1).clamp(0, 1)
    net_a = posmlp.brdf_net("arm", hidden=(64, 64, 64, 64)).to(dev)
    net_a.lin4.weight.data.normal_(0, 0.02)
    net_b = copy.deepcopy(net_a)
    ph = loop.PosMlpBrdfPhase(scene, gt, net_b, start_arm, {"albedo": a0, "roughness": r0, "metallic": m0}, optimize_part="rm", spp=spp)
    opt = torch.optim.AdamW(net_a.parameters(), lr=3e-4)
    orig = {"roughness": start_arm[:, 3:4].reshape(H, W, 1), "metallic": start_arm[:, 4:5].reshape(H, W, 1)}
    for it in range(3):
        arm = net_a(start_arm)
        r = (arm[:, 3:4] * 0.93 + 0.07).clamp(0, 1).reshape(H, W, 1)
        m = arm[:, 4:5].clamp(0, 1).reshape(H, W, 1)
        pred = render.render_w_brdf(scene, a0, r, m, None, spp)
        total, mse, _, _ = loop._loss.brdf_loss(pred, gt, {"roughness": r, "metallic": m}, orig, 0.1)
        total.backward()
        opt.step()
        opt.zero_grad()
        ph.step()
        assert float(ph.stats[0, ops.STAT_MSE]) == pytest.approx(float(mse.detach()), rel=3e-4), it
        assert float(ph.stats[0, ops.STAT_LOSS]) == pytest.approx(float(total.detach()), rel=3e-4), it
    for (ka, va), (kb, vb) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
        assert (va - vb).abs().max().item() < 2e-5, ka


@pytest.mark.parametrize("part", ["rm", "a", "arm"])
def test_arm_mlp_phase_matches_the_torch_composition(part):
    """f2 in the loop without a framework in between: ArmMlpPhase (every launch of the iteration a kernel of libmatpbr.so: sine
    layers, output layer + tanh head, head backward, skinny weight gradients, AdamW on the flat buffer) against the same iterations
    composed from torch ops around the autograd render with the reference network (inverse_img_w_mi.py:470-554): loss values and
    every parameter after three AdamW steps, for the three parts of --opt_order."""
    import copy

    from materialist_amd import loop, ops, posmlp, render, synthetic
    from materialist_amd.armhead import ArmMlpPhase

    dev = _cuda()
    H = W = 128         # 16384 points: above MIN_ROWS, so the image-size kernels are the ones that run
    # spp 64 (the reference's; 5 x 4 specular nodes).  At spp 8 the specular rule has TWO nodes per pixel: a node that crosses the horizon
    # when the roughness moves in its last bit changes that pixel's d_r by a finite amount, and two forward passes that differ by one ulp
    # (measured round 5: sincos_cw against sin_packed) differ by 0.6-1.1 % in the output layer's roughness-bias gradient -- the size of the bound below
    spp = 64
    sc = synthetic.make_scene(9, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp).clone()
    # initial maps squeezed into [0.1, 0.9]: no map value sits on a clamp bound during these iterations (see the note at the end)
    a0, r0, m0 = (0.1 + 0.8 * _t(v, dev).clamp(0, 1) for v in (sc.init_albedo, sc.init_roughness, sc.init_metallic))
    start_arm = torch.cat([a0.reshape(-1, 3), r0.reshape(-1, 1), m0.reshape(-1, 1)], -1).clamp(0, 1)
    torch.manual_seed(11)
    net_a = posmlp.brdf_net("arm").to(dev)
    net_a.lin4.weight.data.normal_(0, 0.02)
    net_a.lin4.bias.data.normal_(0, 0.02)
    net_b = copy.deepcopy(net_a)
    fixed = {"albedo": a0, "roughness": r0, "metallic": m0}
    assert ArmMlpPhase.supported(scene, gt, net_b, part, None)
    ph = loop.pos_mlp_brdf_phase(scene, gt, net_b, start_arm, fixed, optimize_part=part, spp=spp)
    assert isinstance(ph, ArmMlpPhase)
    opt = torch.optim.AdamW(net_a.parameters(), lr=3e-4)
    orig = {"albedo": start_arm[:, 0:3].reshape(H, W, 3), "roughness": start_arm[:, 3:4].reshape(H, W, 1),
            "metallic": start_arm[:, 4:5].reshape(H, W, 1)}
    posmlp._PosMlpHipFn.MIN_ROWS = 1 << 30             # the reference side on the torch/BLAS composition of the network
    try:
        for it in range(3):
            arm = net_a(start_arm)
            maps = {"albedo": arm[:, 0:3].clamp(0, 1).reshape(H, W, 3), "roughness": (arm[:, 3:4] * 0.93 + 0.07).clamp(0, 1).reshape(H, W, 1),
                    "metallic": arm[:, 4:5].clamp(0, 1).reshape(H, W, 1)}
            live = {k: maps[k] for k, c in (("albedo", "a"), ("roughness", "r"), ("metallic", "m")) if c in part}
            use = {k: live.get(k, fixed[k]) for k in maps}
            pred = render.render_w_brdf(scene, use["albedo"], use["roughness"], use["metallic"], None, spp)
            total, mse, _, _ = loop._loss.brdf_loss(pred, gt, live, {k: orig[k] for k in live}, 0.1)
            total.backward()
            if it == 0:
                g_ref = [p.grad.clone() for p in net_a.parameters()]
            opt.step()
            opt.zero_grad()
            ph.step()
            rel = 3e-4 if it == 0 else 2e-3                    # later iterations carry the AdamW-amplified last-bit noise (note at the end)
            assert float(ph.stats[0, ops.STAT_MSE]) == pytest.approx(float(mse.detach()), rel=rel), it
            assert float(ph.stats[0, ops.STAT_LOSS]) == pytest.approx(float(total.detach()), rel=rel), it
            if it == 0:                                        # the gradient of every parameter, before AdamW normalises it
                for l, (gw, gb) in enumerate(ph.gviews):
                    rw, rb = g_ref[2 * l], g_ref[2 * l + 1]
                    # (the L1 terms of the loss carry sign(pred - gt): a few pixels flip between two renders that differ in the last bits;
                    # test_arm_mlp_phase_network_gradients_match_autograd pins the network half to 2e-5)
                    assert (gw[:, :rw.shape[1]] - rw).norm().item() <= 1e-2 * rw.norm().item() + 1e-9, l
                    assert (gb[:rb.shape[0]] - rb).norm().item() <= 1e-2 * rb.norm().item() + 1e-9, l
    finally:
        posmlp._PosMlpHipFn.MIN_ROWS = 8192
    # (Where a map saturates, the value of the reference's straight-through clamp, (clamp(x) + x) - x in fp32, lands on 1 or on
    # 1 + 2^-23 depending on the last bit of x, and the clamp of :494-496 gates the gradient on that: pixels contribute or not in
    # either implementation, AdamW normalises, and parameters with gradients of the size of that noise move by up to lr per step in
    # either direction.  That is a property of the reference's formulation, kept out of this comparison by the squeezed maps.)
    sd_a, sd_b = net_a.state_dict(), net_b.state_dict()
    assert list(sd_a) == list(sd_b)
    for k in sd_a:
        assert sd_a[k].shape == sd_b[k].shape
        diff = (sd_a[k] - sd_b[k]).abs()
        # AdamW normalises: the few parameters whose gradient is of the size of the render's last-bit noise may step the other way
        # (a wrong gradient moves every parameter by ~lr per step: mean ~ 5e-4)
        assert diff.mean().item() < 3e-5 and (diff < 3e-5).float().mean().item() > 0.85 and diff.max().item() <= 6.1 * 3e-4, (k, diff.max().item(), diff.mean().item())
    bw = ph.best_weights
    assert set(bw) == set(sd_b) and all(bw[k].shape == sd_b[k].shape for k in bw)


def test_use_mask_in_the_launch_by_launch_pos_mlp_phase():
    """`--use_mask` (inverse_img_w_mi.py:509-511) in ArmMlpPhase: `matpbr_masked_mean_fill` forward and backward against torch autograd
    through `x.clamp(0, 1)`, `x[mask] = x[mask].mean()`, then three iterations of the phase against the autograd composition
    (`loop.PosMlpBrdfPhase` with the same mask): losses, and maps that are uniform inside the mask."""
    import copy

    from materialist_amd import loop, ops, posmlp, render, synthetic
    from materialist_amd.armhead import ArmMlpPhase

    dev = _cuda()
    H = W = 128
    gen = torch.Generator(device="cpu").manual_seed(3)
    mask = torch.zeros(H, W, dtype=torch.bool)
    mask[20:90, 30:100] = True
    mask[5:9, 5:9] = True
    mk = mask.to(dev)
    x = (torch.rand(H, W, 1, generator=gen) * 1.6 - 0.3).to(dev).requires_grad_(True)       # some entries outside [0, 1]
    y_ref = loop.masked_mean_fill(x.clamp(0, 1), mk)
    gy = torch.randn(H, W, 1, generator=gen).to(dev)
    y_ref.backward(gy)
    u8 = mk.to(torch.uint8).contiguous()
    y = ops.masked_mean_fill(x.detach(), u8)
    assert torch.equal(y[~mk], x.detach()[~mk])                                           # unmasked entries pass through (the kernels clamp them)
    assert (y[mk] - y_ref.detach()[mk]).abs().max().item() < 1e-6
    gx = ops.masked_mean_fill(gy, u8, gate=x.detach())
    inside = ((x.detach() >= 0) & (x.detach() <= 1)).reshape(H, W)
    assert (gx[mk] - x.grad[mk]).abs().max().item() < 1e-6 * float(gy.abs().max())
    assert torch.equal(gx[~mk], gy[~mk]) and float(gx[mk & ~inside].abs().max()) == 0.0
    # ---- the phase
    spp = 8
    sc = synthetic.make_scene(9, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp).clone()
    a0, r0, m0 = (0.1 + 0.8 * _t(v, dev).clamp(0, 1) for v in (sc.init_albedo, sc.init_roughness, sc.init_metallic))
    start_arm = torch.cat([a0.reshape(-1, 3), r0.reshape(-1, 1), m0.reshape(-1, 1)], -1).clamp(0, 1)
    torch.manual_seed(11)
    net_a = posmlp.brdf_net("arm").to(dev)
    net_a.lin4.weight.data.normal_(0, 0.02)
    net_a.lin4.bias.data.normal_(0, 0.02)
    net_b = copy.deepcopy(net_a)
    fixed = {"albedo": a0, "roughness": r0, "metallic": m0}
    for part in ("rm", "a"):
        assert ArmMlpPhase.supported(scene, gt, net_b, part, mk)
        ph = loop.pos_mlp_brdf_phase(scene, gt, net_b, start_arm, fixed, optimize_part=part, spp=spp, mask=mk)
        assert isinstance(ph, ArmMlpPhase)
        ref = loop.PosMlpBrdfPhase(scene, gt, net_a, start_arm, fixed, optimize_part=part, spp=spp, mask=mk)
        for it in range(3):
            ref.step()
            ph.step()
            rel = 3e-4 if it == 0 else 2e-3
            assert float(ph.stats[0, ops.STAT_MSE]) == pytest.approx(float(ref.stats[0, ops.STAT_MSE]), rel=rel), (part, it)
            assert float(ph.stats[0, ops.STAT_LOSS]) == pytest.approx(float(ref.stats[0, ops.STAT_LOSS]), rel=rel), (part, it)
        for k in ("roughness", "metallic"):
            inside_vals = ph.best[k].reshape(H, W)[mk]
            assert float(inside_vals.max() - inside_vals.min()) == 0.0, (part, k)         # one value inside the mask
            assert (ph.best[k] - ref.best[k]).abs().max().item() < 2e-4, (part, k)
        cm = ph.current_maps()
        assert float(cm["roughness"].reshape(H, W)[mk].std()) == 0.0


@pytest.mark.parametrize("part", ["rm", "a", "arm"])
def test_use_mask_none_mode_phase_matches_the_torch_composition(part):
    """`--model_name none --use_mask` launch by launch on the C ABI (`loop.MaskedBrdfPhase`) against the autograd composition
    (`loop.BrdfPhase` with the same mask: torch clamps, masked means, torch losses, torch.optim.Adam): per-iteration statistics, the maps
    after the iterations, one value inside the mask."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H, W, spp = 96, 128, 16
    sc = synthetic.make_scene(14, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp).clone()
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    yy, xx = torch.meshgrid(torch.linspace(0, 6.0, H, device=dev), torch.linspace(0, 9.0, W, device=dev), indexing="ij")
    # a roughness that varies inside the mask (with a constant one the regulariser |mean - r0| sits on its kink in the first iteration and
    # its sign is rounding noise), some values beyond the clamp bounds: their gates must close
    init[1] = (init[1] * 1.3 - 0.1 + 0.35 * (torch.sin(yy) * torch.cos(xx)).unsqueeze(-1)).contiguous()
    if "r" not in part:
        init[1] = init[1].clamp(0.07, 1)                  # a map that is not optimised comes from the stage before, inside its bounds
    mask = torch.zeros(H, W, dtype=torch.bool, device=dev)
    mask[10:70, 20:100] = True
    ref = loop.BrdfPhase(scene, gt, *init, None, optimize_part=part, spp=spp, mask=mask)
    ph = loop.MaskedBrdfPhase(scene, gt, *init, mask, optimize_part=part, spp=spp)
    for it in range(6):
        mse_ref = ref.step()
        ph.step()
        assert float(ph.stats[0, ops.STAT_MSE]) == pytest.approx(float(mse_ref), rel=5e-4), (part, it)
        assert float(ph.stats[0, ops.STAT_LOSS]) == pytest.approx(float(ref.last["loss"]), rel=5e-4), (part, it)
    cur, cur_ref = ph.current_maps(), ref.current_maps()
    for k in ("albedo", "roughness", "metallic"):
        d = (cur[k] - cur_ref[k].detach().reshape(cur[k].shape)).abs()
        free = d if k == "albedo" else d.reshape(H, W)[~mask]
        # Adam normalises: where a gradient is all but zero its sign may differ between the two renders' last bits
        assert float((free < 5e-5).float().mean()) > 0.99 and float(free.mean()) < 1e-5, (part, k, float(free.max()))
    for k in ("roughness", "metallic"):
        inside = cur[k].reshape(H, W)[mask]
        assert float(inside.max() - inside.min()) == 0.0
        assert abs(float(inside[0]) - float(cur_ref[k].detach().reshape(H, W)[mask][0])) < 5e-5, (part, k)
    assert int(ph.poll()["iters"][0]) == 6


@pytest.mark.gpu
def test_use_mask_on_a_batch_is_its_images_alone():
    """`--use_mask` on a batch (`loop.MaskedBatchPhase`: one MaskedBrdfPhase per image on a stream of its own, per-image masked means, SaveBest and
    EarlyStopping): the same bits as the images run alone, and the route `optimize_envmap_ARMN` takes for a batch with masks."""
    from materialist_amd import loop, ops, optimize, render, synthetic

    dev = _cuda()
    H, W, spp, B = 64, 96, 16, 3
    scs = [synthetic.make_scene(20 + b, H, W) for b in range(B)]
    st = lambda k: torch.from_numpy(np.stack([np.ascontiguousarray(getattr(s_, k), dtype=np.float32) for s_ in scs])).to(dev)
    scene = render.load_estimated_mesh(st("depth"), use_mesh_normal=True)
    scene._set("emitter.data", st("light"))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, st("albedo"), st("roughness"), st("metallic"), None, spp).clone()
    init = [st("init_albedo"), st("init_roughness"), st("init_metallic")]
    mask = torch.zeros(B, H, W, dtype=torch.bool, device=dev)
    mask[0, 8:40, 10:70] = True
    mask[1, 30:60, 0:50] = True                              # (image 2: no masked pixel at all)
    ph = loop.MaskedBatchPhase(scene, gt, *init, mask, optimize_part="rm", spp=spp, history_len=8)
    ph.run(5)
    cur, best, st_b = ph.current_maps(), ph.best, ph.stats
    for b in range(B):
        sc1 = render.load_estimated_mesh(st("depth")[b], use_mesh_normal=True)
        sc1._set("emitter.data", st("light")[b])
        one = loop.MaskedBrdfPhase(sc1, gt[b], *(x[b] for x in init), mask[b], optimize_part="rm", spp=spp, history_len=8)
        one.run(5)
        assert torch.equal(st_b[b], one.stats[0]), b
        for k in ("albedo", "roughness", "metallic"):
            assert torch.equal(cur[k][b], one.current_maps()[k]) and torch.equal(best[k][b], one.best[k]), (b, k)
        assert torch.equal(ph.pred[b], one.pred) and torch.equal(ph.history()[:, b], one.history()[:, 0])
    inside = cur["roughness"][0].reshape(H, W)[mask[0]]
    assert float(inside.max() - inside.min()) == 0.0 and int(ph.poll()["iters"].min()) == 5
    # the whole optimisation on the batch takes this route and keeps the maps uniform inside every image's mask
    mat = {"albedo": init[0], "roughness": init[1], "metallic": init[2], "gt_image": gt, "mask": mask}
    scene2 = render.load_estimated_mesh(st("depth"), use_mesh_normal=True)
    res = optimize.optimize_envmap_ARMN(scene2, mat, optimize_order=["rm"], spp=spp, num_epochs=12, model_name="none", use_mask=True, log=lambda *_: None)
    for b in range(2):
        for k in ("roughness", "metallic"):
            v = res[k][b].reshape(H, W)[mask[b]]
            assert float(v.max() - v.min()) == 0.0, (b, k)
    assert len(res["psnr_per_image"]) == B


@pytest.mark.parametrize("packed", [True, False])
def test_output_layer_backward_in_one_pass_equals_the_separate_kernels(packed):
    """matpbr_mlp_out_layer_bwd (weight / bias gradient of the 5-output layer, dL/d pre of the last sine layer and its bias gradient in one
    pass over that layer's sines) against matpbr_mlp_skinny_bwd_weight + matpbr_mlp_layer_bwd_input on the same operands, and against fp64."""
    from materialist_amd import ops

    dev = _cuda()
    M, n_prev = 128 * 128 + 4 * 37, 241                    # ragged tiles and slabs, a skip layer's width
    gen = torch.Generator(device="cpu").manual_seed(5)
    pre = (torch.randn(M, 256, generator=gen) * 3).to(dev)
    sin, cos = torch.sin(pre), torch.cos(pre)
    s_buf = torch.empty(M, 256, device=dev)
    if packed:                                             # sines that carry the sign of their cosine in the last mantissa bit, as the forward kernels store them
        bits = (sin.view(torch.int32) & ~1) | (cos < 0).to(torch.int32)
        s_buf.copy_(bits.view(torch.float32))
    else:
        s_buf.copy_(sin)
    d_x = torch.zeros(M, 8, device=dev)
    d_x[:, :5] = torch.randn(M, 5, generator=gen).to(dev)
    w_out = (torch.randn(5, 256, generator=gen) * 0.1).to(dev)
    g_new, gw_new, gb_new, gbp_new = torch.empty(M, 256, device=dev), torch.empty(5, 256, device=dev), torch.empty(8, device=dev), torch.empty(256, device=dev)
    ops.mlp_out_layer_bwd(d_x, s_buf, None if packed else cos.contiguous(), w_out, g_new, gw_new, gb_new, gbp_new, 5, n_prev)
    # the separate kernels
    g_old, gw_old, gb_old, gbp_old = torch.empty(M, 256, device=dev), torch.empty(5, 256, device=dev), torch.empty(8, device=dev), torch.empty(256, device=dev)
    ops.mlp_skinny_bwd_weight(d_x, s_buf, gw_old, 5, 256, d_bias=gb_old)
    w_t = torch.zeros(256, 8, device=dev)
    w_t[:, :5] = w_out.t()
    ops.mlp_layer_bwd_input(d_x, w_t, s_buf if packed else cos.contiguous(), g_old, n_prev, 5, gbp_old, packed=packed)
    assert torch.equal(gw_new, gw_old) and torch.equal(gb_new[:5], gb_old[:5])          # the same pass, the same order
    ref_g = (d_x[:, :5].double() @ w_out.double()) * cos.double()
    scale = float(ref_g.abs().max())
    tol = 3e-6 if not packed else 2e-4                     # packed: |cos| rebuilt from 1 - s^2 (error ~1.5e-7 / |cos|)
    assert (g_new[:, :n_prev].double() - ref_g[:, :n_prev]).abs().max().item() <= tol * scale
    assert (g_new[:, :n_prev] - g_old[:, :n_prev]).abs().max().item() <= tol * scale
    ref_b = ref_g[:, :n_prev].sum(0)
    assert (gbp_new[:n_prev].double() - ref_b).abs().max().item() <= 1e-5 * float(ref_b.abs().max()) + 5 * tol * scale * M ** 0.5   # a sum of M rounded terms
    assert (gbp_new[:n_prev] - gbp_old[:n_prev]).abs().max().item() <= 1e-4 * float(ref_b.abs().max())


@pytest.mark.parametrize("packed", [True, False])
def test_first_layer_backward_without_its_pre_activation_gradient_in_memory(packed):
    """matpbr_mlp_first_layer_bwd_bx (the first layer's weight and bias gradient from the epilogue of the split-operand input-gradient
    kernel above it) against matpbr_mlp_layer_bwd_input_bx + matpbr_mlp_skinny_bwd_weight on the same operands, and against fp64."""
    from materialist_amd import ops

    dev = _cuda()
    M, n0, n_red, d0 = 128 * 131, 256, 256, 15
    gen = torch.Generator(device="cpu").manual_seed(9)
    g = torch.randn(M, 256, generator=gen).to(dev)
    w1 = (torch.randn(n_red, 256, generator=gen) / 16).to(dev)          # the second layer's forward weight [n_red, K = n0]
    pre0 = (torch.randn(M, 256, generator=gen) * 3).to(dev)
    sin, cos = torch.sin(pre0), torch.cos(pre0)
    c_op = torch.empty(M, 256, device=dev)
    if packed:
        c_op.copy_(((sin.view(torch.int32) & ~1) | (cos < 0).to(torch.int32)).view(torch.float32))
    else:
        c_op.copy_(cos)
    x0 = torch.zeros(M, 16, device=dev)
    x0[:, :d0] = torch.randn(M, d0, generator=gen).to(dev)
    wts = ops.mlp_split_weights(w1, n0, n_red, transposed=True)          # (W1[:, :n0])^T as the input-gradient operand
    gw_new, gb_new = torch.zeros(n0, 16, device=dev), torch.empty(n0, device=dev)
    ops.mlp_first_layer_bwd_bx(g, wts, c_op, x0, gw_new, d0, n0, n_red, gb_new, 6, packed=packed)
    g0, gb_old, gw_old = torch.empty(M, 256, device=dev), torch.empty(n0, device=dev), torch.zeros(n0, 16, device=dev)
    ops.mlp_layer_bwd_input_bx(g, wts, c_op, g0, n0, n_red, gb_old, 6, packed=packed)
    ops.mlp_skinny_bwd_weight(x0, g0, gw_old, d0, n0, transposed_out=True)
    # the same column sums, grouped by 512-thread workgroups here and by 256-thread ones in the stand-alone kernel's default loop
    assert (gb_new - gb_old).abs().max().item() <= 4e-6 * (gb_old.abs().max().item() + 1.0) * (M / 4096) ** 0.5
    from materialist_amd import _lib
    was = _lib.load().matpbr_mlp_set_lds_dma(1)
    try:
        gb_one = torch.empty(n0, device=dev)
        ops.mlp_layer_bwd_input_bx(g, wts, c_op, torch.empty(M, 256, device=dev), n0, n_red, gb_one, 6, packed=packed)
        torch.cuda.synchronize()
    finally:
        _lib.load().matpbr_mlp_set_lds_dma(was)
    assert torch.equal(gb_new, gb_one)                                   # the same epilogue sums in the same order
    ref_g0 = (g.double() @ w1.double()) * cos.double()
    ref_w = ref_g0.t() @ x0.double()[:, :d0]
    scale = float(ref_w.abs().max())
    assert (gw_new[:, :d0].double() - ref_w).abs().max().item() <= 2e-5 * scale
    assert (gw_new[:, :d0] - gw_old[:, :d0]).abs().max().item() <= 2e-5 * scale
    assert float(gw_new[:, d0:].abs().max()) == 0.0                      # the padding column is never written


def test_arm_mlp_phase_network_gradients_match_autograd():
    """The network half of ArmMlpPhase on its own, where nothing is chaotic: forward() against the reference module's maps, and
    backward() fed with given map gradients against torch autograd through the reference network (the straight-through clamp has
    an identity gradient; the gating clamp of :494-496 lives in the loss kernels, not here)."""
    import copy

    from materialist_amd import loop, posmlp, render, synthetic
    from materialist_amd.armhead import ArmMlpPhase

    dev = _cuda()
    H = W = 128
    sc = synthetic.make_scene(5, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    gt = torch.rand(H, W, 3, device=dev)
    a0, r0, m0 = _t(sc.init_albedo, dev), _t(sc.init_roughness, dev), _t(sc.init_metallic, dev)
    start_arm = torch.cat([a0.reshape(-1, 3), r0.reshape(-1, 1), m0.reshape(-1, 1)], -1).clamp(0, 1)
    torch.manual_seed(4)
    net_a = posmlp.brdf_net("arm").to(dev)
    net_a.lin4.weight.data.normal_(0, 0.05)
    net_a.lin4.bias.data.normal_(0, 0.05)
    net_b = copy.deepcopy(net_a)
    ph = ArmMlpPhase(scene, gt, net_b, start_arm, {"albedo": a0, "roughness": r0, "metallic": m0}, optimize_part="arm", spp=8)
    maps = ph.forward()
    posmlp._PosMlpHipFn.MIN_ROWS = 1 << 30             # the reference side on the torch/BLAS composition of the network
    try:
        arm = net_a(start_arm)
        ref = {"albedo": arm[:, 0:3].reshape(H, W, 3), "roughness": (arm[:, 3:4] * 0.93 + 0.07).reshape(H, W, 1), "metallic": arm[:, 4:5].reshape(H, W, 1)}
        for k in ref:                                      # one ulp of the straight-through clamp's (c + x) - x on top of the network's rounding
            assert (maps[k] - ref[k].detach()).abs().max().item() <= 3e-6, k
        g = {k: torch.randn_like(v) for k, v in ref.items()}
        torch.autograd.backward([ref[k] for k in ref], [g[k] for k in ref])
    finally:
        posmlp._PosMlpHipFn.MIN_ROWS = 8192
    for k in g:
        ph.g[k].copy_(g[k])
    ph.backward()
    g_ref = [p.grad for p in net_a.parameters()]
    for l, (gw, gb) in enumerate(ph.gviews):
        rw, rb = g_ref[2 * l], g_ref[2 * l + 1]
        assert (gw[:, :rw.shape[1]] - rw).abs().max().item() <= 3e-4 * rw.abs().max().item(), l
        assert (gw[:, :rw.shape[1]] - rw).norm().item() <= 2e-5 * rw.norm().item(), l
        assert (gw[:, rw.shape[1]:] == 0).all()            # the padding column of the first layer never receives a gradient
        assert (gb[:rb.shape[0]] - rb).norm().item() <= 2e-5 * rb.norm().item(), l


def test_arm_mlp_phase_schedule_snapshot_and_early_stopping():
    """The host-visible behaviour of ArmMlpPhase around the kernels: StepLR(100, 0.8) stepped only while lr > 1.5e-4
    (inverse_img_w_mi.py:471,553-554) against torch's scheduler, SaveBest's weight snapshot = the weights that PRODUCED the best
    render (:546-547; taken before the optimiser step), `load_state_dict` of that snapshot into the live module, and the host
    EarlyStopping of `step_and_check` (:550)."""
    from materialist_amd import loop, ops, posmlp, render, synthetic
    from materialist_amd.armhead import ArmMlpPhase

    dev = _cuda()
    H = W = 96
    sc = synthetic.make_scene(2, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, 8).clone()
    a0, r0, m0 = _t(sc.init_albedo, dev), _t(sc.init_roughness, dev), _t(sc.init_metallic, dev)
    start_arm = torch.cat([a0.reshape(-1, 3), r0.reshape(-1, 1), m0.reshape(-1, 1)], -1).clamp(0, 1)
    torch.manual_seed(1)
    net = posmlp.brdf_net("arm").to(dev)
    ph = ArmMlpPhase(scene, gt, net, start_arm, {"albedo": a0, "roughness": r0, "metallic": m0}, optimize_part="rm", spp=8, patience=0)
    # the learning-rate trajectory against torch's StepLR driven by the reference's rule
    probe = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([probe], lr=3e-4)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=100, gamma=0.8)
    first_mse, flat_before_best, best_seen = None, None, float("inf")
    for it in range(520):
        flat_now = ph.flat.clone() if it < 40 else None
        ph.step()
        opt.step()
        if opt.param_groups[0]["lr"] > 1.5e-4:
            sched.step()
        assert ph.opt.param_groups[0]["lr"] == pytest.approx(opt.param_groups[0]["lr"], rel=1e-12), it
        if it % 100 == 99 or it < 3:
            assert float(ph.hyper[0]) == pytest.approx(opt.param_groups[0]["lr"], rel=1e-6), it
        if it < 40:                                        # snapshot semantics on the first iterations (one host sync each)
            mse = float(ph.stats[0, ops.STAT_MSE])
            first_mse = mse if first_mse is None else first_mse
            if mse < best_seen:
                best_seen, flat_before_best = mse, flat_now
            assert torch.equal(ph._best_flat, flat_before_best), it
    assert opt.param_groups[0]["lr"] == pytest.approx(3e-4 * 0.8 ** 4)       # 1.2288e-4: the schedule has stopped
    assert float(ph.hyper[1]) == 520.0
    assert float(ph.stats[0, ops.STAT_BEST]) < first_mse
    bw = ph.best_weights
    net.load_state_dict(bw)                                # what optimize_envmap_ARMN does after every part (:586-587)
    for k, v in net.state_dict().items():
        assert torch.equal(v, bw[k]), k
    assert torch.equal(ph.flat, ph._best_flat)             # the parameters are still views of the flat buffer
    # EarlyStopping on the device: with a huge min_delta nothing counts as an improvement after the first iteration, so the state machine
    # fires in iteration 4 (which still completes, as the reference's loop does); the host polls every `sync_every` iterations and the
    # iterations enqueued in between are no-ops: statistics, snapshot and AdamW rest
    ph2 = loop.pos_mlp_brdf_phase(scene, gt, net, start_arm, {"albedo": a0, "roughness": r0, "metallic": m0}, optimize_part="rm", spp=8,
                                  patience=3, min_delta=1.0)
    assert isinstance(ph2, ArmMlpPhase) and ph2.sync_every == 8
    stops = [ph2.step_and_check() for _ in range(4)]
    flat4, best4, stats4 = ph2.flat.clone(), ph2._best_flat.clone(), ph2.stats.clone()
    stops += [ph2.step_and_check() for _ in range(4)]
    assert stops == [False] * 7 + [True]
    assert ph2.iterations_run == 4
    assert torch.equal(ph2.flat, flat4) and torch.equal(ph2._best_flat, best4)
    assert float(ph2.stats[0, ops.STAT_MSE]) == float(stats4[0, ops.STAT_MSE]) and float(ph2.stats[0, ops.STAT_STOPPED]) == 2.0
    host = loop.EarlyStopping(patience=3, min_delta=1.0)
    for v in ph2.history()[:4, 0].cpu().tolist():
        host(v)
    assert host.early_stop and (ph2.history()[4:8] == 0).all()


def test_last_sine_layer_with_the_head_in_its_epilogue():
    """matpbr_mlp_layer_fwd_bx_head = matpbr_mlp_layer_fwd_bx followed by matpbr_mlp_arm_head_fwd on its output: identical sines and
    cosines, tanh / maps equal up to the summation order of the five 256-term dot products."""
    from materialist_amd import ops

    dev = _cuda()
    torch.manual_seed(21)
    M = 128 * 300
    x = torch.randn(M, 256, device=dev)
    w = (torch.rand(256, 256, device=dev) * 2 - 1) / 16
    b = torch.randn(256, device=dev) * 0.1
    w_out = torch.randn(5, 256, device=dev) / 8
    b_out = torch.randn(5, device=dev) * 0.1
    start = torch.rand(M, 5, device=dev)
    ws = ops.mlp_split_weights(w, 256, 256)
    s0, c0 = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    th0 = torch.empty(M, 8, device=dev)
    a0, r0, m0 = torch.empty(M, 3, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.mlp_layer_fwd_bx(x, ws, b, s0, c0, 256, 256, 6)
    ops.mlp_arm_head_fwd(s0, w_out, b_out, start, th0, a0, r0, m0, 256)
    s1, c1 = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    th1 = torch.full((M, 8), float("nan"), device=dev)
    a1, r1, m1 = torch.full((M, 3), float("nan"), device=dev), torch.full((M,), float("nan"), device=dev), torch.full((M,), -1.0, device=dev)
    ops.mlp_layer_fwd_bx_head(x, ws, b, s1, c1, 256, 6, w_out, b_out, start, th1, a1, r1, None)
    assert torch.equal(s0, s1) and torch.equal(c0, c1)
    assert (th1[:, :5] - th0[:, :5]).abs().max().item() <= 2e-6
    assert (a1 - a0).abs().max().item() <= 3e-6 and (r1 - r0).abs().max().item() <= 3e-6
    assert (m1 == -1.0).all()                              # a map passed as None is not written
    ref = torch.tanh(s0.double() @ w_out.double().t() + b_out.double())
    assert (th1[:, :5].double() - ref).abs().max().item() <= 3e-6


def test_skinny_layers_and_arm_head_match_torch():
    """The skinny ends of the network at image size: output layer (J = 3, 5, 8) against an fp64 product, the 'arm' head against
    torch's (tanh, residual, clamp, 0.93 r + 0.07) bit for bit on the same pre-activations, its backward against autograd, the two
    skinny weight gradients (+ bias gradient) against fp64, and AdamW on a flat buffer against torch.optim.AdamW."""
    from materialist_amd import ops

    dev = _cuda()
    torch.manual_seed(2)
    M = 128 * 77 + 5
    x = torch.randn(M, 256, device=dev)
    for J in (3, 5, 8):
        w = torch.randn(J, 256, device=dev) / 16
        b = torch.randn(J, device=dev)
        out = torch.full((M, 8), float("nan"), device=dev)
        ops.mlp_skinny_fwd(x, w, b, out, 256)
        ref = x.double() @ w.double().t() + b.double()
        assert (out[:, :J].double() - ref).abs().max().item() < 2e-5
        assert torch.isnan(out[:, J:]).all()
    w = torch.randn(5, 256, device=dev) / 8
    b = torch.randn(5, device=dev) * 0.1
    start = torch.rand(M, 5, device=dev)
    pre = torch.empty(M, 8, device=dev)
    ops.mlp_skinny_fwd(x, w, b, pre, 256)
    th = torch.empty(M, 8, device=dev)
    ma, mr, mm = torch.empty(M, 3, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.mlp_arm_head_fwd(x, w, b, start, th, ma, mr, mm, 256)
    xs = pre[:, :5].clone().requires_grad_(True)
    y = 1.3 * torch.tanh(xs) + start
    y = y.clamp(0, 1).detach() + y - y.detach()
    a_t, r_t, m_t = y[:, 0:3], y[:, 3] * 0.93 + 0.07, y[:, 4]
    assert (th[:, :5] - torch.tanh(xs.detach())).abs().max().item() <= 2e-7
    assert (ma - a_t.detach()).abs().max().item() <= 3e-7 and (mr - r_t.detach()).abs().max().item() <= 3e-7
    assert (mm - m_t.detach()).abs().max().item() <= 3e-7
    mr_only = torch.full((M,), -1.0, device=dev)
    ops.mlp_arm_head_fwd(x, w, b, start, th, None, mr_only, None, 256)      # maps that are not optimised are not written
    assert torch.equal(mr_only, mr)
    ga, gr, gm = torch.randn(M, 3, device=dev), torch.randn(M, device=dev), torch.randn(M, device=dev)
    ((a_t * ga).sum() + (r_t * gr).sum() + (m_t * gm).sum()).backward()
    d_x = torch.full((M, 8), float("nan"), device=dev)
    ops.mlp_arm_head_bwd(ga, gr, gm, th, d_x)
    assert (d_x[:, :5] - xs.grad).abs().max().item() <= 2e-6 * xs.grad.abs().max().item()
    assert (d_x[:, 5:] == 0).all()
    ops.mlp_arm_head_bwd(None, gr, gm, th, d_x)
    assert (d_x[:, 0:3] == 0).all() and (d_x[:, 3:5] - xs.grad[:, 3:5]).abs().max().item() <= 2e-6 * xs.grad.abs().max().item()
    # skinny weight gradients
    ops.mlp_arm_head_bwd(ga, gr, gm, th, d_x)
    gw, gb = torch.full((5, 256), float("nan"), device=dev), torch.full((8,), float("nan"), device=dev)
    ops.mlp_skinny_bwd_weight(d_x, x, gw, 5, 256, d_bias=gb)
    ref = d_x[:, :5].double().t() @ x.double()
    assert (gw.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() * (M / 4096) ** 0.5
    refb = d_x[:, :5].double().sum(0)
    assert (gb[:5].double() - refb).abs().max().item() <= 2e-6 * (refb.abs().max().item() + d_x.abs().max().item() * M ** 0.5)
    x0p = torch.zeros(M, 16, device=dev)
    x0p[:, :15] = torch.randn(M, 15, device=dev)
    x0p[:, 0] = torch.arange(M, device=dev) % 512
    g1 = torch.randn(M, 256, device=dev)
    g1[:, 241:] = float("nan")                         # columns beyond the layer's width hold scratch
    gw0 = torch.zeros(241, 16, device=dev)
    ops.mlp_skinny_bwd_weight(x0p, g1, gw0, 15, 241, transposed_out=True)
    ref = g1[:, :241].double().t() @ x0p[:, :15].double()
    assert (gw0[:, :15].double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() * (M / 4096) ** 0.5
    assert (gw0[:, 15] == 0).all()
    # AdamW on the flat buffer
    p0 = torch.randn(5000, device=dev)
    p_t = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p_t], lr=3e-4)
    p_h, m_h, v_h = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    hyper = torch.tensor([3e-4, 0.0], device=dev)
    for it in range(4):
        g = torch.randn(5000, device=dev)
        p_t.grad = g.clone()
        opt.step()
        ops.adamw_step_dev(p_h, g, m_h, v_h, hyper, 0.01)
    assert float(hyper[1]) == 4.0
    assert (p_h - p_t.detach()).abs().max().item() <= 5e-7          # p * (1 - lr wd) - update in one fma against torch's two roundings


def test_light_kinds_sh9_and_env_texels():
    """SURVEY 8b's other light parameterisations: SH9 (bands 0..2) and the equirectangular texel map that is `emitter.data` of the
    reference scene -> the SH25 light of the kernels, against the host-side projection matrix (fp64), with the gradient back."""
    from materialist_amd import ops, sh

    dev = _cuda()
    torch.manual_seed(0)
    for He in (16, 8):
        env = torch.rand(2, He * 2 * He, 3, device=dev, requires_grad=True)
        coef = ops.light_to_sh25(env, ops.LIGHT_ENV_TEXELS)
        P = torch.from_numpy(sh.envmap_to_sh_matrix(He, 2 * He)).to(dev)                     # [25, T] float64
        ref = P @ env.detach().double()
        assert (coef.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
        g = torch.randn_like(coef)
        coef.backward(g)
        gref = P.t() @ g.double()
        assert (env.grad.double() - gref).abs().max().item() < 2e-5 * gref.abs().max().item()
    sh9 = torch.randn(3, 9, 3, device=dev, requires_grad=True)
    c9 = ops.light_to_sh25(sh9, ops.LIGHT_SH9)
    assert torch.equal(c9[:, :9], sh9.detach()) and (c9[:, 9:] == 0).all()
    c9.backward(torch.ones_like(c9))
    assert torch.equal(sh9.grad, torch.ones_like(sh9))
    with pytest.raises(Exception):
        ops.light_to_sh25(torch.rand(1, 100, 3, device=dev), ops.LIGHT_ENV_TEXELS)            # not He x 2He


def test_column_sum():
    from materialist_amd import ops

    dev = _cuda()
    for M, N in ((262144, 256), (1000, 241), (7, 5), (4096, 1024)):
        x = torch.randn(M, N, device=dev)
        got = ops.column_sum(x)
        ref = x.double().sum(0)
        assert (got.double() - ref).abs().max().item() <= 1e-5 * (x.abs().double().sum(0).max().item())
        assert torch.equal(got, ops.column_sum(x))      # fixed-order reduction: bit-reproducible


def test_transfer_relight_equals_direct_render():
    """f4: the render is linear in the light; relighting through the precomputed transfer reproduces shade_fwd, and the SH
    y-rotation reproduces rolling the envmap columns (render_final.py:290-298)."""
    from materialist_amd import ops, sh, synthetic

    dev = _cuda()
    H, W, spp = 40, 56, 16
    sc, n = _scene_arrays(H, W, image_id=10)
    a, r, m, nn = (_t(x, dev) for x in (sc.albedo, sc.roughness, sc.metallic, n))
    T = ops.shade_transfer(a, r, m, nn, spp)
    rng = np.random.default_rng(2)
    env = rng.random((16, 32, 3)) + 0.1
    P = sh.envmap_to_sh_matrix(16, 32)
    lights = []
    for shift in range(11):                      # more than one 8-light pass
        lights.append(P @ np.roll(env, shift, axis=1).reshape(512, 3))
    L = _t(np.stack(lights), dev)
    out = ops.relight(T, L, H, W)
    for f in (0, 3, 10):
        direct = ops.shade_fwd(a, r, m, nn, L[f].contiguous(), spp)
        assert (out[f] - direct).abs().max().item() <= 2e-5 * direct.abs().max().item()
    # rolling the envmap by whole columns == rotating the SH light about +y
    c0 = P @ env.reshape(512, 3)
    for shift in (1, 5):
        rot = sh.rotate_y_matrix(2 * np.pi * shift / 32) @ c0
        np.testing.assert_allclose(rot, lights[shift], atol=1e-10)


@pytest.mark.parametrize("hw", [(96, 131), (33, 47), (5, 7)])
def test_image_sizes_that_leave_the_last_wave_partly_empty(hw):
    """Image sizes whose pixel count is not a multiple of 128 (two pixels per lane, 64 lanes): the kernels that read the quadrature table
    out of every lane's registers (diffuse cache, cached-diffuse render, radiance transfer) must keep the lanes beyond the image alive --
    with a divergent early exit their table entries were never loaded and the last pixels of the image came out NaN (round 4 finding)."""
    from materialist_amd import ops, synthetic

    dev = _cuda()
    H, W = hw
    spp = 64
    sc, n = _scene_arrays(H, W, image_id=12)
    a, r, m, nn, light = (_t(x, dev) for x in (sc.albedo, sc.roughness, sc.metallic, n, sc.light))
    direct = ops.shade_fwd(a, r, m, nn, light, spp)
    dcache = ops.diffuse_cache(nn, light, spp)
    assert bool(torch.isfinite(dcache).all())
    jac = ops.plane9(a)
    cached = ops.shade_fwd(a, r, m, nn, light, spp, dcache=dcache, jac=jac)
    assert bool(torch.isfinite(jac).all())
    assert (cached - direct).abs().max().item() <= 2e-5 * direct.abs().max().item()
    T = ops.shade_transfer(a, r, m, nn, spp)
    out = ops.relight(T, light.reshape(1, 25, 3).contiguous(), H, W)
    assert (out[0] - direct).abs().max().item() <= 2e-5 * direct.abs().max().item()


def test_materialnet_runs_on_the_gpu(tmp_path):
    """f3 on PyTorch-ROCm: random-weight MaterialNet through the pipeline's initial-guess path (SDPA attention on the GPU)."""
    from PIL import Image

    from materialist_amd import pipeline
    from materialist_amd.materialnet import MaterialNet

    dev = _cuda()
    net = MaterialNet().to(dev).eval()
    img = (np.random.default_rng(1).random((64, 80, 3)) * 255).astype(np.uint8)
    out = net.infer_image(img, input_size=140)
    assert out["albedo"].shape == (64, 80, 3) and np.isfinite(out["depth"]).all() and (out["roughness"] >= 0).all()
    wpath = str(tmp_path / "w.pth")
    torch.save(net.state_dict(), wpath)
    src = str(tmp_path / "in.png")
    Image.fromarray(img).save(src)
    res = pipeline.inverse_image(src, "mn_case", opt_src="arm", opt_order=["arm"], save_path=str(tmp_path), size=32, spp=8, num_epochs=4,
                                 sync_every=4, log=lambda *_: None, matnet_weights=wpath)
    assert os.path.exists(os.path.join(res["output_dir"], "albedoPred.exr"))


@pytest.mark.gpu
def test_real_image_run_lands_where_the_reference_run_did(tmp_path):
    """BASELINE configs[1] on the reference's own sample (examples/indoor2.png + the MaterialNet predictions it shipped,
    tests/golden/indoor2.npz): `--model_name pos_mlp --opt_order rm a --opt_env_from 2 --opt_src a`, spp 64.  The reference's
    Mitsuba-based run of this command ended at 27.65 dB (its final render vs the photograph); this build must end at least there,
    and at maps close to the reference's final maps (far closer than the initial guess is)."""
    import importlib.util

    _cuda()
    spec = importlib.util.spec_from_file_location("real_image", os.path.join(os.path.dirname(__file__), "..", "tools", "real_image.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(0)
    out = mod.run(mod.parse(["--out", str(tmp_path)]))
    assert out["psnr_vs_photo"]["reference_mitsuba_final_render_unscaled"] == pytest.approx(27.65, abs=0.05)
    assert out["psnr_vs_photo"]["this_build_final_render"] > 28.0
    assert out["psnr_build_render_vs_mitsuba_render"] > 26.5
    for k, bound in (("albedo", 0.08), ("roughness", 0.12), ("metallic", 0.12)):
        d = out["mean_abs_map_difference"][k]
        assert d["final_vs_reference_final"] < bound < d["initial_vs_reference_final"], (k, d)
    for name in ("best_results/albedo.exr", "best_results/envmap.hdr", "opt_env_img.png", "final_envmap.hdr", "config.json"):
        assert os.path.exists(os.path.join(str(tmp_path), "indoor2", name)), name


def test_hip_render_of_the_references_final_maps_against_mitsubas_render_of_them(golden_dir):
    """The image-level pin against Mitsuba, on the GPU (the CPU twin with the full-precision files is
    tests/test_reference_outputs.py).  The reference shipped, for its sample photograph, the maps its run ended at (best_results/{albedo,roughness,
    metallic}.exr), the 16 x 32 envmap, MaterialNet's depth and Mitsuba's render of exactly those (best_results/rendered_img.exr: `path`,
    max_depth 4, spp 64: inverse_img_w_mi.py:49-52).  The HIP render of the same maps under the same light and geometry is the depth-1,
    un-shadowed term of that estimator: what separates the two images is occlusion, inter-reflection and Monte-Carlo
    noise, not a convention.  The floors below are what that residual measures today (21-25 dB); a convention regression (camera, normals
    from depth, envmap -> SH, gamma) lands far below them.  Only `indoor` has such a pair: jinjya's rendered_img.exr is not the render of
    its maps (19.8 dB against its own panel, tests/test_reference_outputs.py)."""
    from materialist_amd import render, sh

    dev = _cuda()
    z = np.load(os.path.join(golden_dir, "indoor2.npz"))
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
    a = t(z["ref_albedo_u8"].astype(np.float32) / 255.0)
    r = t(z["ref_roughness_u8"].astype(np.float32)[..., None] / 255.0).clamp(0.07, 1.0)
    m = t(z["ref_metallic_u8"].astype(np.float32)[..., None] / 255.0)
    ref = z["ref_render_f16"].astype(np.float64)
    depth = z["depth_pred_f32"]
    depth = 2 * depth.max() - depth                                                  # inverse_img_w_mi.py:722
    scene = render.load_estimated_mesh(t(depth), use_mesh_normal=True)
    scene._set("emitter.data", t(z["ref_envmap_f32"]))                               # 16 x 32 texels -> SH25 by the fixed projection
    with torch.no_grad():
        img = render.render_w_brdf(scene, a, r, m, None, 64).cpu().numpy().astype(np.float64)
    g = lambda x: np.clip(x, 0, 1) ** (1 / 2.2)
    psnr = lambda x, y: -10 * np.log10(np.mean((g(x) - g(y)) ** 2))
    raw, matched = psnr(img, ref), psnr(img * (ref.mean() / img.mean()), ref)        # the BRDF loss is scale-free (:388-391)
    _report("HIP render vs Mitsuba render of the reference's final maps (indoor), dB raw", raw, 20.5, floor=True)
    _report("HIP render vs Mitsuba render of the reference's final maps (indoor), dB mean-matched", matched, 24.5, floor=True)
    assert raw > 20.5 and matched > 24.5, (raw, matched)
    # the two renders agree on WHERE the light is: the correlation of their luminance gradients along x and y
    lum = lambda x: g(x).mean(-1)
    gx = lambda x: np.diff(x, axis=1)[8:-8, 8:-8].reshape(-1)
    c = np.corrcoef(gx(lum(img)), gx(lum(ref)))[0, 1]
    assert c > 0.5, c


@pytest.mark.parametrize("M,hidden,skip,d0,n_out", [(4096, (256, 256, 256, 256), (1, 3), 15, 5), (1000, (64, 128, 64), (2,), 12, 3),
                                                   (2 * 128 + 7, (256, 256), (), 10, 8), (1024 + 40, (256, 128, 256), (), 12, 4)])
def test_posmlp_mfma_kernels_match_the_torch_composition(M, hidden, skip, d0, n_out):
    """f2: the hand-written f32-MFMA sine layers (forward with sin/cos epilogue, dL/d input with the cos and bias-gradient
    epilogue, slab-split weight gradient) against the torch/BLAS composition of the same network: outputs and every gradient.
    The four shapes reach every kernel: full-width (256 outputs), pipelined (128 outputs, K = 256), general (narrow layers, K = 15
    and K = 5) and the ragged last row tile.
    fp32 both sides; tolerance = accumulated rounding of K <= 256 dot products and M-row reductions."""
    from materialist_amd import posmlp

    dev = _cuda()
    dims = [d0] + list(hidden) + [n_out]
    wb = []
    for l in range(len(dims) - 1):
        n = dims[l + 1] - d0 if (l + 1) in skip else dims[l + 1]
        k = dims[l]
        wb += [(torch.rand(n, k, device=dev) * 2 - 1) / k ** 0.5, (torch.rand(n, device=dev) * 2 - 1) / k ** 0.5]
    x0 = torch.randn(M, d0, device=dev)
    x0[:, 0] = torch.arange(M, device=dev) % 512          # pixel coordinates: pre-activations of a few hundred radians
    posmlp._PosMlpHipFn.MIN_ROWS = 1
    assert posmlp._PosMlpHipFn.supported(x0, skip, wb[0::2])
    posmlp._PosMlpHipFn.MIN_ROWS = 8192
    go = torch.randn(M, n_out, device=dev)
    res = []
    for fn in ("hip", "torch"):
        ps = [t.clone().requires_grad_(True) for t in wb]
        out = posmlp._PosMlpHipFn.apply(x0, skip, *ps) if fn == "hip" else posmlp._PosMlpFn.apply(x0, skip, len(dims) - 2, *ps)
        out.backward(go)
        res.append((out.detach(), [p.grad for p in ps]))
    (o_h, g_h), (o_t, g_t) = res
    assert (o_h - o_t).abs().max().item() <= 2e-5 * max(1.0, o_t.abs().max().item())
    for i, (a, b) in enumerate(zip(g_h, g_t)):
        scale = b.abs().max().item() + 1e-6
        assert (a - b).abs().max().item() <= 3e-4 * scale, (i, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("N,K", [(256, 256), (241, 256), (256, 241)])
@pytest.mark.parametrize("nprod", [6, 9])
def test_split_operand_sine_layers_match_fp64_and_the_f32_kernels(N, K, nprod):
    """f2: the split-operand kernels (each f32 operand = three bf16 pieces, 6 or 9 bf16 MFMA products, f32 accumulate) are an
    f32-accurate product: against an fp64 product of the same rows their error equals that of the exact-f32 MFMA kernels (the
    dominating term is the f32 accumulation over K <= 256), for the three operand shapes of the 8-layer network (full, skip-layer
    outputs N = 241, skip-layer reduction K = 241 whose padding columns hold scratch), forward (sin/cos) and input gradient (cos
    factor and bias-gradient column sums)."""
    from materialist_amd import ops

    dev = _cuda()
    torch.manual_seed(3)
    M = 128 * 300                                         # more tiles than CUs: every workgroup streams more than one tile
    x = torch.randn(M, 256, device=dev)
    x[:, 0] *= 50.0                                       # pre-activations of tens of radians, as with pixel coordinates
    wp = torch.full((N, 256), 7.0, device=dev)            # columns >= K are scratch and must not be read as weights
    wp[:, :K] = (torch.rand(N, K, device=dev) * 2 - 1) / 16
    b = torch.randn(N, device=dev) * 0.1
    rows = slice(M - 4096, M)
    pre = x[rows, :K].double() @ wp[:, :K].double().t() + b.double()
    ws = ops.mlp_split_weights(wp, N, K)
    s0, c0 = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    s1, c1 = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    ops.mlp_layer_fwd(x, wp, b, s0, c0, K)
    ops.mlp_layer_fwd_bx(x, ws, b, s1, c1, N, K, nprod)
    torch.cuda.synchronize()
    e_f32 = (s0[rows, :N].double() - torch.sin(pre)).abs().max().item()
    e_bx = (s1[rows, :N].double() - torch.sin(pre)).abs().max().item()
    assert e_bx <= max(2.0 * e_f32, 3e-5), (e_bx, e_f32)
    assert (c1[rows, :N].double() - torch.cos(pre)).abs().max().item() <= 3e-5
    assert (s1[:, :N] - s0[:, :N]).abs().max().item() <= 6e-5

    g = torch.randn(M, 256, device=dev)
    cprev = torch.rand(M, 256, device=dev)
    ref = (g[rows, :K].double() @ wp[:, :K].double().t()) * cprev[rows, :N].double()
    gp0, gp1 = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    db0, db1 = torch.empty(N, device=dev), torch.empty(N, device=dev)
    ops.mlp_layer_bwd_input(g, wp, cprev, gp0, N, K, db0)
    ops.mlp_layer_bwd_input_bx(g, ws, cprev, gp1, N, K, db1, nprod)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert (gp1[rows, :N].double() - ref).abs().max().item() <= 2e-6 * scale
    assert (gp1[:, :N] - gp0[:, :N]).abs().max().item() <= 4e-6 * scale
    col = gp1[:, :N].double().sum(0)
    assert (db1.double() - col).abs().max().item() <= 1e-5 * (col.abs().max().item() + 1.0)

    # weight gradient g^T x over all rows; the columns beyond N / K hold non-finite scratch that must not leak
    xin = torch.randn(M, 256, device=dev)
    xin[:, K:] = float("nan")
    gg = g.clone()
    gg[:, N:] = float("inf")
    ref_w = torch.zeros(N, K, dtype=torch.float64, device=dev)
    for c in range(0, M, 12800):
        ref_w += gg[c:c + 12800, :N].double().t() @ xin[c:c + 12800, :K].double()
    dw0 = ops.mlp_layer_bwd_weight(gg, xin, N, K)
    dw1 = ops.mlp_layer_bwd_weight_bx(gg, xin, N, K, nprod)
    torch.cuda.synchronize()
    scale_w = ref_w.abs().max().item()
    e0 = (dw0.double() - ref_w).abs().max().item()
    e1 = (dw1.double() - ref_w).abs().max().item()
    assert torch.isfinite(dw1).all()
    assert e1 <= max(2.0 * e0, 2e-6 * scale_w), (e1, e0, scale_w)


@pytest.mark.parametrize("N,K", [(256, 256), (241, 256)])
def test_two_piece_f16_forward_layers_are_f32_accurate(N, K):
    """f2 (round 5): the forward sine layers on TWO f16 pieces per operand and three products (nprod 3; mymodels/mlps.py:102-103).  Against an
    fp64 product of the same rows the error is that of the exact-f32 MFMA kernel and of the three-bf16-piece form (the f32 accumulation
    over K = 256 dominates all three), for sine-like rows with a coordinate column of hundreds, for rows of small numbers (second pieces
    in f16's subnormal range: the matrix cores keep them) and for the skip-layer shape with its x0 tail; the packed sines' cosine sign
    is right wherever |cos| > 1e-5."""
    from materialist_amd import ops

    dev = _cuda()
    torch.manual_seed(17)
    M = 128 * 300
    w = torch.zeros(N, 256, device=dev)
    w[:, :K] = (torch.rand(N, K, device=dev) * 2 - 1) / 16
    b = torch.randn(N, device=dev) * 0.1
    rows = slice(M - 4096, M)
    for xscale, coord in ((1.0, 300.0), (1e-2, 0.0)):
        x = torch.sin(torch.randn(M, 256, device=dev) * 3) * xscale
        if coord:
            x[:, 255] = torch.randint(0, 512, (M,), device=dev).float()       # a pixel coordinate of the x0 tail
            w[:, 255] *= 0.05
        pre = x[rows, :K].double() @ w[:, :K].double().t() + b.double()
        s0, c0 = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
        ops.mlp_layer_fwd(x, w, b, s0, c0, K)
        outs = {}
        for nprod in (6, 3):
            ws = ops.mlp_split_weights(w, N, K, f16=(nprod == 3))
            tail = None
            s1 = torch.empty(M, 256, device=dev)
            if N < 256:
                tail = torch.zeros(M, 16, device=dev)
                tail[:, :256 - N] = torch.randn(M, 256 - N, device=dev)
            ops.mlp_layer_fwd_bx(x, ws, b, s1, None, N, K, nprod, tail=tail)
            if N < 256:
                assert torch.equal(s1[:, N:], tail[:, :256 - N])
            outs[nprod] = s1
        torch.cuda.synchronize()
        ref = torch.sin(pre)
        e = {k: (v[rows, :N].double() - ref).abs() for k, v in (("f32", s0), (6, outs[6]), (3, outs[3]))}
        mx = {k: v.max().item() for k, v in e.items()}
        rms = {k: v.pow(2).mean().sqrt().item() for k, v in e.items()}
        assert mx[3] <= max(1.5 * mx["f32"], 1.5e-7), (xscale, mx)
        assert rms[3] <= max(1.25 * rms["f32"], 2.5e-8), (xscale, rms)
        assert rms[3] <= max(1.25 * rms[6], 2.5e-8), (xscale, rms)
        cs = torch.cos(pre)
        bit = outs[3][rows, :N].view(torch.int32) & 1
        wrong = ((cs < 0) != (bit == 1)) & (cs.abs() > 1e-5)
        assert not wrong.any()
    # an F16X2 image is not an operand of nprod 6 and the other way round: only the documented pairs are exercised; limits are rejected
    with pytest.raises(Exception):
        ops.mlp_layer_fwd_bx(x, ws, b, s1, torch.empty_like(s1), 200, K, 3)            # N < 256 without a tail


@pytest.mark.parametrize("size", [(128, 128), (192, 192)])
def test_forward_chain_is_the_layer_by_layer_forward(size):
    """f2 (round 5): `matpbr_mlp_chain_fwd` -- the whole 'arm' network (15 -> 241 -> 256 -> 241 -> 256 -> 5, skip concatenations, tanh head,
    mymodels/mlps.py:211-236) in one launch with the activations in registers between the layers -- against the layer-by-layer launches of
    ArmMlpPhase (same arithmetic, another order of the k index) and against the reference module in fp64: every layer's sign-carrying sines
    (x0 tails included), tanh(x), the three maps.  128 tiles (one per workgroup) and 288 tiles (workgroups that stream a second tile)."""
    import copy

    from materialist_amd import loop, ops, posmlp, render, synthetic
    from materialist_amd.armhead import ArmMlpPhase

    dev = _cuda()
    H, W = size
    sc = synthetic.make_scene(4, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    gt = torch.rand(H, W, 3, device=dev)
    a0, r0, m0 = (_t(v, dev).clamp(0, 1) for v in (sc.init_albedo, sc.init_roughness, sc.init_metallic))
    start_arm = torch.cat([a0.reshape(-1, 3), r0.reshape(-1, 1), m0.reshape(-1, 1)], -1).contiguous()
    torch.manual_seed(5)
    net = posmlp.brdf_net("arm").to(dev)
    net.lin4.weight.data.normal_(0, 0.05)
    net.lin4.bias.data.normal_(0, 0.05)
    fixed = {"albedo": a0, "roughness": r0, "metallic": m0}
    outs = {}
    for chain in (False, True):
        ArmMlpPhase.FWD_CHAIN = chain
        try:
            ph = ArmMlpPhase(scene, gt, copy.deepcopy(net), start_arm, fixed, optimize_part="arm", spp=8)
            assert ph.chain == chain
            for l_, b_ in enumerate(ph.bufs):                            # (the layer-by-layer path wrote the x0 tails once, at construction)
                if chain:
                    b_.fill_(float("nan"))
                else:
                    b_[:, :ph.ns[l_]] = float("nan")
            maps = ph.forward()
            torch.cuda.synchronize()
            outs[chain] = ([b_.clone() for b_ in ph.bufs], ph.th.clone(), {k: v.clone() for k, v in maps.items()})
        finally:
            ArmMlpPhase.FWD_CHAIN = True
    (b0, th0, m0_), (b1, th1, m1_) = outs[False], outs[True]
    for l, (x, y) in enumerate(zip(b0, b1)):
        assert torch.isfinite(y).all(), l
        n = ph.ns[l]
        assert torch.equal(x[:, n:], y[:, n:]), l                                       # the x0 tail of a skip layer's buffer
        # the first layer's arguments are pixel coordinates times weights (|pre| of order 100: one f32 ulp is 8e-6, and the two paths add the
        # 15 terms in different orders); deeper layers amplify the rounding below them.  Measured: 6e-6 (192 x 192), 3.5e-6 (128 x 128)
        assert (x[:, :n] - y[:, :n]).abs().max().item() <= 2e-5, (l, (x[:, :n] - y[:, :n]).abs().max().item())
        sx, sy = x[:, :n].view(torch.int32) & 1, y[:, :n].view(torch.int32) & 1        # the sign of the cosine, away from cos = 0
        big = (1 - x[:, :n].double() ** 2).clamp_min(0).sqrt() > 1e-3
        assert torch.equal(sx[big], sy[big]), l
    assert (th0[:, :5] - th1[:, :5]).abs().max().item() <= 2e-5 and float(th1[:, 5:].abs().max()) == 0.0
    for k in m0_:
        assert (m0_[k] - m1_[k]).abs().max().item() <= 2.6e-5, k
    # the reference module in fp64 (mymodels/mlps.py:211-236 restated in posmlp.PosMLP)
    ref = copy.deepcopy(net).double()(start_arm.double())
    got = torch.cat([m1_["albedo"].reshape(-1, 3), ((m1_["roughness"].reshape(-1, 1) - 0.07) / 0.93), m1_["metallic"].reshape(-1, 1)], -1)
    assert (got.double() - ref).abs().max().item() <= 4e-5


def test_deferred_folds_of_the_backward_pass_are_the_same_bits():
    """`matpbr_mlp_reduce_jobs`: the folds of the backward pass's partial sums (three weight gradients' 256 slabs, the column sums behind four bias
    gradients, the output layer's and the first layer's skinny products) in ONE launch before the optimiser step, against one launch behind every
    product (ArmMlpPhase.DEFER_REDUCE False): the same gradient buffer and the same weights after three AdamW steps, bit for bit."""
    import copy

    from materialist_amd import posmlp, render, synthetic
    from materialist_amd.armhead import ArmMlpPhase

    dev = _cuda()
    H = W = 128
    sc = synthetic.make_scene(6, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    gt = torch.rand(H, W, 3, device=dev)
    a0, r0, m0 = (_t(v, dev).clamp(0, 1) for v in (sc.init_albedo, sc.init_roughness, sc.init_metallic))
    start_arm = torch.cat([a0.reshape(-1, 3), r0.reshape(-1, 1), m0.reshape(-1, 1)], -1).contiguous()
    torch.manual_seed(9)
    net = posmlp.brdf_net("arm").to(dev)
    net.lin4.weight.data.normal_(0, 0.05)
    fixed = {"albedo": a0, "roughness": r0, "metallic": m0}
    runs = {}
    for defer in (False, True):
        ArmMlpPhase.DEFER_REDUCE = defer
        try:
            ph = ArmMlpPhase(scene, gt, copy.deepcopy(net), start_arm, fixed, optimize_part="arm", spp=8)
            assert ph.bwd_f16
            for _ in range(3):
                ph.step()
            torch.cuda.synchronize()
            runs[defer] = (ph.gflat.clone(), ph.flat.clone(), ph.stats.clone())
            from materialist_amd import ops as _ops

            assert float(ph.hyper[1]) == 3.0 == float(ph.stats[0, _ops.STAT_ITERS])      # (the optimiser's step count IS the statistics row's)
        finally:
            ArmMlpPhase.DEFER_REDUCE = True
    assert float(runs[True][0].abs().max()) > 0.0
    for a, b in zip(runs[True], runs[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n_prev,n_red", [(256, 256), (241, 256), (256, 241)])
def test_block_scaled_f16_backward_products(n_prev, n_red):
    """f2 (round 5): the backward products of the 256-wide layers on two f16 pieces with one exponent per 128-row tile
    (`matpbr_mlp_layer_bwd_input_blk`, `matpbr_mlp_layer_bwd_weight_blk`, `matpbr_mlp_out_layer_bwd_tmax`; the autograd backward of
    mymodels/mlps.py:102-103).  Gradients of the size a mean loss over 512 x 512 pixels produces (1e-6), tiles whose magnitudes differ by
    1e4 (each tile has its own exponent), a few elements far below their tile's largest: against fp64 the error is that of the three-bf16-piece
    form and within 1.5 x the exact-f32 kernels' (the f32 accumulation dominates), ROW by row inside a tile as well; the tile maxima the
    kernels write are the maxima; the bias gradient is the column sum; a zero tile stays zero."""
    from materialist_amd import ops

    dev = _cuda()
    torch.manual_seed(23)
    M = 128 * 520                                         # more tiles than workgroups
    T = M // 128
    tile_mag = torch.exp(torch.randn(T, device=dev) * 2.0).clamp(1e-2, 1e2).repeat_interleave(128)[:, None]
    g = torch.randn(M, 256, device=dev) * 1e-6 * tile_mag
    g[:, ::7] *= 1e-3                                     # elements far below their tile's largest
    g[128 * 5:128 * 6] = 0.0                              # a tile without any gradient
    g[:, n_red:] = float("nan")                           # scratch columns of a ragged reduction
    w = (torch.rand(n_red, 256, device=dev) * 2 - 1) / 16          # forward weight W[n_red, n_prev]: the operand is its transpose
    x_prev = torch.randn(M, 256, device=dev)
    wb = torch.randn(n_prev, 256, device=dev) / 16
    s_prev = torch.empty(M, 256, device=dev)
    ops.mlp_layer_fwd_bx(x_prev, ops.mlp_split_weights(wb, n_prev, 256, f16=True), torch.zeros(n_prev, device=dev), s_prev, None, n_prev, 256, 3,
                         tail=torch.zeros(M, 16, device=dev) if n_prev < 256 else None)
    tmax = g[:, :n_red].abs().view(T, 128, n_red).amax((1, 2)).contiguous().view(torch.int32)
    cosv = torch.cos(x_prev.double() @ wb[:, :256].double().t())
    ref = (g[:, :n_red].double() @ w[:, :n_prev].double()) * cosv
    outs = {}
    for mode in ("bf16", "f16"):
        gp, db = torch.full((M, 256), float("nan"), device=dev), torch.empty(n_prev, device=dev)
        if mode == "bf16":
            ops.mlp_layer_bwd_input_bx(g, ops.mlp_split_weights(w, n_prev, n_red, transposed=True), s_prev, gp, n_prev, n_red, db, 6, packed=True)
            tm_out = None
        else:
            tm_out = ops.mlp_tile_max(M, dev)
            ops.mlp_layer_bwd_input_blk(g, tmax, ops.mlp_split_weights(w, n_prev, n_red, transposed=True, f16=True), s_prev, gp, n_prev, n_red, db, tm_out)
        outs[mode] = (gp, db, tm_out)
    torch.cuda.synchronize()
    # the rebuilt cosine (|error| ~ 2e-7 / |cos|) is common to both forms: compare on the product before the cosine where it matters
    row_scale = ref.abs().amax(1, keepdim=True).view(T, 128, 1).amax(1, keepdim=True).expand(T, 128, 1).reshape(M, 1) + 1e-30   # per TILE
    e = {k: (v[0][:, :n_prev].double() - ref).abs() / row_scale for k, v in outs.items()}
    assert torch.isfinite(outs["f16"][0][:, :n_prev]).all()
    assert e["f16"].max().item() <= max(1.5 * e["bf16"].max().item(), 2e-5), (e["f16"].max().item(), e["bf16"].max().item())
    rms = {k: v.pow(2).mean().sqrt().item() for k, v in e.items()}
    assert rms["f16"] <= 1.25 * rms["bf16"] + 1e-9, rms
    assert (outs["f16"][0][:, :n_prev] - outs["bf16"][0][:, :n_prev]).abs().div(row_scale).max().item() <= 4e-6
    assert outs["f16"][0][128 * 5:128 * 6, :n_prev].abs().max().item() == 0.0
    gp16, db16, tm_out = outs["f16"]
    want = gp16[:, :n_prev].abs().view(T, 128, n_prev).amax((1, 2))
    assert torch.equal(tm_out.view(torch.float32), want)
    col = gp16[:, :n_prev].double().sum(0)
    assert (db16.double() - col).abs().max().item() <= 1e-5 * (col.abs().max().item() + 1e-30)
    # weight gradient g^T x: x the sign-carrying sines (and a coordinate column), g as above
    xin = s_prev.clone()
    xin[:, 255] = torch.randint(0, 512, (M,), device=dev).float()
    gg = g.clone()
    gg[:, n_red:] = float("inf")
    ref_w = torch.zeros(n_red, 256, dtype=torch.float64, device=dev)
    for c in range(0, M, 16640):
        ref_w += gg[c:c + 16640, :n_red].double().t() @ xin[c:c + 16640].double()
    dw6 = ops.mlp_layer_bwd_weight_bx(gg, xin, n_red, 256, 6)
    dw3 = ops.mlp_layer_bwd_weight_blk(gg, tmax, xin, n_red, 256)
    dw0 = ops.mlp_layer_bwd_weight(g.nan_to_num(0.0), xin, n_red, 256)
    torch.cuda.synchronize()
    assert torch.isfinite(dw3).all()
    sw = ref_w.abs().max().item()
    e0, e6, e3 = ((d.double() - ref_w).abs().max().item() / sw for d in (dw0, dw6, dw3))
    assert e3 <= max(1.5 * e0, 1.5 * e6, 2e-6), (e3, e6, e0)


@pytest.mark.parametrize("M", [16, 16 * 3, 16 * 1001, 16 * 4099])
def test_split_operand_weight_gradient_ragged_slabs(M):
    """The slab partition of the split-operand weight gradient for row counts that leave a short (odd number of 16-row steps)
    last slab, a single step, and fewer steps than the prefetch depth."""
    from materialist_amd import ops

    dev = _cuda()
    torch.manual_seed(M)
    g = torch.randn(M, 256, device=dev)
    x = torch.randn(M, 256, device=dev)
    ref = g.double().t() @ x.double()
    dw = ops.mlp_layer_bwd_weight_bx(g, x, 256, 256, 6)
    torch.cuda.synchronize()
    assert (dw.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() * max(1.0, (M / 4096) ** 0.5)


@pytest.mark.parametrize("M,K", [(128, 256), (128 * 5, 256), (128 * 700, 256), (128 * 300, 241)])
def test_split_operand_layer_loops_agree_bit_for_bit(M, K):
    """f2: the three main loops of the 256-wide split-operand layers -- register-staged (0), LDS-DMA with one 512-thread workgroup per CU
    (1), LDS-DMA with two 256-thread workgroups per CU and 64 x 128 wave tiles (2, the default) -- form the same products in the same
    order: sines, cosines, packed sines and input gradients are the same bits; the bias gradient (a different grouping of the column
    sums) agrees to rounding.  One tile, fewer tiles than workgroups, more tiles than workgroups; K = 241: the reduction of the layer
    after a skip layer's gradient, whose operand columns at and beyond K hold scratch (here NaN) that no loop may read as data."""
    from materialist_amd import _lib, ops

    dev = _cuda()
    lib = _lib.load()
    torch.manual_seed(11)
    x = torch.randn(M, 256, device=dev)
    x[:, 0] *= 30.0
    w = torch.randn(256, 256, device=dev) / 16
    b = torch.randn(256, device=dev)
    g = torch.randn(M, 256, device=dev)
    x[:, K:] = float("nan")
    g[:, K:] = float("nan")
    ws = ops.mlp_split_weights(w, 256, K)
    outs = []
    was = lib.matpbr_mlp_set_lds_dma(0)
    try:
        for mode in (0, 1, 2):
            lib.matpbr_mlp_set_lds_dma(mode)
            s_, c_, sp = (torch.zeros(M, 256, device=dev) for _ in range(3))
            gp, gq = torch.zeros(M, 256, device=dev), torch.zeros(M, 256, device=dev)
            db, dq = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
            for _ in range(2):                                   # twice: the persistent loops leave nothing behind
                ops.mlp_layer_fwd_bx(x, ws, b, s_, c_, 256, K, 6)
                ops.mlp_layer_fwd_bx(x, ws, b, sp, None, 256, K, 6)
                ops.mlp_layer_bwd_input_bx(g, ws, c_, gp, 256, K, db, 6)
                ops.mlp_layer_bwd_input_bx(g, ws, sp, gq, 256, K, dq, 6, packed=True)
            torch.cuda.synchronize()
            outs.append((s_, c_, sp, gp, gq, db, dq))
    finally:
        lib.matpbr_mlp_set_lds_dma(was)
    assert all(torch.isfinite(t).all() for t in outs[0])
    for mode in (1, 2):
        for k, name in enumerate(("s", "c", "packed s", "g'", "g' (packed)")):
            assert torch.equal(outs[0][k], outs[mode][k]), (mode, name)
        for k in (5, 6):
            scale = outs[0][k].abs().max().item() + 1.0
            assert (outs[0][k] - outs[mode][k]).abs().max().item() <= 2e-6 * scale * max(1.0, (M / 4096) ** 0.5), (mode, k)


def test_split_operand_layers_full_size_properties():
    """BASELINE configs[1] size (512 x 512 = 262144 rows), through size-independent properties: sin^2 + cos^2 = 1 at every output,
    the x0 tail of a skip layer's buffer survives the forward untouched, the input gradient and the weight gradient are linear in g
    (within f32 rounding), the bias gradient is the column sum of the input gradient, and a repeated launch gives the same bits."""
    from materialist_amd import ops

    dev = _cuda()
    torch.manual_seed(9)
    M, N, K = 512 * 512, 241, 256
    x = torch.randn(M, 256, device=dev)
    x[:, 0] = torch.arange(M, device=dev) % 512                       # pixel coordinates: arguments of hundreds of radians
    w = torch.zeros(N, 256, device=dev)
    w[:, :K] = (torch.rand(N, K, device=dev) * 2 - 1) / 16
    w[:, 0] *= 0.05
    b = torch.randn(N, device=dev) * 0.1
    tail = torch.randn(M, 256 - N, device=dev)
    s_out, c_out = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    s_out[:, N:] = tail
    ws = ops.mlp_split_weights(w, N, K)
    ops.mlp_layer_fwd_bx(x, ws, b, s_out, c_out, N, K, 6)
    assert torch.equal(s_out[:, N:], tail)
    one = s_out[:, :N] ** 2 + c_out[:, :N] ** 2
    assert (one - 1).abs().max().item() < 5e-6
    s2, c2 = torch.empty_like(s_out), torch.empty_like(c_out)
    ops.mlp_layer_fwd_bx(x, ws, b, s2, c2, N, K, 6)
    assert torch.equal(s2[:, :N], s_out[:, :N]) and torch.equal(c2[:, :N], c_out[:, :N])
    # the tail handed to the kernel instead (whole 16-byte stores): same outputs, the tail columns hold the given values
    x0p = torch.zeros(M, 16, device=dev)
    x0p[:, :256 - N] = tail
    s3, c3 = torch.full((M, 256), float("nan"), device=dev), torch.empty(M, 256, device=dev)
    ops.mlp_layer_fwd_bx(x, ws, b, s3, c3, N, K, 6, tail=x0p)
    assert torch.equal(s3[:, :N], s_out[:, :N]) and torch.equal(s3[:, N:], tail) and torch.equal(c3[:, :N], c_out[:, :N])
    # the first layer's kernel (K = 15, exact-f32 MFMA) with and without the tail
    xk = torch.zeros(M, 16, device=dev)
    xk[:, :15] = torch.randn(M, 15, device=dev)
    wk = torch.zeros(N, 16, device=dev)
    wk[:, :15] = torch.randn(N, 15, device=dev) / 4
    sa, ca = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    sb, cb = torch.full((M, 256), float("nan"), device=dev), torch.empty(M, 256, device=dev)
    sa[:, N:] = tail
    ops.mlp_layer_fwd(xk, wk, b, sa, ca, 15)
    ops.mlp_layer_fwd(xk, wk, b, sb, cb, 15, tail=x0p)
    assert torch.equal(sa, sb) and torch.equal(ca[:, :N], cb[:, :N])
    ref = torch.sin(xk[:4096, :15].double() @ wk[:, :15].double().t() + b.double())
    assert (sa[:4096, :N].double() - ref).abs().max().item() < 5e-6
    # backward operands: g [M, 256] with n_red = 256 columns, n_prev = N
    g1, g2 = torch.randn(M, 256, device=dev), torch.randn(M, 256, device=dev)
    wt = (torch.rand(N, 256, device=dev) * 2 - 1) / 16
    wts = ops.mlp_split_weights(wt, N, 256)
    outs, dbs, dws = [], [], []
    for g in (g1, g2, g1 + g2):
        gp, db = torch.empty(M, 256, device=dev), torch.empty(N, device=dev)
        ops.mlp_layer_bwd_input_bx(g, wts, c_out, gp, N, 256, db, 6)
        outs.append(gp[:, :N].clone()); dbs.append(db)
        dws.append(ops.mlp_layer_bwd_weight_bx(g, x, 256, 256, 6))
    scale = outs[2].abs().max().item()
    assert (outs[0] + outs[1] - outs[2]).abs().max().item() < 4e-6 * scale
    col = outs[2].double().sum(0)
    assert (dbs[2].double() - col).abs().max().item() < 1e-5 * (col.abs().max().item() + 1.0)
    assert (dws[0] + dws[1] - dws[2]).abs().max().item() < 2e-5 * dws[2].abs().max().item()
    assert torch.equal(ops.mlp_layer_bwd_weight_bx(g1, x, 256, 256, 6), dws[0])


@pytest.mark.parametrize("products", [0, 6, 9])
def test_posmlp_autograd_function_at_image_size_for_every_product_mode(products):
    """f2: the whole 8-layer network through `_PosMlpHipFn` at 128 x 128 pixels (above MIN_ROWS, so the split-operand kernels are
    the ones dispatched when PRODUCTS != 0) against the torch/BLAS composition: outputs and every parameter gradient."""
    from materialist_amd import posmlp

    dev = _cuda()
    torch.manual_seed(5)
    M, d0, n_out, skip = 128 * 128, 15, 5, (4,)
    dims = [d0] + [256] * 8 + [n_out]
    wb = []
    for l in range(len(dims) - 1):
        n = dims[l + 1] - d0 if (l + 1) in skip else dims[l + 1]
        k = dims[l]
        wb += [(torch.rand(n, k, device=dev) * 2 - 1) / k ** 0.5, (torch.rand(n, device=dev) * 2 - 1) / k ** 0.5]
    x0 = torch.randn(M, d0, device=dev)
    go = torch.randn(M, n_out, device=dev)
    assert posmlp._PosMlpHipFn.supported(x0, skip, wb[0::2])
    saved = posmlp._PosMlpHipFn.PRODUCTS
    posmlp._PosMlpHipFn.PRODUCTS = products
    try:
        res = []
        for fn in ("hip", "torch"):
            ps = [t.clone().requires_grad_(True) for t in wb]
            out = posmlp._PosMlpHipFn.apply(x0, skip, *ps) if fn == "hip" else posmlp._PosMlpFn.apply(x0, skip, len(dims) - 2, *ps)
            out.backward(go)
            res.append((out.detach(), [p.grad for p in ps]))
    finally:
        posmlp._PosMlpHipFn.PRODUCTS = saved
    (o_h, g_h), (o_t, g_t) = res
    assert (o_h - o_t).abs().max().item() <= 3e-5 * max(1.0, o_t.abs().max().item())
    for i, (a, b) in enumerate(zip(g_h, g_t)):
        scale = b.abs().max().item() + 1e-6
        assert (a - b).abs().max().item() <= 5e-4 * scale, (i, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("tag,kw", [("arm", dict(color_ch=5, out_dims=5, multires_view=2, output_type="arm")),
                                    ("armn", dict(color_ch=8, out_dims=8, multires_view=0, output_type="armn")),
                                    ("env", dict(color_ch=3, out_dims=3, multires_view=2, output_type="envmap"))])
def test_posmlp_mfma_path_matches_the_reference_module(golden_dir, tag, kw):
    """f2 pinned to the reference itself on the GPU: the networks of tests/golden/posmlp.npz (weights, inputs, outputs and gradients
    recorded from mymodels/mlps.py in fp64) through the hand-written MFMA sine layers in fp32."""
    from materialist_amd import posmlp

    dev = _cuda()
    g = np.load(os.path.join(golden_dir, "posmlp.npz"))
    net = posmlp.PosMLP(hidden=(64, 64, 64, 64), skip=(1, 3), **kw)
    net.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith(f"{tag}.sd.")})
    net = net.to(dev)
    x = torch.from_numpy(g[f"{tag}.in"]).float().to(dev)
    posmlp._PosMlpHipFn.MIN_ROWS = 1
    try:
        wb = [t for l in range(net.n_layers) for t in ((getattr(net, f"lin{l}").linear if l < net.n_layers - 1 else getattr(net, f"lin{l}")).weight,)]
        assert posmlp._PosMlpHipFn.supported(net._points(x), net.skip, wb)
        out = net(x)
        (out * torch.from_numpy(g[f"{tag}.w"]).float().to(dev)).sum().backward()
    finally:
        posmlp._PosMlpHipFn.MIN_ROWS = 8192
    ref = g[f"{tag}.out"]
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    for got, key in ((net.lin0.linear.weight.grad, "d_lin0_w"), (net.lin4.bias.grad, "d_lin4_b")):
        r = g[f"{tag}.{key}"]
        assert np.abs(got.cpu().numpy() - r).max() <= 2e-4 * (np.abs(r).max() + 1e-9), key


def test_outdoor_sample_with_mesh_mask(tmp_path):
    """The reference's outdoor sample (output_imgs/jinjya at 256x256, tests/golden/jinjya256.npz): 29 % of the pixels are sky
    (`mesh_mask.png`) and must be explained by the environment light alone; `--model_name none --opt_order rm a --opt_env_from 2`."""
    import importlib.util

    _cuda()
    spec = importlib.util.spec_from_file_location("real_image", os.path.join(os.path.dirname(__file__), "..", "tools", "real_image.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(0)
    args = mod.parse(["--sample", "jinjya", "--model_name", "none", "--num_epochs", "1500", "--out", str(tmp_path)])
    out = mod.run_jinjya(args)
    assert any("see the environment directly" in ln for ln in out["log"])
    assert out["psnr_vs_photo"]["this_build_ground_only"] > 24.0      # 27 dB at this epoch cap, 42 dB with the default 5000
    assert out["psnr_vs_photo"]["this_build_sky_only"] > 28.0         # a 25-coefficient light has to carry the whole sky
    assert os.path.exists(os.path.join(str(tmp_path), "jinjya", "opt_env_img.png"))


def test_env_mlp_phase_matches_the_autograd_composition():
    """Hot loop A with the reference's envmap MLP (inverse_img_w_mi.py:117-124,238-254), every launch on the C ABI (EnvMlpPhase:
    small-tile MFMA layers, softplus + SH projection, the pass over the radiance transfer, explicit backward chain, one Adam
    launch with device-side step count) against the same iterations composed with torch autograd around `render_envmap`
    (PosMLP module -> softplus -> projection -> autograd render -> env_loss -> torch.optim.Adam)."""
    import copy

    from materialist_amd import loop, ops, posmlp, render, synthetic
    from materialist_amd.envhead import EnvMlpPhase

    dev = _cuda()
    H = W = 48
    spp = 16
    sc = synthetic.make_scene(8, H, W)

    def make_scene():
        s = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
        p = render.traverse(s)
        p["shape.bsdf.a"], p["shape.bsdf.r"], p["shape.bsdf.m"] = _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev)
        return s

    with torch.no_grad():
        gt = render.render_envmap(make_scene(), _t(sc.light, dev), spp).clone()
    torch.manual_seed(3)
    net_a = posmlp.envmap_net().to(dev)
    net_a.lin4.weight.data.normal_(0, 0.05)            # the reference zero-initialises the last layer: give the gradients something to do
    net_b = copy.deepcopy(net_a)
    ones = torch.ones(512, 3, device=dev)
    opt_a = torch.optim.Adam(net_a.parameters(), lr=1e-3)
    sa = make_scene()
    ph = EnvMlpPhase(make_scene(), gt, net_b, ones, spp=spp, lr=1e-3, use_graph=True)
    for it in range(7):                                  # 3 eager iterations, the capture, 3 replays
        if it == 5:
            for gp in opt_a.param_groups:
                gp["lr"] = 5e-4
            ph.set_lr(5e-4)
        env = net_a(ones).reshape(16, 32, 3)
        pred = render.render_envmap(sa, env, spp)
        total, mse, _ = loop._loss.env_loss(pred, gt)
        total.backward()
        opt_a.step()
        opt_a.zero_grad()
        ph.step()
        st = ph.stats[0].cpu().numpy()
        assert st[ops.STAT_MSE] == pytest.approx(float(mse.detach()), rel=5e-4), it
        assert st[ops.STAT_LOSS] == pytest.approx(float(total.detach()), rel=5e-4), it
    assert ph._graph is not None and ph.poll()["iters"].tolist() == [7]
    for (ka, va), (kb, vb) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
        assert ka == kb and (va - vb).abs().max().item() < 2e-5, ka       # 7 Adam steps of <= 1e-3
    assert_close(ph.head(), env.detach().cpu().numpy(), rtol=1e-3, what="envmap of the last iteration")
    assert ph.best_env.shape == (16, 32, 3) and ph.best_img.shape == (H, W, 3)
    # the module keeps working as a module (parameters are views of the flat buffer): state_dict round trip
    before = net_b(ones).detach().clone()
    sd = {k: v.clone() for k, v in net_b.state_dict().items()}
    net_b.load_state_dict(sd)
    assert torch.equal(net_b(ones).detach(), before)
    # the one-workgroup tail behind the pass over the transfer (matpbr_env_mlp_phase_step) against the three launches it replaces: the same bits,
    # with EarlyStopping armed and firing
    runs = {}
    for fused in (True, False):
        EnvMlpPhase.FUSED_TAIL = fused
        try:
            torch.manual_seed(2)
            net_c = copy.deepcopy(net_a)
            pc = EnvMlpPhase(make_scene(), gt, net_c, ones, spp=spp, lr=1e-2, patience=3, min_delta=0.3, use_graph=True)
        finally:
            EnvMlpPhase.FUSED_TAIL = True
        assert len(pc._calls) == (13 if fused else 15), len(pc._calls)
        for _ in range(4):
            pc.step()
        pc.step_many(6)
        pc.step_many(6)
        runs[fused] = (pc.flat.clone(), pc.adam_m.clone(), pc.stats.clone(), pc.history().clone(), pc.best_env.clone(), pc.g_out.clone(), pc.poll())
    for a, b in zip(runs[True][:6], runs[False][:6]):
        assert torch.equal(a, b)
    assert runs[True][6]["stopped"].tolist() == [True] and runs[True][6]["iters"].tolist() == runs[False][6]["iters"].tolist()
    assert runs[True][6]["iters"].tolist()[0] < 16


def test_sines_that_carry_the_sign_of_their_cosine():
    """One float per sine activation (include/matpbr.h `matpbr_mlp_layer_fwd_sgn` / `_bx` with c_out NULL): the stored sine is the exact
    kernel's sine up to its last mantissa bit, which holds the sign of the cosine; the backward epilogue rebuilds cos = +-sqrt(1 - sin^2):
    never the wrong sign, |error| <= 3e-7 / |cos| + 3e-6, and an input gradient whose rms error against fp64 is 1e-5 of its scale.
    Both kernels that write such sines (thin first layer, 256-wide split-operand layer) and both that read them."""
    from materialist_amd import ops

    dev = _cuda()
    torch.manual_seed(11)
    M = 128 * 200
    # 256-wide layer
    x = torch.randn(M, 256, device=dev)
    x[:, 0] *= 50.0
    w = (torch.rand(256, 256, device=dev) * 2 - 1) / 16
    b = torch.randn(256, device=dev) * 0.1
    ws = ops.mlp_split_weights(w, 256, 256)
    s_ref, c_ref = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    s_pk = torch.empty(M, 256, device=dev)
    ops.mlp_layer_fwd_bx(x, ws, b, s_ref, c_ref, 256, 256, 6)
    ops.mlp_layer_fwd_bx(x, ws, b, s_pk, None, 256, 256, 6)
    bits_pk = s_pk.view(torch.int32)
    # round 5: the packed form has its own one-polynomial sine (sin_packed: reduction by multiples of pi, |error| <= 1.2e-7 + the last bit)
    pre64 = x.double() @ w.double().t() + b.double()
    assert (s_pk.double() - torch.sin(pre64)).abs().max().item() <= max(1.2 * (s_ref.double() - torch.sin(pre64)).abs().max().item(), 3e-5)
    assert (s_pk - s_ref).abs().max().item() <= 3e-7                                   # the exact kernel's sine to two ulp ...
    assert torch.equal((bits_pk & 1).bool() | (c_ref.abs() < 1e-5), (c_ref < 0) | (c_ref.abs() < 1e-5))   # ... its last bit the sign of the cosine
    g = torch.randn(M, 256, device=dev)
    wt = (torch.rand(256, 256, device=dev) * 2 - 1) / 16
    wts = ops.mlp_split_weights(wt, 256, 256)
    gp_c, gp_s = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    db_c, db_s = torch.empty(256, device=dev), torch.empty(256, device=dev)
    ops.mlp_layer_bwd_input_bx(g, wts, c_ref, gp_c, 256, 256, db_c, 6)
    ops.mlp_layer_bwd_input_bx(g, wts, s_pk, gp_s, 256, 256, db_s, 6, packed=True)
    prod = g.double() @ wt.double().t()
    pre = x.double() @ w.double().t() + b.double()
    ref = prod * torch.cos(pre)
    scale = ref.abs().max().item()
    # the rebuilt cosine against the exact kernel's, seen through the two gradients: |error| <= 1.5e-7 / |cos|, at most 3e-4 (measured:
    # tools/dbg/sgn.py), never the opposite sign
    ct = c_ref.double()
    # (round 5: the packed sine is sin_packed's, 1.8e-7 from the true sine with its last bit: 4e-7 / |cos|, at most sqrt(2 x 1.8e-7) = 6e-4)
    bound = prod.abs() * torch.minimum(4e-7 / ct.abs().clamp_min(1e-9), torch.full_like(ct, 7e-4)) + 2e-6 * scale
    assert ((gp_s.double() - gp_c.double()).abs() <= bound).all()
    big = prod.abs() > 1e-2 * prod.abs().max()
    assert ((gp_s.double() * gp_c.double())[big & (ct.abs() > 1e-3)] > 0).all()
    rms = lambda t: float(t.pow(2).mean().sqrt())
    assert rms(gp_s.double() - ref) <= 2e-5 * rms(ref) and rms(gp_c.double() - ref) <= 2e-6 * rms(ref)
    assert (db_s.double() - gp_s.double().sum(0)).abs().max().item() <= 1e-5 * (gp_s.double().sum(0).abs().max().item() + 1.0)
    assert (gp_s - gp_c).abs().max().item() <= 2e-3 * scale                           # the worst unit: cos within 1e-3 of zero
    # thin first layer (K = 15) and the thin input gradient (5 columns) at image size
    x0 = torch.zeros(M, 16, device=dev)
    x0[:, :15] = torch.randn(M, 15, device=dev) * 3
    w0 = torch.zeros(241, 16, device=dev)
    w0[:, :15] = torch.randn(241, 15, device=dev) * 0.3
    b0 = torch.randn(241, device=dev) * 0.1
    t_ref, tc_ref, t_pk = torch.zeros(M, 256, device=dev), torch.zeros(M, 256, device=dev), torch.zeros(M, 256, device=dev)
    ops.mlp_layer_fwd(x0, w0, b0, t_ref, tc_ref, 15)
    ops.mlp_layer_fwd(x0, w0, b0, t_pk, None, 15, packed=True)
    assert (t_pk[:, :241] - t_ref[:, :241]).abs().max().item() <= 3e-7
    tc = tc_ref[:, :241]
    assert torch.equal((t_pk.view(torch.int32)[:, :241] & 1).bool() | (tc.abs() < 1e-5), (tc < 0) | (tc.abs() < 1e-5))
    assert float(t_pk[:, 241:].abs().max()) == 0.0                                    # the skip layer's tail columns are not touched
    d5 = torch.zeros(M, 8, device=dev)
    d5[:, :5] = torch.randn(M, 5, device=dev)
    wo_t = torch.zeros(256, 8, device=dev)
    wo_t[:, :5] = torch.randn(256, 5, device=dev) * 0.1
    o_c, o_s = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    bb_c, bb_s = torch.empty(256, device=dev), torch.empty(256, device=dev)
    ops.mlp_layer_bwd_input(d5, wo_t, c_ref, o_c, 256, 5, bb_c)
    ops.mlp_layer_bwd_input(d5, wo_t, s_pk, o_s, 256, 5, bb_s, packed=True)
    assert rms((o_s - o_c).double()) <= 2e-5 * rms(o_c.double())
