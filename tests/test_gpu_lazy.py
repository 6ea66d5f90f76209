"""GPU tests of the lazy re-sampling path of hot loop B (include/matpbr.h `matpbr_shade_fwd_lazy`, csrc/matpbr_lazy.hpp).

The gate: at EVERY iteration of a 2 000-iteration 512 x 512 'rm' run,
    |lazy - exact| <= 1e-3 max(|exact|, mean|exact|)   on every pixel,
where `exact` walks the 20 GGX samples of every pixel (matpbr_shade_fwd_ex) at the same parameters; d out / d r within 2e-3 of the
exact (detached) derivative at the pixels that were just re-sampled.  The models are checked against the oracle's specification
(oracle/matpbr_oracle.c `lazy_refresh_pixel`), the fused phase against the phase that walks every sample."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    torch.manual_seed(20250629)
    return torch.device("cuda:0")


def _t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)


def _rel(lazy, exact):
    """max over pixels of |lazy - exact| / max(|exact|, mean|exact|) (device scalar)"""
    scale = torch.maximum(exact.abs(), exact.abs().mean())
    return ((lazy - exact).abs() / scale).max()


def test_lazy_models_match_the_oracle_specification(oracle64):
    """A forced call builds the model of every pixel; a second call after a random move of the roughness re-samples exactly the pixels
    the specification says have left their interval (up to fp32 / fp16 rounding at the interval ends) and renders the others from
    their model.  Values, slopes, derivatives, intervals and renders against oracle.lazy_fwd."""
    from materialist_amd import ops, synthetic

    dev = _cuda()
    H, W, spp = 48, 64, 64
    sc = synthetic.make_scene(2, H, W)
    o = oracle64
    n64 = o.normals_from_depth(sc.depth.astype(np.float64))
    rng = np.random.default_rng(5)
    n64 = n64 + 0.25 * rng.normal(size=n64.shape)           # tilt: grazing views, samples near the horizon
    n64 /= np.linalg.norm(n64, axis=-1, keepdims=True)
    a, r, m, n, light = (_t(x, dev) for x in (sc.albedo, sc.roughness, sc.metallic, n64, sc.light))
    a64, r64, m64 = (x.cpu().numpy().astype(np.float64) for x in (a, r, m))
    n64 = n.cpu().numpy().astype(np.float64)
    l64 = light.cpu().numpy().astype(np.float64)
    dcache = ops.diffuse_cache(n, light, spp)
    exact = ops.shade_fwd(a, r, m, n, light, spp, dcache=dcache)
    floor = 0.5 * float(exact.mean())
    state = ops.lazy_state(a)
    out, j16 = ops.shade_fwd_lazy(a, r, m, n, light, spp, dcache, state, force=True, floor=floor)
    st, ref = ops.lazy_state_unpack(state, a)
    assert int(ref.sum()) == H * W
    assert float(_rel(out, exact)) < 2e-5                    # a forced call IS the exact render (fp32 summation order aside)
    st_o = np.zeros((H, W, o.lazy_nstate()))
    out_o, jac_o, ref_o = o.lazy_fwd(a64, r64, m64, n64, l64, st_o, spp, floor, force=True)
    st = st[0].cpu().numpy().astype(np.float64)
    S = np.abs(st_o[..., 4:10]).mean()
    assert np.abs(st[..., 0] - st_o[..., 0]).max() < 1e-6
    assert np.abs(st[..., 4:10] - st_o[..., 4:10]).max() < 2e-5 * S, "SD / S1 at the reference point"
    # slopes: one-sided differences over h = 1e-3 in fp32 (noise ~ 1e-6 S / h) stored as fp16
    tol_g = 3e-3 * (np.abs(st_o[..., 10:16]) + S)
    assert (np.abs(st[..., 10:16] - st_o[..., 10:16]) <= tol_g).all(), "gSD / gS1"
    assert (np.abs(st[..., 16:22] - st_o[..., 16:22]) <= 1e-3 * (np.abs(st_o[..., 16:22]) + np.abs(st_o[..., 16:22]).mean())).all(), "dSD / dS1"
    assert np.allclose(st[..., 3], st_o[..., 3], rtol=1e-5), "rho"
    # intervals: never wider than specified (rounded down), and the same up to 3 % on all but a handful of pixels (the crossing
    # prediction divides two small fp32 numbers when a sample sits on the horizon)
    for k in (1, 2):
        assert (st[..., k] <= st_o[..., k] * 1.02 + 1e-6).mean() > 0.995
        assert (np.abs(st[..., k] - st_o[..., k]) <= 0.03 * st_o[..., k] + 2e-5).mean() > 0.99
    jac = ops.jac16_unpack(j16, a)[:, 0].permute(1, 2, 0).cpu().numpy().astype(np.float64)   # [H,W,9] planes P, SD, JR
    assert (np.abs(jac - jac_o) <= 1.5e-3 * (np.abs(jac_o) + np.abs(jac_o).mean())).all(), "jac16 vs the specification"

    # second call: move r by up to +-0.01
    r2 = (r + _t(rng.uniform(-0.01, 0.01, (H, W, 1)), dev)).clamp(0.07, 1.0)
    exact2 = ops.shade_fwd(a, r2, m, n, light, spp, dcache=dcache)
    out2, _ = ops.shade_fwd_lazy(a, r2, m, n, light, spp, dcache, state, floor=floor)
    st2, ref2 = ops.lazy_state_unpack(state, a)
    out2_o, _, ref2_o = o.lazy_fwd(a64, r2.cpu().numpy().astype(np.float64), m64, n64, l64, st_o, spp, floor)
    ref2 = ref2[0].cpu().numpy()
    frac = ref2.mean()
    assert 0.02 < frac < 0.98, frac
    dr = np.abs(r2.cpu().numpy()[..., 0] - st[..., 0])
    edge = np.minimum(np.abs(dr - st[..., 1]), np.abs(dr - st[..., 2])) < 0.04 * np.maximum(st[..., 1], st[..., 2]) + 3e-5
    assert ((ref2 == ref2_o) | edge).all(), "a pixel well inside / outside its interval was (not) re-sampled"
    assert float(_rel(out2, exact2)) < 1e-3
    same = ref2 == ref2_o
    o2 = out2[0].cpu().numpy() if out2.ndim == 4 else out2.cpu().numpy()
    assert (np.abs(o2 - out2_o)[same] <= 3e-5 * (np.abs(out2_o)[same] + np.abs(out2_o).mean())).all(), "lazy render vs the specification"
    # the re-sampled pixels now sit at their new roughness
    st2 = st2[0].cpu().numpy()
    assert np.abs(st2[..., 0] - r2.cpu().numpy()[..., 0])[ref2 == 1].max() < 1e-6


def test_lazy_batch_equals_stand_alone_and_is_reproducible():
    from materialist_amd import ops, synthetic

    dev = _cuda()
    H, W, spp, B = 96, 80, 64, 3
    scs = [synthetic.make_scene(i, H, W) for i in range(B)]
    st = lambda k: _t(np.stack([getattr(s, k) for s in scs]), dev)
    a, r, m, light = st("albedo"), st("roughness"), st("metallic"), st("light")
    n = ops.normals_from_depth(st("depth"))
    dcache = ops.diffuse_cache(n, light, spp)
    floor = 0.3
    s_b = ops.lazy_state(a)
    ops.shade_fwd_lazy(a, r, m, n, light, spp, dcache, s_b, force=True, floor=floor)
    r2 = (r + 0.004 * torch.randn_like(r)).clamp(0.07, 1)
    nsum = ops._lib.load().matpbr_lazy_sums_count(H, W)
    sums = torch.zeros((B, nsum), device=dev)
    out_b, j_b = ops.shade_fwd_lazy(a, r2, m, n, light, spp, dcache, s_b, floor=floor, sums=sums)
    # the partial sums add up to the sum of the render
    assert torch.allclose(sums.sum(1), out_b.reshape(B, -1).sum(1), rtol=1e-5)
    for b in range(B):
        s1 = ops.lazy_state(a[b])
        dc1 = ops.diffuse_cache(n[b], light[b], spp)
        ops.shade_fwd_lazy(a[b], r[b], m[b], n[b], light[b], spp, dc1, s1, force=True, floor=floor)
        out1, j1 = ops.shade_fwd_lazy(a[b], r2[b], m[b], n[b], light[b], spp, dc1, s1, floor=floor)
        assert torch.equal(out_b[b], out1.reshape(out_b[b].shape)), b
        assert torch.equal(j_b[:, b], j1[:, 0]), b
    # same call sequence again: bit-identical
    s_c = ops.lazy_state(a)
    ops.shade_fwd_lazy(a, r, m, n, light, spp, dcache, s_c, force=True, floor=floor)
    out_c, _ = ops.shade_fwd_lazy(a, r2, m, n, light, spp, dcache, s_c, floor=floor)
    assert torch.equal(out_b, out_c) and torch.equal(s_b, s_c)


def _phase_setup(dev, H, W, spp, image_id=0):
    from materialist_amd import render, synthetic

    sc = synthetic.make_scene(image_id, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    return scene, gt, init


def test_lazy_render_stays_within_1e3_of_exact_sampling_on_every_pixel_of_every_iteration(oracle64):
    """THE GATE: 2 000 iterations of part 'rm' at 512 x 512 (lr schedule of inverse_img_w_mi.py:363-365,431-432), on the default path (the
    folded, persistent step).  Every iteration's lazy render (of the parameters the step before it wrote) against the exact render of the same
    parameters; d out / d r of the just re-sampled pixels against the exact jac; every 50th iteration the loss GRADIENTS the step formed
    (d loss / d r, d loss / d m, every pixel) against the streaming backward pass on the exact jac of the same parameters; and at iteration
    500 the render of a 64 x 64 window against the oracle's evaluation of the models as they stand (the specification, fp64)."""
    from materialist_amd import loop, ops

    dev = _cuda()
    H = W = 512
    spp, iters = 64, 2000
    scene, gt, init = _phase_setup(dev, H, W, spp)
    ph = loop.FusedBrdfPhase(scene, gt, *init, optimize_part="rm", spp=spp, lazy=True, history_len=iters, keep_grads=True)
    assert ph.lazy and ph.lazy_state is not None and ph.fold
    worst = torch.zeros((), device=dev)
    worst_dr = torch.zeros((), device=dev)
    worst_g = {k: {"max": 0.0, "l2": 0.0, "median": 0.0, "cos": 1.0} for k in ("roughness", "metallic")}
    per_check = {k: {"l2": [], "p999": []} for k in ("roughness", "metallic")}
    nref = torch.zeros(iters, device=dev)
    exact, jac = torch.empty_like(gt), ops.plane9(gt)
    g_ref = {k: torch.empty_like(v) for k, v in ph.g.items()}
    for it in range(iters):
        if it % 50 == 0:          # the parameters this step differentiates at, and the lazy loop's render of them (what its loss gradient is formed on)
            p_at = [ph.p[k].clone() for k in ("albedo", "roughness", "metallic")]
            pred_at = ph.pred.clone()
        ph.step()
        if it % 50 == 0:
            # the same gradient from the exact jac of these parameters, the statistics row the step committed (ratio, l1 / mse) and the
            # same regularisers / clamp gating: matpbr_brdf_loss_bwd_jac, the backward pass of the loop that walks every sample
            ops.shade_fwd(*p_at, ph.n, ph.light, spp, clamp_params=True, out=exact, dcache=ph.dcache, jac=jac)
            ops.brdf_loss_bwd_jac(*p_at, jac, exact, ph.gt_srgb, ph.stats, ph.orig["albedo"], ph.orig["roughness"], ph.orig["metallic"], 0.1,
                                  g_ref["albedo"], g_ref["roughness"], g_ref["metallic"], optimize_part="rm")
            for k, w in worst_g.items():
                e = (ph.g[k] - g_ref[k]).abs() / torch.maximum(g_ref[k].abs(), g_ref[k].abs().mean())
                w["max"] = max(w["max"], float(e.max()))
                w["median"] = max(w["median"], float(e.median()))
                w["l2"] = max(w["l2"], float((ph.g[k] - g_ref[k]).norm() / g_ref[k].norm()))
                w["cos"] = min(w["cos"], float((ph.g[k] * g_ref[k]).sum() / (ph.g[k].norm() * g_ref[k].norm())))
                per_check[k]["l2"].append(float((ph.g[k] - g_ref[k]).norm() / g_ref[k].norm()))
                per_check[k]["p999"].append(float(torch.quantile(e.reshape(-1)[::2].float(), 0.999)))
            # the JACOBIAN alone: the same backward pass on the exact planes but with d loss / d pred formed on the lazy loop's own render -- the L1
            # term of the loss carries sign(pred - gt), and at a converged pixel two renders that differ by 1e-4 (the gate allows 1e-3) disagree on
            # that sign, i.e. on the pixel's whole gradient: that is a property of comparing two renders, not of the models' derivatives
            if it > 0:
                ops.brdf_loss_bwd_jac(*p_at, jac, pred_at, ph.gt_srgb, ph.stats, ph.orig["albedo"], ph.orig["roughness"], ph.orig["metallic"], 0.1,
                                      g_ref["albedo"], g_ref["roughness"], g_ref["metallic"], optimize_part="rm")
                for k in ("roughness", "metallic"):
                    e = (ph.g[k] - g_ref[k]).abs() / torch.maximum(g_ref[k].abs(), g_ref[k].abs().mean())
                    per_check[k].setdefault("jl2", []).append(float((ph.g[k] - g_ref[k]).norm() / g_ref[k].norm()))
                    per_check[k].setdefault("jp999", []).append(float(torch.quantile(e.reshape(-1)[::2].float(), 0.999)))
                    per_check[k].setdefault("jmax", []).append(float(e.max()))
        if it == 500:
            # the specification on a window of the state as it stands: the oracle evaluates the (generic) models of these pixels at the
            # current parameters; pixels it would re-sample (their roughness sits on an interval's edge) are left out
            st, _ = ops.lazy_state_unpack(ph.lazy_state, ph.p["albedo"])
            i0, j0, n = 200, 300, 64
            win = lambda x: x.reshape(H, W, -1)[i0:i0 + n, j0:j0 + n].reshape(n * n, -1).cpu().numpy().astype(np.float64)
            ii, jj = np.meshgrid(np.arange(i0, i0 + n), np.arange(j0, j0 + n), indexing="ij")
            f = (0.5 * W) / np.tan(0.5 * np.deg2rad(35.0))
            wo = np.stack([(0.5 * (W - 1) - jj) / f, (ii - 0.5 * (H - 1)) / f, np.ones_like(ii, dtype=np.float64)], -1).reshape(-1, 3)
            wo /= np.linalg.norm(wo, axis=-1, keepdims=True)
            cm = ph.current_maps()
            st_w = np.ascontiguousarray(win(st[0]))
            out_o, _, ref_o = oracle64.lazy_fwd_lanes(win(cm["albedo"]), win(cm["roughness"]), win(cm["metallic"]), win(ph.n), wo,
                                                      ph.light.cpu().numpy().astype(np.float64), st_w, spp, 0.5 * float(gt.mean()))
            keep = ref_o == 0
            got = win(ph.pred)
            e_spec = np.abs(got - out_o)[keep] / (np.abs(out_o)[keep] + np.abs(out_o).mean())
            assert keep.mean() > 0.9 and e_spec.max() < 2e-4, (keep.mean(), e_spec.max())
        # the step's last launch has rendered the parameters it has just written (what the next iteration judges): `pred`
        pa, pr, pm = ph.p["albedo"].clone(), ph.p["roughness"].clone(), ph.p["metallic"].clone()
        ops.shade_fwd(pa, pr, pm, ph.n, ph.light, spp, clamp_params=True, out=exact, dcache=ph.dcache, jac=jac)
        worst = torch.maximum(worst, _rel(ph.pred, exact))
        # the pixels listed for re-sampling by this step's last launch (their new roughness left the model's interval)
        _, ref = ops.lazy_state_unpack(ph.lazy_state, pa)
        nref[it] = ref.float().mean() if it else 1.0
        if it % 100 == 1:
            # d out / d r right after a re-sampling: the same models, stand-alone, at the parameters as they are now
            pn = [ph.p[k].clone() for k in ("albedo", "roughness", "metallic")]
            st2 = ph.lazy_state.clone()
            _, j16 = ops.shade_fwd_lazy(*pn, ph.n, ph.light, spp, ph.dcache, st2, clamp_params=True, floor=0.5 * float(gt.mean()))
            _, ref2 = ops.lazy_state_unpack(st2, pa)
            ops.shade_fwd(*pn, ph.n, ph.light, spp, clamp_params=True, out=exact, dcache=ph.dcache, jac=jac)
            jl = ops.jac16_unpack(j16, pa)
            sel = ref2.reshape(-1) > 0
            if bool(sel.any()):
                e = (jl[6:9].reshape(3, -1) - jac[6:9].reshape(3, -1)).abs() / torch.maximum(jac[6:9].reshape(3, -1).abs(), jac[6:9].abs().mean())
                worst_dr = torch.maximum(worst_dr, e[:, sel].max())
    nref = nref.cpu().numpy()
    print(f"lazy gate: worst |lazy - exact| / scale over {iters} iterations = {float(worst):.3e}; d_r at refresh points {float(worst_dr):.3e}; "
          f"re-sampled fraction: first 100 its {nref[1:100].mean():.4f}, 100-500 {nref[100:500].mean():.4f}, 500-2000 {nref[500:].mean():.4f}")
    print("lazy gate: loss gradients against the exact backward pass, worst of 40 checks (every 50th iteration, every pixel): " +
          "; ".join(f"{k}: rel. L2 {w['l2']:.4f}, cosine {w['cos']:.6f}, median pixel {w['median']:.2e}, worst pixel {w['max']:.3f}" for k, w in worst_g.items()))
    print("lazy gate: per check (every 50th iteration) d_r rel. L2: " + " ".join(f"{v:.4f}" for v in per_check["roughness"]["l2"]))
    print("lazy gate: per check d_r 99.9th-percentile pixel: " + " ".join(f"{v:.4f}" for v in per_check["roughness"]["p999"]))
    print("lazy gate: per check d_r JACOBIAN ONLY rel. L2: " + " ".join(f"{v:.4f}" for v in per_check["roughness"]["jl2"]))
    print("lazy gate: per check d_r JACOBIAN ONLY 99.9th-percentile pixel: " + " ".join(f"{v:.4f}" for v in per_check["roughness"]["jp999"]))
    print("lazy gate: per check d_r JACOBIAN ONLY worst pixel: " + " ".join(f"{v:.3f}" for v in per_check["roughness"]["jmax"]))
    print("lazy gate: per check d_m JACOBIAN ONLY rel. L2: " + " ".join(f"{v:.5f}" for v in per_check["metallic"]["jl2"]))
    assert float(worst) <= 1e-3                    # (measured 5.9e-4; round 5: 8.1e-4)
    assert float(worst_dr) <= 2e-3
    # The DERIVATIVES of the models (d loss / d pred formed on the lazy loop's own render for both sides), every pixel, 39 checks.  Round 5: L2 7e-4,
    # 99.9th-percentile pixel 7.8e-3, worst pixel 3.1e-2 -- the tail was pixels just beyond a sample's horizon crossing (the interval's kink
    # allowance was stated on the VALUE; the crossing sample's share of the derivative weighs ~50 x more relative to d out / d r) and pixels at the
    # far end of intervals whose radius answered to the value's extrapolation error only; the folded slopes travelled in e5m2.  Round 6 (radius and
    # kink allowance also on the derivative, half-precision slopes, JA0 = JX0 + m_ref JY0; tolerances 5e-4 / 1e-3): L2 <= 3e-4, 99.9th percentile
    # <= 1.8e-3, worst pixel <= 5e-3.  What is left is the half-precision storage of the derivative words (2^-11 each, several per pixel) where
    # the channels' terms of d loss / d r cancel: tolerances of 2.5e-4 / 5e-4 and 1.5e-4 / 3e-4 (25 % and 60 % more pixels walked) leave the same
    # tail (1.7e-3 / 4e-3: tools/lazy_grad_diag.py).  Either term alone does not: the radius control without the kinks leaves worst pixels of
    # 2e-2, the kinks without the radius control 2.9e-2 (p99.9 5e-3).  Cost: 36 % more pixels walked per iteration (8 x 512 x 512, its. 300-1300).
    jr, jm = per_check["roughness"], per_check["metallic"]
    assert max(jr["jl2"]) <= 6e-4 and max(jr["jp999"]) <= 2.5e-3 and max(jr["jmax"]) <= 6e-3, (max(jr["jl2"]), max(jr["jp999"]), max(jr["jmax"]))
    assert max(jm["jl2"]) <= 6e-4 and max(jm["jp999"]) <= 3.5e-3, (max(jm["jl2"]), max(jm["jp999"]))      # (measured 3.8e-4, 2.4e-3: d out / d m = Y(dr), the value model)
    # The LOSS gradient of the lazy loop against the loss gradient on the exact render (below) differs by more, and by more the further the
    # part has converged (rel. L2 2e-4 at the first check, 4 % at iteration 1950): d loss / d pred carries sign(pred - gt) from the L1 term, and
    # where a pixel has converged the two renders (within 1e-3 of each other by the gate above) disagree on that sign -- the pixel's whole
    # gradient flips.  That is a property of comparing two renders, whatever produced them (two spp-8 renders that differ by one ulp show the
    # same, tests/test_gpu_parity.py); the bounds below keep the field statistics where round 4 measured them.
    assert worst_g["roughness"]["l2"] <= 6e-2 and worst_g["roughness"]["cos"] >= 0.998 and worst_g["roughness"]["median"] <= 5e-4, worst_g
    assert worst_g["metallic"]["l2"] <= 1.5e-2 and worst_g["metallic"]["cos"] >= 0.9999 and worst_g["metallic"]["median"] <= 6e-5, worst_g      # (median 2.8e-5: d out / d m = Y, whose Y0 keeps 15 mantissa bits since round 6)
    assert nref[0] == 1.0 and nref[1:].mean() < 0.1


@pytest.mark.parametrize("part", ["rm", "r", "m", "a", "arm"])
@pytest.mark.parametrize("masked", [False, True])
def test_statistics_formed_inside_the_step_are_the_statistics_of_its_render(part, masked):
    """Round 6: from its second iteration on a part of the render-ahead loop has no statistics launch -- the step that forms the render of iteration
    t + 1 leaves, per block, the sums from which the next step has the exposure ratio, the MSE (exactly: A + 2 e Bq + e^2 Cq around the ratio of
    iteration t) and the L1 (up to pixels whose sign flips inside e); the walked pixels' shares arrive as integer atomics.  ('arm' runs the generic
    step, whose statistics launch reads the stored render: the same check holds for it.)  Against the loss lines of the reference
    (inverse_img_w_mi.py:388-418) evaluated in fp64 on the render the loop itself reports for the same parameters (`pred`, formed from the models
    by matpbr_brdf_phase_resolve): ratio, MSE and L1 of every iteration, on an image of a size that is no multiple of anything, with and without
    pixels without geometry."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H, W, spp = 96, 131, 64
    sc = synthetic.make_scene(3, H, W)
    mask = None
    if masked:
        mask = torch.zeros(H, W, dtype=torch.bool)
        mask[:17] = True
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True, mesh_mask=mask)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    ph = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp)
    assert ph.lazy and ph.fold == (part != "arm")
    gt64, gts64 = ph.gt.double(), ph.gt_srgb.double()
    worst = 0.0
    ph.step()
    for it in range(2, 32):
        pred = ph.pred.double()                               # the render of the parameters iteration `it` will judge
        ph.step()
        st = ph.stats[0].double().cpu()
        ratio = gt64.sum() / pred.sum()
        x = (pred * ratio).clamp_min(1e-8) ** (1.0 / 2.2)
        want = {ops.STAT_RATIO: float(ratio), ops.STAT_MSE: float(((x - gts64) ** 2).mean()), ops.STAT_L1: float((x - gts64).abs().mean())}
        for k, v in want.items():
            worst = max(worst, abs(float(st[k]) / v - 1.0))
    from test_gpu_parity import _report

    _report("in-step statistics against fp64 on the loop's own render, iterations 2-31: worst relative difference of ratio / MSE / L1", worst, 5e-6)
    assert worst <= 5e-6, worst


@pytest.mark.parametrize("part", ["rm", "a", "arm"])
def test_savebest_without_copies_leaves_what_the_copying_step_leaves(part):
    """MATPBR_FLAG_ROTATE_BEST (the default of the lazy loop): parameters, SaveBest's maps and image, statistics and history against the
    form whose step kernel copies the snapshot in every improving iteration -- bit for bit, at several points of the phase (a resolve in
    the middle of a phase must not disturb what follows), with EarlyStopping armed; and a phase that never improves on the best_mse it
    is given: its best maps and best image must stay what they were."""
    from materialist_amd import loop, ops

    dev = _cuda()
    H = W = 96
    spp = 64
    scene, gt, init = _phase_setup(dev, H, W, spp, image_id=2)
    best0 = torch.tensor([1e30], device=dev)
    kw = dict(optimize_part=part, spp=spp, patience=40, min_delta=1e-3, history_len=400)
    rot = loop.FusedBrdfPhase(scene, gt, *init, best_mse=best0, **kw)
    cpy = loop.FusedBrdfPhase(scene, gt, *init, best_mse=best0, rotate_best=False, **kw)
    assert rot.rotate and not cpy.rotate
    done = 0
    for upto in (1, 2, 7, 60, 61, 300):
        rot.run(upto - done)
        cpy.run(upto - done)
        done = upto
        for name in ("albedo", "roughness", "metallic"):
            assert torch.equal(rot.p[name], cpy.p[name]), (upto, name)
            assert torch.equal(rot.best[name], cpy.best[name]), (upto, name)
        if part == "rm":
            # a folded phase stores no renders: the rotating form's best image is formed when it is asked for, from the RENDERER on SaveBest's
            # maps where the roughness has moved (matpbr_brdf_phase_resolve) -- the copying form keeps the model's render of the improving
            # iteration, within the models' tolerance of it (1e-3 of max(|.|, mean) in linear radiance, 1 / 2.2 of that after the gamma)
            err = (rot.best_img - cpy.best_img).abs() / torch.maximum(cpy.best_img, cpy.best_img.mean())
            assert float(err.max()) <= 1e-3, (upto, float(err.max()))
        else:
            assert torch.equal(rot.best_img, cpy.best_img), upto
        assert torch.equal(rot.pred, cpy.pred), upto              # one meaning in both forms: the render of the current parameters
        assert torch.equal(rot.stats, cpy.stats), upto
        assert torch.equal(rot.history(), cpy.history()), upto
    if part == "rm":
        # ... and IS the renderer's image of SaveBest's maps: best_img^2.2 / shade(best maps) is one number per image (the best iteration's exposure ratio)
        lin = torch.empty_like(gt)
        ops.shade_fwd(init[0].clamp(0, 1), rot.best["roughness"], rot.best["metallic"], rot.n, rot.light, spp, clamp_params=True, out=lin, dcache=rot.dcache)
        q = (rot.best_img.double() ** 2.2) / lin.double().clamp_min(1e-12)
        sel = lin > 1e-4
        assert float(q[sel].std() / q[sel].mean()) <= 2e-6, float(q[sel].std() / q[sel].mean())
    # an image that cannot improve: the snapshots it came with survive the phase
    tiny = torch.tensor([0.0], device=dev)
    keep = loop.FusedBrdfPhase(scene, gt, *init, best_mse=tiny, optimize_part=part, spp=spp)
    b0 = {k: v.clone() for k, v in keep.best.items()}
    img0 = keep.best_img.clone()
    keep.run(25)
    assert all(torch.equal(keep.best[k], b0[k]) for k in b0) and torch.equal(keep.best_img, img0)
    assert float(keep.stats[0, ops.STAT_BEST]) == 0.0


def test_lazy_phase_lands_where_the_phase_that_walks_every_sample_lands():
    """400 iterations of 'rm' then 'arm' style steps, lazy against exact sampling: the loss curves agree to 1 % and the maps end
    within a fraction of the distance travelled."""
    from materialist_amd import loop, ops

    dev = _cuda()
    H = W = 256
    spp, iters = 64, 400
    scene, gt, init = _phase_setup(dev, H, W, spp, image_id=1)
    for part in ("rm", "arm"):
        lz = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp, lazy=True, history_len=iters)
        ex = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp, lazy=False, history_len=iters)
        lz.run(iters)
        ex.run(iters)
        h_l, h_e = lz.history()[:, 0].cpu().numpy(), ex.history()[:, 0].cpu().numpy()
        assert h_e[-1] < 0.8 * h_e[0]
        assert np.abs(h_l - h_e).max() <= 0.01 * h_e.max(), part
        assert abs(h_l[-1] - h_e[-1]) <= 0.01 * h_e[-1], part
        for k in ("roughness", "metallic") + (("albedo",) if "a" in part else ()):
            moved = (ex.p[k] - init[{"albedo": 0, "roughness": 1, "metallic": 2}[k]]).abs().mean().item()
            diff = (lz.p[k] - ex.p[k]).abs().mean().item()
            assert diff < 0.05 * moved + 1e-5, (part, k, diff, moved)
        assert float(lz.stats[0, ops.STAT_BEST]) == pytest.approx(float(ex.stats[0, ops.STAT_BEST]), rel=0.01)


def test_fixed_roughness_parts_run_on_the_models_without_ever_resampling():
    """Parts that leave the roughness alone ('a', 'am', 'm' of --opt_order): on the per-pixel models no pixel ever leaves its interval, the
    iteration is the same two launches as in the 'rm' part; against the phase that reuses the walked specular sums (lazy=False)."""
    from materialist_amd import loop, ops

    dev = _cuda()
    H = W = 128
    spp, iters = 64, 60
    scene, gt, init = _phase_setup(dev, H, W, spp, image_id=2)
    for part in ("a", "am"):
        lz = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp, history_len=iters)
        ex = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp, lazy=False, history_len=iters)
        assert lz.lazy and lz.s1cache is None and ex.s1cache is not None
        lz.run(iters)
        ex.run(iters)
        h_l, h_e = lz.history()[:, 0].cpu().numpy(), ex.history()[:, 0].cpu().numpy()
        assert np.abs(h_l - h_e).max() <= 2e-4 * h_e.max(), part
        for k in ("albedo", "metallic"):
            d = (lz.p[k] - ex.p[k]).abs()
            assert float((d < 5e-5).float().mean()) > 0.99 and float(d.mean()) < 1e-5, (part, k, float(d.max()))
        assert torch.equal(lz.p["roughness"], init[1].reshape(lz.p["roughness"].shape))
        _, refreshed = ops.lazy_state_unpack(lz.lazy_state, lz.p["albedo"])
        assert int(refreshed.sum()) == 0                       # nothing was re-sampled in the last launch (nor in any after the first)


@pytest.mark.parametrize("part", ["rm", "r", "m", "a"])
def test_folded_persistent_step_is_the_generic_step(part):
    """The folded, persistent step (csrc/matpbr_pstep.hpp: the maps a part leaves alone folded into the per-pixel models, listed pixels
    walked inside the step launch) against the generic step + resampling launch on the same part: the same loss curve, the same pixels
    re-sampled (up to the few whose roughness sits on an interval's edge), parameters, SaveBest's maps and image within the rounding of the two
    expressions pushed through 150 Adam steps.  An image size that is not a multiple of the 512-pixel block exercises the ragged tail."""
    from materialist_amd import loop, ops

    dev = _cuda()
    H, W, spp, iters = 96, 131, 64, 150
    scene, gt, init = _phase_setup(dev, H, W, spp, image_id=3)
    kw = dict(optimize_part=part, spp=spp, history_len=iters, patience=50, min_delta=1e-3)
    fo = loop.FusedBrdfPhase(scene, gt, *init, fold=True, **kw)
    ge = loop.FusedBrdfPhase(scene, gt, *init, fold=False, **kw)
    assert fo.fold and not ge.fold
    n_f = n_g = 0
    for it in range(iters):
        fo.step()
        ge.step()
        _, rf = ops.lazy_state_unpack(fo.lazy_state, fo.p["albedo"])
        _, rg = ops.lazy_state_unpack(ge.lazy_state, ge.p["albedo"])
        n_f, n_g = n_f + int(rf.sum()), n_g + int(rg.sum())
        if it in (0, 1, 10, iters - 1):      # the render of the current parameters (what the next iteration judges); Adam normalises the
            # gradient, so where it is nearly zero rounding decides a step's sign and single pixels part ways over 150 steps
            worst = (fo.pred - ge.pred).abs() / ge.pred.abs().mean()
            assert float(worst.max()) < {0: 4e-5, 1: 2e-4, 10: 1e-3}.get(it, 3e-2) and float(worst.mean()) < 2e-4, (it, float(worst.max()), float(worst.mean()))      # (it 0: the folded X0 / Y0 keep 15 mantissa bits, 2^-16 of terms that partly cancel)
    h_f, h_g = fo.history()[:, 0].cpu().numpy(), ge.history()[:, 0].cpu().numpy()
    assert np.abs(h_f - h_g).max() <= 5e-4 * h_g.max()
    assert abs(n_f - n_g) <= 0.01 * n_g + 2 and (n_g > 0) == ("r" in part)
    for src_f, src_g in ((fo.p, ge.p), (fo.best, ge.best)):
        for k in ("albedo", "roughness", "metallic"):
            d = (src_f[k] - src_g[k]).abs()
            assert float(d.mean()) < 1e-4 and float((d < 5e-4).float().mean()) > 0.98, (part, k, float(d.mean()), float(d.max()))
    assert float((fo.best_img - ge.best_img).abs().max()) < 5e-3
    assert torch.allclose(fo.stats, ge.stats, rtol=2e-3, atol=1e-6)
    # maps the part does not move are left alone, bit for bit
    for k, ch in (("albedo", "a"), ("roughness", "r"), ("metallic", "m")):
        if ch not in part:
            assert torch.equal(fo.p[k], init[{"albedo": 0, "roughness": 1, "metallic": 2}[k]].reshape(fo.p[k].shape)), k


def test_folded_step_gradients_are_the_generic_steps():
    """`keep_grads`: the folded step forms the gradients of the maps its part moves; = the generic step's at the same parameters (the first
    iteration of a phase: both start from the same models)."""
    from materialist_amd import loop

    dev = _cuda()
    H, W, spp = 128, 128, 64
    scene, gt, init = _phase_setup(dev, H, W, spp, image_id=5)
    for part, keys in (("rm", ("roughness", "metallic")), ("a", ("albedo",))):
        g = {}
        for fold in (True, False):
            ph = loop.FusedBrdfPhase(scene, gt, *init, optimize_part=part, spp=spp, keep_grads=True, fold=fold)
            assert ph.fold == fold
            ph.step()
            g[fold] = {k: ph.g[k].clone() for k in keys}
        for k in keys:
            e = (g[True][k] - g[False][k]).abs() / torch.maximum(g[False][k].abs(), g[False][k].abs().mean())
            # d out / d r travels in half precision in both forms (generic: dSD, dS1 each; folded: their combinations JX0, JY0)
            assert float(e.max()) < (2e-3 if k == "roughness" else 2e-4), (part, k, float(e.max()))


def test_lazy_phase_early_stopping_and_batch():
    """The device-side EarlyStopping and the per-image skip work the same on the lazy path; a batched lazy phase equals its images
    run alone, bit for bit."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H, W, spp, B = 64, 96, 16, 8             # eight images: the per-GPU shard of BASELINE configs[2] (workspace regions are sized per image)
    scs = [synthetic.make_scene(10 + i, H, W) for i in range(B)]
    st = lambda k: _t(np.stack([getattr(s, k) for s in scs]), dev)
    scene_b = render.load_estimated_mesh(st("depth"), use_mesh_normal=True)
    scene_b._set("emitter.data", st("light"))
    with torch.no_grad():
        gt = render.render_w_brdf(scene_b, st("albedo"), st("roughness"), st("metallic"), None, spp)
    init = [st(k) for k in ("init_albedo", "init_roughness", "init_metallic")]
    fb = loop.FusedBrdfPhase(scene_b, gt, *init, optimize_part="rm", spp=spp, lazy=True)
    fb.run(30)
    for b in range(B):
        s1 = render.load_estimated_mesh(st("depth")[b], use_mesh_normal=True)
        s1._set("emitter.data", st("light")[b])
        f1 = loop.FusedBrdfPhase(s1, gt[b], *[x[b] for x in init], optimize_part="rm", spp=spp, lazy=True)
        f1.run(30)
        for k in ("roughness", "metallic"):
            assert torch.equal(fb.p[k][b], f1.p[k]), (b, k)
        assert torch.equal(fb.stats[b], f1.stats[0]), b               # every slot: loss terms, SaveBest, EarlyStopping state, iteration count
        assert torch.equal(fb.history()[:, b], f1.history()[:, 0]), b
    es = loop.FusedBrdfPhase(scene_b, gt, *init, optimize_part="arm", spp=spp, patience=4, min_delta=0.5, lazy=True)
    es.run(12)
    info = es.poll()
    assert info["stopped"].tolist() == [True] * B and info["iters"].tolist() == [5] * B
    frozen = {k: v.clone() for k, v in es.p.items()}
    es.run(3)
    assert all(torch.equal(frozen[k], es.p[k]) for k in frozen)


def test_lazy_phase_offers_the_attached_roughness_gradient():
    """`attached_sampling=True`: d loss / d r of the fused loop follows the derivative of the rendered value THROUGH the GGX sample
    directions (the live reference's convention; the models' slopes are that derivative), = matpbr_shade_bwd(MATPBR_FLAG_ATTACHED_SAMPLING)
    chained with the same loss gradient; the default stays the stop-gradient convention."""
    from materialist_amd import loop, ops

    dev = _cuda()
    H = W = 128
    spp = 64
    scene, gt, init = _phase_setup(dev, H, W, spp, image_id=4)
    got = {}
    for att in (False, True):
        ph = loop.FusedBrdfPhase(scene, gt, *init, optimize_part="rm", spp=spp, lazy=True, keep_grads=True, attached_sampling=att)
        pa, pr, pm = (ph.p[k].clone() for k in ("albedo", "roughness", "metallic"))
        ph.step()
        got[att] = ph.g["roughness"].clone()
        if att:
            # the loss gradient w.r.t. the render at these parameters, from the torch composition, pushed through the operator face
            pred = ops.shade_fwd(pa, pr, pm, ph.n, ph.light, spp, clamp_params=True).requires_grad_(True)
            lo, _, _, _ = loop._loss.brdf_loss(pred, gt, {}, {}, 0.1, ph.gt_srgb)
            (d_pred,) = torch.autograd.grad(lo, pred)
            ref = {a: ops.shade_bwd(pa.clamp(0, 1), pr.clamp(0.07, 1), pm.clamp(0, 1), ph.n, ph.light, d_pred.contiguous(), spp, attached=a)[1] for a in (False, True)}
    # stop-gradient default: the jac of the models (half precision) against the exact kernels, every pixel
    e = (got[False] - ref[False]).abs() / torch.maximum(ref[False].abs(), ref[False].abs().mean())
    assert float(e.max()) < 3e-3
    # attached: the slopes are one-sided differences over h = 1e-3, the operator face differentiates analytically at the point: equal up to
    # half a per cent on the typical pixel, apart where a sample crosses the horizon within h (the derivative jumps there)
    e = (got[True] - ref[True]).abs() / torch.maximum(ref[True].abs(), ref[True].abs().mean())
    assert float(e.flatten().median()) < 1e-2 and float((got[True] - ref[True]).norm() / ref[True].norm()) < 0.08
    # the two conventions differ measurably, and each run follows its own
    assert float((ref[True] - ref[False]).norm() / ref[False].norm()) > 0.2
    assert float((got[True] - ref[True]).norm()) < 0.3 * float((got[True] - ref[False]).norm())


def test_pixels_without_geometry_run_in_the_fused_loops():
    """mesh_mask.png scenes (inverse_img_w_mi.py:713-724) in the fused loops: a pixel whose camera ray leaves the scene is given a constant
    model (P = S0-S1 = 0, S1 = radiance of the light along the ray), so it renders the environment, hands no gradient to its materials
    and is never re-sampled.  The fused BRDF phases (lazy 'rm', cached 'a') and the fused env phase against the operator-face compositions
    (`BrdfPhase`, `EnvHeadPhase`, which compose the background with torch.where)."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H, W, spp = 64, 96, 16
    sc = synthetic.make_scene(12, H, W)
    mask = torch.zeros(H, W, dtype=torch.bool)
    mask[:17] = True
    mask[30:37, 40:61] = True
    m = mask.to(dev)

    def scene_():
        s = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True, mesh_mask=mask)
        s._set("emitter.data", _t(sc.light, dev))
        return s

    with torch.no_grad():
        gt = render.render_w_brdf(scene_(), _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    bg = (scene_().bg_basis @ _t(sc.light, dev)).reshape(H, W, 3)
    for part in ("rm", "a"):
        ref = loop.BrdfPhase(scene_(), gt, *init, None, optimize_part=part, spp=spp)
        fused = loop.FusedBrdfPhase(scene_(), gt, *init, optimize_part=part, spp=spp)
        assert fused.lazy                                      # every part runs on the models (no pixel of a fixed-roughness part is ever re-sampled)
        for it in range(6):
            mse_ref = ref.step()
            fused.step()
            assert float(fused.stats[0, ops.STAT_MSE]) == pytest.approx(float(mse_ref), rel=5e-4), (part, it)
            assert float(fused.stats[0, ops.STAT_LOSS]) == pytest.approx(float(ref.last["loss"]), rel=5e-4), (part, it)
            assert torch.allclose(fused.pred[m], bg[m], rtol=2e-5, atol=1e-7)      # (the folded models carry X0 = the background in 24 bits: 15 mantissa bits, rounded to nearest: 2^-16)
        for k, j in (("albedo", 0), ("roughness", 1), ("metallic", 2)):
            assert torch.equal(fused.p[k][m], init[j][m]), (part, k)                    # no gradient, no drift behind the mask
            if k in ref.params:
                d = (fused.p[k] - ref.params[k].detach()).abs()
                if fused.lazy:      # half-precision jac: where a gradient is all but zero its sign, and with it Adam's first steps, may differ
                    assert float((d < 5e-5).float().mean()) > 0.99 and float(d.mean()) < 1e-5, (part, k)
                else:
                    assert d.max().item() < 5e-5, (part, k)
        if fused.lazy:
            _, refreshed = ops.lazy_state_unpack(fused.lazy_state, fused.p["albedo"])
            assert int(refreshed.reshape(H, W)[m].sum()) == 0
    # hot loop A on the transfer: masked pixels carry the SH basis along their ray
    from materialist_amd import posmlp

    s_env = scene_()
    pr = render.traverse(s_env)
    pr["shape.bsdf.a"], pr["shape.bsdf.r"], pr["shape.bsdf.m"] = _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev)
    light = torch.nn.Parameter(_t(sc.light, dev) * 0.7)
    opt = torch.optim.Adam([light], lr=1e-2)
    fe = loop.FusedEnvPhase(s_env, gt, lambda: light, opt, spp=spp)
    fe.step()
    with torch.no_grad():
        s_env._set("emitter.data", light.detach() + 0.0)
    # its render and loss at the first iterate = the operator face's
    s_ref = scene_()
    pr = render.traverse(s_ref)
    pr["shape.bsdf.a"], pr["shape.bsdf.r"], pr["shape.bsdf.m"] = _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev)
    l0 = (_t(sc.light, dev) * 0.7).requires_grad_(True)
    pred = render.render_envmap(s_ref, l0, spp)
    lo, mse, _ = loop._loss.env_loss(pred, gt)
    lo.backward()
    assert float(fe.stats[0, ops.STAT_MSE]) == pytest.approx(float(mse), rel=2e-4)
    assert torch.allclose(fe.pred, pred.detach(), rtol=2e-4, atol=1e-6)
    assert (fe.d_light.reshape(25, 3) - l0.grad).abs().max().item() <= 2e-4 * float(l0.grad.abs().max())


def test_a_batch_with_per_image_background_masks_equals_the_images_alone():
    """A shard of photographs whose meshes leave different pixels uncovered (run_batch.py, pipeline.inverse_images_batched): the masks ride
    in the batch dimension; every image of the batch goes through the fused BRDF and env phases exactly as it does alone."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    H, W, spp = 64, 96, 16
    scs = [synthetic.make_scene(20 + k, H, W) for k in range(2)]
    masks = torch.zeros(2, H, W, dtype=torch.bool)
    masks[0, :11] = True
    masks[1, 40:, 70:] = True
    masks[1, 5:9, 5:30] = True

    def scene_(ks):
        depth = torch.stack([_t(scs[k].depth, dev) for k in ks])
        mk = masks[list(ks)]
        s = render.load_estimated_mesh(depth if len(ks) > 1 else depth[0], use_mesh_normal=True, mesh_mask=mk if len(ks) > 1 else mk[0])
        light = torch.stack([_t(scs[k].light, dev) for k in ks])
        s._set("emitter.data", light if len(ks) > 1 else light[0])
        return s

    def maps(ks, names):
        out = [torch.stack([_t(getattr(scs[k], nm), dev) for k in ks]) for nm in names]
        return out if len(ks) > 1 else [x[0] for x in out]

    true_names, init_names = ("albedo", "roughness", "metallic"), ("init_albedo", "init_roughness", "init_metallic")
    with torch.no_grad():
        gt_b = render.render_w_brdf(scene_((0, 1)), *maps((0, 1), true_names), None, spp)
        for k in range(2):
            gt_k = render.render_w_brdf(scene_((k,)), *maps((k,), true_names), None, spp)
            assert torch.equal(gt_b[k], gt_k)
            bg = scene_((k,)).background_radiance(_t(scs[k].light, dev))
            assert torch.equal(gt_k[masks[k].to(dev)], bg[masks[k].to(dev)])
    for part in ("rm", "a"):
        batch = loop.FusedBrdfPhase(scene_((0, 1)), gt_b, *maps((0, 1), init_names), optimize_part=part, spp=spp)
        alone = [loop.FusedBrdfPhase(scene_((k,)), gt_b[k].contiguous(), *maps((k,), init_names), optimize_part=part, spp=spp) for k in range(2)]
        for it in range(8):
            batch.step()
            for ph in alone:
                ph.step()
        for k in range(2):
            for name in ("albedo", "roughness", "metallic"):
                assert torch.equal(batch.p[name][k], alone[k].p[name]), (part, k, name)
            assert torch.equal(batch.pred[k], alone[k].pred), (part, k)
            assert torch.equal(batch.stats[k, : ops.STAT_BEST + 1], alone[k].stats[0, : ops.STAT_BEST + 1]), (part, k)
    # hot loop A: one light per image, the masked pixels' transfer rows are the SH basis along their rays
    lights = torch.nn.Parameter(torch.stack([_t(s.light, dev) for s in scs]) * 0.7)
    s_b = scene_((0, 1))
    pr = render.traverse(s_b)
    pr["shape.bsdf.a"], pr["shape.bsdf.r"], pr["shape.bsdf.m"] = maps((0, 1), true_names)
    fe = loop.FusedEnvPhase(s_b, gt_b, lambda: lights, torch.optim.Adam([lights], lr=1e-2), spp=spp)
    fe.step()
    for k in range(2):
        s_k = scene_((k,))
        pr = render.traverse(s_k)
        pr["shape.bsdf.a"], pr["shape.bsdf.r"], pr["shape.bsdf.m"] = maps((k,), true_names)
        l0 = (_t(scs[k].light, dev) * 0.7).requires_grad_(True)
        pred = render.render_envmap(s_k, l0, spp)
        lo, mse, _ = loop._loss.env_loss(pred, gt_b[k])
        lo.backward()
        assert float(fe.stats[k, ops.STAT_MSE]) == pytest.approx(float(mse), rel=2e-4)
        assert torch.allclose(fe.pred[k], pred.detach(), rtol=2e-4, atol=1e-6)
        assert (fe.d_light[k] - l0.grad).abs().max().item() <= 2e-4 * float(l0.grad.abs().max())


@pytest.mark.parametrize("part", ["rm", "a", "arm"])
def test_a_batch_cut_into_groups_on_streams_of_their_own_is_the_batch(part):
    """loop.PipelinedBrdfPhase (groups of images stepping on their own streams, the step on 512 workgroups so that another group's walk and
    statistics launches fit beside it) = one FusedBrdfPhase over the whole batch, bit for bit: parameters, SaveBest's maps and image, history,
    statistics, the render -- with EarlyStopping armed."""
    from materialist_amd import loop, render, synthetic

    dev = _cuda()
    B, H, W, spp = 4, 128, 160, 16
    scs = [synthetic.make_scene(40 + i, H, W) for i in range(B)]
    st = lambda f: torch.stack([_t(f(s), dev) for s in scs])

    def make_scene():
        s = render.load_estimated_mesh(st(lambda s: s.depth), use_mesh_normal=True)
        s._set("emitter.data", st(lambda s: s.light))
        return s

    with torch.no_grad():
        gt = render.render_w_brdf(make_scene(), st(lambda s: s.albedo), st(lambda s: s.roughness), st(lambda s: s.metallic), None, spp)
    init = [st(lambda s: s.init_albedo), st(lambda s: s.init_roughness), st(lambda s: s.init_metallic)]
    kw = dict(optimize_part=part, spp=spp, patience=5, min_delta=0.02)
    one = loop.FusedBrdfPhase(make_scene(), gt, *init, **kw)
    two = loop.PipelinedBrdfPhase(make_scene(), gt, *init, groups=2, **kw)
    for n in (1, 7, 40):
        one.run(n)
        two.run(n)
        assert torch.equal(one.stats, two.stats), n
    for k in ("albedo", "roughness", "metallic"):
        assert torch.equal(one.p[k], two.p[k]) and torch.equal(one.best[k], two.best[k]), k
    assert torch.equal(one.best_img, two.best_img) and torch.equal(one.pred, two.pred)
    assert torch.equal(one.history(), two.history())
    a, b = one.poll(), two.poll()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert two.t == one.t == 48 and two.lr_at(5) == one.lr_at(5)
    # the groups' streams are the process's, not the phase's: a run builds a phase per part, and HIP has only a few hardware queues to spread a
    # process's streams over (a second phase on streams of its own lost the overlap of its groups: bench.py --mode fused, 70 k instead of 110 k)
    again = loop.PipelinedBrdfPhase(make_scene(), gt, *init, groups=2, **kw)
    assert all(x is y for x, y in zip(again.streams, two.streams)) and len(again.streams) == 2
    again.run(48)
    assert torch.equal(again.stats, two.stats) and all(torch.equal(again.p[k], two.p[k]) for k in ("albedo", "roughness", "metallic"))

