"""BASELINE.json's configurations at their full sizes on the GPU (the parity tests proper run at sizes the oracle finishes in
seconds; here the full-size runs are tied back to them through crops, size-independent properties and trace shapes).

  configs[0]  256x256, --model_name none --opt_order arm (plumbing)
  configs[2]  batch of 8 x 512x512 per GPU: the kernels' batch dimension
  configs[3]  1024x1024 image through MaterialNet (ViT-B as shipped; ViT-L as the random-weight throughput variant)
  configs[4]  2048x2048 rolling relight, 360 envmap frames
"""
import os

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


def _close(got, ref, rtol, what):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64).reshape(got.shape)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), np.abs(ref).mean() + 1e-30)
    assert err.max() <= rtol, f"{what}: max scaled rel err {err.max():.3e}"


# ---------------------------------------------------------------------------------------------- configs[2]
def test_batch_of_8_at_512_matches_stand_alone_renders_and_the_oracle_on_a_crop(oracle64):
    """The C3 per-GPU shard: 8 independent 512x512 images in one launch.  Every image equals its stand-alone render bit for
    bit (forward, jac-based material gradients, cached-diffuse variant); a 64x64 window of one of them equals the fp64 oracle
    evaluated with the full image's view directions; one fused optimisation iteration on the batch equals the single-image one."""
    from materialist_amd import loop, ops, render, synthetic

    dev = _cuda()
    B, H, W, spp = 8, 512, 512, 64
    scs = [synthetic.make_scene(i, H, W) for i in range(B)]
    st = lambda k: _t(np.stack([getattr(s, k) for s in scs]), dev)
    depth = st("depth")
    n = ops.normals_from_depth(depth)
    a, r, m, light = st("albedo"), st("roughness"), st("metallic"), st("light")
    out = ops.shade_fwd(a, r, m, n, light, spp)
    dcache = ops.diffuse_cache(n, light, spp)
    jac = ops.plane9(a)
    out_c = ops.shade_fwd(a, r, m, n, light, spp, dcache=dcache, jac=jac)
    d_out = torch.randn_like(out)
    g = ops.shade_bwd_jac(a, r, m, jac, d_out)
    for b in (0, 3, 7):
        one = ops.shade_fwd(a[b], r[b], m[b], n[b], light[b], spp)
        assert torch.equal(out[b], one), f"image {b}: batched render differs from the stand-alone one"
        dc1 = ops.diffuse_cache(n[b], light[b], spp)
        j1 = ops.plane9(a[b])
        assert torch.equal(out_c[b], ops.shade_fwd(a[b], r[b], m[b], n[b], light[b], spp, dcache=dc1, jac=j1))
        g1 = ops.shade_bwd_jac(a[b], r[b], m[b], j1, d_out[b].contiguous())
        assert all(torch.equal(x[b], y.reshape(x[b].shape)) for x, y in zip(g, g1))
    # cached diffuse lobe == in-kernel diffuse lobe (the same arithmetic up to fp32 contraction order)
    assert (out - out_c).abs().max().item() <= 2e-6 * out.abs().max().item()
    # oracle on a 64x64 window of image 5, view directions of the full image
    b, i0, j0, w = 5, 301, 77, 64
    sl = (slice(i0, i0 + w), slice(j0, j0 + w))
    n5 = n[b].cpu().numpy().astype(np.float64)
    ref = oracle64.shade_fwd_win(scs[b].albedo[sl], scs[b].roughness[sl], scs[b].metallic[sl], n5[sl], scs[b].light, spp, H, W, i0, j0)
    _close(out[b][sl], ref, 1e-3, "512x512 render vs oracle on a 64x64 window")
    # one fused iteration of hot loop B on the batch == the same iteration on image 2 alone
    scene_b = render.load_estimated_mesh(depth, use_mesh_normal=True)
    scene_b._set("emitter.data", light)
    init = [st(k) for k in ("init_albedo", "init_roughness", "init_metallic")]
    gt = out
    fb = loop.FusedBrdfPhase(scene_b, gt, *init, optimize_part="rm", spp=spp, lazy=False)
    fb.run(2)
    scene_1 = render.load_estimated_mesh(depth[2], use_mesh_normal=True)
    scene_1._set("emitter.data", light[2])
    f1 = loop.FusedBrdfPhase(scene_1, gt[2], *[x[2] for x in init], optimize_part="rm", spp=spp, lazy=False)
    f1.run(2)
    for k in ("roughness", "metallic"):
        assert torch.equal(fb.p[k][2], f1.p[k]), k
    assert fb.poll()["iters"].tolist() == [2] * B
    assert float(fb.stats[2, ops.STAT_MSE]) == float(f1.stats[0, ops.STAT_MSE])


# ---------------------------------------------------------------------------------------------- configs[4]
def test_rolling_relight_2048_360_frames():
    """render_final.py --mode rolling at 2048x2048, 360 frames of 1 degree: frames {0, 179, 359} through the precomputed transfer
    equal the direct render under the rotated light; rotating the SH light about +y equals rolling the envmap columns."""
    from materialist_amd import ops, sh, synthetic

    dev = _cuda()
    S, spp = 2048, 64
    sc = synthetic.make_scene(0, S, S)
    n = ops.normals_from_depth(_t(sc.depth, dev))
    a, r, m = _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev)
    T = ops.shade_transfer(a, r, m, n, spp)
    c0 = sc.light.astype(np.float64)
    lights = np.stack([sh.rotate_y_matrix(np.radians(f)) @ c0 for f in range(360)])
    L = _t(lights, dev)
    out8 = torch.empty(8, S, S, 3, device=dev)
    for f in (0, 179, 359):
        f0 = (f // 8) * 8
        ops.relight(T, L[f0:f0 + 8].contiguous(), S, S, out8)
        direct = ops.shade_fwd(a, r, m, n, L[f].contiguous(), spp)
        assert (out8[f - f0] - direct).abs().max().item() <= 3e-5 * direct.abs().max().item(), f"frame {f}"
        assert torch.isfinite(out8[f - f0]).all()
    # a frame is the same bits however many share its pass over the transfer (24 per pass since round 4: matpbr_relight's chunk; 8; 1)
    out24 = torch.empty(24, S, S, 3, device=dev)
    ops.relight(T, L[:24].contiguous(), S, S, out24)
    ops.relight(T, L[:8].contiguous(), S, S, out8)
    assert torch.equal(out24[:8], out8)
    ops.relight(T, L[16:24].contiguous(), S, S, out8)
    assert torch.equal(out24[16:], out8)
    assert torch.equal(ops.relight(T, L[13:14].contiguous(), S, S)[0], out24[13])
    assert torch.equal(ops.relight(T, L[3:16].contiguous(), S, S), out24[3:16])          # 13 frames: a partly filled pass of the 24-frame kernel
    assert torch.equal(ops.relight(T, L[:31].contiguous(), S, S)[24:], ops.relight(T, L[24:31].contiguous(), S, S))   # 24 + 7
    # all 360 frames in 15 passes: finite, and the mean radiance is invariant under rotations about the pole up to the
    # view-dependence of the image (sanity: within 20 % of frame 0)
    means = []
    for f0 in range(0, 360, 24):
        ops.relight(T, L[f0:f0 + 24].contiguous(), S, S, out24)
        means.append(out24.mean(dim=(1, 2, 3)).cpu())
    means = torch.cat(means)
    assert means.numel() == 360
    assert torch.isfinite(means).all() and float((means / means[0] - 1).abs().max()) < 0.2
    # rotate_y == column roll of the 16x32 envmap (render_final.py:290-298), whole columns: 360/32 = 11.25 degrees
    rng = np.random.default_rng(2)
    env = rng.random((16, 32, 3)) + 0.1
    P = sh.envmap_to_sh_matrix(16, 32)
    for shift in (1, 7, 31):
        np.testing.assert_allclose(sh.rotate_y_matrix(2 * np.pi * shift / 32) @ (P @ env.reshape(512, 3)),
                                   P @ np.roll(env, shift, axis=1).reshape(512, 3), atol=1e-10)


# ---------------------------------------------------------------------------------------------- configs[3]
def test_materialnet_on_the_gpu_matches_the_reference_golden_and_runs_at_1024(golden_dir):
    """f3 on PyTorch-ROCm (SDPA attention, hipBLASLt / MIOpen): fp32 on the GPU against the reference module's recorded fp64
    outputs (tests/golden/materialnet.npz, name-seeded weights), and a 1024x1024 input against the same module in fp64 on the host."""
    from materialist_amd.materialnet import MaterialNet, init_from_names, network_input_size

    dev = _cuda()
    g = np.load(os.path.join(golden_dir, "materialnet.npz"))
    net64 = init_from_names(MaterialNet().double().eval())
    net = init_from_names(MaterialNet().eval()).to(dev)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        out = net(x.float().to(dev))
    for k in ("depth", "albedo", "roughness", "metallic", "normal"):
        ref = g[k]
        err = np.abs(out[k].cpu().numpy() - ref).max() / (np.abs(ref).mean() + 1e-12)
        assert err < 2e-3, f"{k}: fp32 GPU vs reference fp64 golden: {err:.2e}"
    # BASELINE configs[3] shape: 1024x1024 image -> network input 518x518 (1369 tokens)
    nw, nh = network_input_size(1024, 1024)
    assert (nw, nh) == (518, 518)
    xs = torch.rand(1, 3, nh, nw, dtype=torch.float64)
    with torch.no_grad():
        ref = net64(xs)
        got = net(xs.float().to(dev))
    for k in ref:
        r_ = ref[k].numpy()
        err = np.abs(got[k].cpu().numpy() - r_).max() / (np.abs(r_).mean() + 1e-12)
        assert err < 5e-3, f"{k} at 518x518: {err:.2e}"
    img = (np.random.default_rng(0).random((1024, 1024, 3)) * 255).astype(np.uint8)
    maps = net.infer_image(img)
    assert maps["albedo"].shape == (1024, 1024, 3) and maps["depth"].shape == (1024, 1024) and np.isfinite(maps["normal"]).all()
    assert np.linalg.norm(maps["normal"], axis=-1).max() <= 1.0 + 1e-4                   # bilinear upsampling of unit normals


def test_materialnet_vit_large_variant_runs():
    """configs[3]'s DINOv2-L variant (random weights; the reference wires ViT-B only): 304 M-parameter encoder, same interface."""
    from materialist_amd.materialnet import MaterialNet

    dev = _cuda()
    net = MaterialNet(encoder="vitl").to(dev).eval()
    enc = sum(p.numel() for p in net.pretrained.parameters())
    assert 300e6 < enc < 310e6
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(torch.rand(1, 3, 518, 518, device=dev))
    assert out["albedo"].shape == (1, 3, 518, 518) and out["depth"].shape == (1, 1, 518, 518)
    assert all(torch.isfinite(v.float()).all() for v in out.values())


# ---------------------------------------------------------------------------------------------- configs[0]
def test_config0_256_none_arm_schedule_trace():
    """256x256, --model_name none --opt_order arm: the alternating schedule runs end to end on the fused loops and emits the
    phase / part / stop-reason sequence of inverse_img_w_mi.py:211-235,288-312,425-432 (three loops: env, brdf 'arm'; end)."""
    from materialist_amd import loss, optimize, render, synthetic

    dev = _cuda()
    H = W = 256
    spp = 64
    sc = synthetic.make_scene(0, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp).clone()
    mat = {"albedo": _t(sc.init_albedo, dev), "roughness": _t(sc.init_roughness, dev), "metallic": _t(sc.init_metallic, dev), "gt_image": gt}
    scene0 = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    with torch.no_grad():
        first = render.render_w_brdf(scene0, mat["albedo"], mat["roughness"], mat["metallic"], None, spp)
        psnr0 = float(loss.psnr(first * (gt.mean() / first.mean()), gt))
    out = optimize.optimize_envmap_ARMN(scene0, mat, optimize_order=("arm",), spp=spp, opt_env_from=0, opt_src="arm", num_epochs=400,
                                        sync_every=50, model_name="none")
    tr = out["trace"]
    assert [(t.loop, t.phase, t.part) for t in tr] == [(1, "env", ""), (1, "brdf", "arm"), (2, "env", ""), (2, "brdf", "arm"),
                                                       (3, "env", ""), (3, "end", "")]
    assert all(t.stop in ("num_epochs", "early_stop") for t in tr[:-1]) and tr[-1].stop in ("loop>=3", "early_stopping_all")
    assert all(0 <= t.epoch < 400 for t in tr[:-1])
    assert tr[0].lr == pytest.approx(1e-3 * 0.8 ** (tr[0].epoch // 100), rel=1e-6)      # env loop 1: StepLR(100, 0.8) from 1e-3
    assert tr[2].lr == pytest.approx(1e-4)                                              # later loops: 1e-4
    assert out["psnr"] > psnr0 + 3.0, (psnr0, out["psnr"])
    assert out["albedo"].shape == (H, W, 3) and out["envmap"].shape == (16, 32, 3)


# ------------------------------------------------------------------------- the estimator against its converged integral
@pytest.mark.parametrize("kind", ["scene", "stress"])
def test_hip_render_is_within_45_db_of_the_converged_integral(oracle64, kind):
    """The quadrature the kernels run (spp 64: 5 x 4 GGX half vectors + 3 x 6 cosine-weighted directions) against the
    reference-literal BSDF-sampling estimator at spp 4096 (fp64 oracle), on a 48x48 window of synthetic scene 0 at 512x512 and on
    the r = 0.1, m = 1 stress case of the same window; the literal estimator at spp 64 is measured beside it."""
    import math

    from materialist_amd import ops, synthetic

    dev = _cuda()
    H = W = 512
    i0, j0, w = 232, 232, 48
    sc = synthetic.make_scene(0, H, W)
    n = ops.normals_from_depth(_t(sc.depth, dev))
    a, r, m = sc.albedo.copy(), sc.roughness.copy(), sc.metallic.copy()
    if kind == "stress":
        r[:], m[:] = 0.1, 1.0
    out = ops.shade_fwd(_t(a, dev), _t(r, dev), _t(m, dev), n, _t(sc.light, dev), 64)
    sl = (slice(i0, i0 + w), slice(j0, j0 + w))
    n64 = n.cpu().numpy().astype(np.float64)
    args = (a[sl], r[sl], m[sl], n64[sl], sc.light)
    ref = oracle64.shade_fwd_win(*args, 4096, H, W, i0, j0, kind=1)
    lit = oracle64.shade_fwd_win(*args, 64, H, W, i0, j0, kind=1)
    g = lambda v: np.clip(v, 0, None) ** (1 / 2.2)
    psnr = lambda x: -10 * math.log10(np.mean((g(x) - g(ref)) ** 2))
    got, literal = psnr(out[sl].cpu().numpy().astype(np.float64)), psnr(lit)
    print(f"{kind}: HIP spp 64 {got:.1f} dB, reference-literal estimator spp 64 {literal:.1f} dB (vs spp 4096)")
    assert got >= 45.0 and got >= literal - 1.0


# ---------------------------------------------------------------------------------------------- configs[3], the HIP optimisation path
def test_hip_path_at_1024_render_batch_fused_iterations_and_one_pos_mlp_step(oracle64):
    """configs[3]'s resolution through the kernels of libmatpbr.so (the reference hard-codes 512 in six places): a 1024x1024
    render against the fp64 oracle on a 64x64 window with the full image's view directions; a batch of two equals the stand-alone
    renders bit for bit; two FusedBrdfPhase iterations (exact sampling and the lazy path) keep their 512x512 properties; one
    ArmMlpPhase iteration runs its M = 1 048 576-row kernels and improves on nothing worse than its first loss."""
    from materialist_amd import loop, ops, posmlp, render, synthetic
    from materialist_amd.armhead import ArmMlpPhase

    dev = _cuda()
    H = W = 1024
    spp = 64
    scs = [synthetic.make_scene(20 + i, H, W) for i in range(2)]
    st = lambda k: _t(np.stack([getattr(s, k) for s in scs]), dev)
    n = ops.normals_from_depth(st("depth"))
    a, r, m, light = st("albedo"), st("roughness"), st("metallic"), st("light")
    out = ops.shade_fwd(a, r, m, n, light, spp)
    for b in range(2):
        assert torch.equal(out[b], ops.shade_fwd(a[b], r[b], m[b], n[b], light[b], spp)), b
    i0, j0, w = 700, 131, 64
    sl = (slice(i0, i0 + w), slice(j0, j0 + w))
    n1 = n[1].cpu().numpy().astype(np.float64)
    ref = oracle64.shade_fwd_win(scs[1].albedo[sl], scs[1].roughness[sl], scs[1].metallic[sl], n1[sl], scs[1].light, spp, H, W, i0, j0)
    _close(out[1][sl], ref, 1e-3, "1024x1024 render vs oracle on a 64x64 window")
    # hot loop B, --model_name none: exact sampling and the lazy path, batch == stand-alone, lazy within 1e-3 of exact
    scene_b = render.load_estimated_mesh(st("depth"), use_mesh_normal=True)
    scene_b._set("emitter.data", light)
    init = [st(k) for k in ("init_albedo", "init_roughness", "init_metallic")]
    scene_1 = render.load_estimated_mesh(st("depth")[1], use_mesh_normal=True)
    scene_1._set("emitter.data", light[1])
    for lazy in (False, True):
        fb = loop.FusedBrdfPhase(scene_b, out, *init, optimize_part="rm", spp=spp, lazy=lazy)
        f1 = loop.FusedBrdfPhase(scene_1, out[1], *[x[1] for x in init], optimize_part="rm", spp=spp, lazy=lazy)
        pr_before = fb.p["roughness"].clone()
        fb.run(2)
        f1.run(2)
        for k in ("roughness", "metallic"):
            assert torch.equal(fb.p[k][1], f1.p[k]), (lazy, k)
        assert float(fb.stats[1, ops.STAT_MSE]) == float(f1.stats[0, ops.STAT_MSE])
        assert fb.poll()["iters"].tolist() == [2, 2]
        assert float((fb.p["roughness"] - pr_before).abs().max()) <= 2 * 3e-4 * 1.001       # Adam's bound: lr per step
        if lazy:      # the second iteration's render came from the models: against the phase that walks every sample
            e1 = loop.FusedBrdfPhase(scene_1, out[1], *[x[1] for x in init], optimize_part="rm", spp=spp, lazy=False)
            e1.run(2)
            scale = torch.maximum(e1.pred.abs(), e1.pred.abs().mean())
            assert float(((f1.pred - e1.pred).abs() / scale).max()) < 1e-3
    # hot loop B, --model_name pos_mlp: one iteration with 1 048 576 rows through the image-size MLP kernels
    net = posmlp.brdf_net("arm").to(dev)
    i1 = [x[1] for x in init]
    start_arm = torch.cat([i1[0].reshape(-1, 3), i1[1].reshape(-1, 1), i1[2].reshape(-1, 1)], -1).clamp(0, 1)
    ph = loop.pos_mlp_brdf_phase(scene_1, out[1], net, start_arm, {"albedo": i1[0], "roughness": i1[1], "metallic": i1[2]}, optimize_part="rm", spp=spp)
    assert isinstance(ph, ArmMlpPhase) and ph.bufs[0].shape[0] == H * W
    ph.step()
    first = float(ph.stats[0, ops.STAT_MSE])
    # the zero-initialised output layer makes the first render the render of the start maps (mymodels/mlps.py:174-176, 232-234)
    r0 = ops.shade_fwd(i1[0].clamp(0, 1), (i1[1] * 0.93 + 0.07).clamp(0.07, 1), i1[2].clamp(0, 1), ph._n, ph._light, spp, dcache=ph.dcache)
    assert torch.allclose(ph.pred, r0, rtol=2e-5, atol=1e-6)
    for _ in range(4):
        ph.step()
    assert np.isfinite(first) and float(ph.stats[0, ops.STAT_BEST]) <= first
    assert all(torch.isfinite(v).all() for v in (ph.flat, ph.maps["roughness"], ph.pred))


def test_split_operand_and_exact_f32_loops_reach_the_same_loss_at_512():
    """20 iterations of the headline loop (512x512, part 'rm') with the 256-wide layers on the split-operand bf16-MFMA kernels and
    on the exact-f32 MFMA kernels (--mlp-products 0): the same loss to 1e-3 at every iteration."""
    from materialist_amd import loop, ops, posmlp, render, synthetic
    from materialist_amd.posmlp import _PosMlpHipFn

    dev = _cuda()
    H = W = 512
    spp = 64
    sc = synthetic.make_scene(0, H, W)
    scene = render.load_estimated_mesh(_t(sc.depth, dev), use_mesh_normal=True)
    scene._set("emitter.data", _t(sc.light, dev))
    with torch.no_grad():
        gt = render.render_w_brdf(scene, _t(sc.albedo, dev), _t(sc.roughness, dev), _t(sc.metallic, dev), None, spp)
    init = [_t(x, dev) for x in (sc.init_albedo, sc.init_roughness, sc.init_metallic)]
    start_arm = torch.cat([init[0].reshape(-1, 3), init[1].reshape(-1, 1), init[2].reshape(-1, 1)], -1).clamp(0, 1)
    curves = {}
    keep = _PosMlpHipFn.PRODUCTS
    try:
        for products in (6, 0):
            _PosMlpHipFn.PRODUCTS = products
            torch.manual_seed(7)
            net = posmlp.brdf_net("arm").to(dev)
            ph = loop.pos_mlp_brdf_phase(scene, gt, net, start_arm, {"albedo": init[0], "roughness": init[1], "metallic": init[2]}, optimize_part="rm", spp=spp)
            losses = []
            for _ in range(20):
                ph.step()
                losses.append(ph.stats[0, ops.STAT_LOSS].clone() if hasattr(ph, "stats") else ph.last["loss"].clone())
            curves[products] = torch.stack(losses).cpu().numpy()
    finally:
        _PosMlpHipFn.PRODUCTS = keep
    assert curves[6][-1] < curves[6][0]
    assert np.abs(curves[6] - curves[0]).max() <= 1e-3 * np.abs(curves[0]).max(), (curves[6], curves[0])


# ---------------------------------------------------------------------------------------------- configs[1] as BASELINE.json writes it
def test_config1_as_written_indoor1_pos_mlp_rm_a_opt_env_from_2(golden_dir, tmp_path):
    """BASELINE configs[1]: `examples/indoor1.png 512x512, --model_name=pos_mlp --opt_order='rm a' --opt_env_from=2`.  The photograph is
    427 x 423 RGBA: centre crop to 423 x 423, alpha dropped, bilinear to 512 x 512 (myutils/misc.py:10-34), sRGB -> linear (:643-646).  The
    reference ships no MaterialNet prediction and no result for this photograph (the every-other real-image test uses indoor2, which has
    both), so the run starts from the flat prior and the test checks the RUN: the schedule (loop 1: one-epoch env phase because
    opt_env_from = 2 > 1, part 'rm', 'a' skipped; loop 2: env phase, 'rm', 'a'), a loss that goes down, and the reference's output layout."""
    import json
    import warnings

    from PIL import Image

    from materialist_amd import pipeline
    from materialist_amd.imageio_exr import read_exr

    _cuda()
    path = os.path.join(golden_dir, "indoor1.npz")
    if not os.path.exists(path):
        pytest.skip("indoor1 fixture missing")
    rgba = np.load(path)["image_rgba_u8"]
    assert rgba.shape == (423, 427, 4)
    src = str(tmp_path / "indoor1.png")
    Image.fromarray(rgba, "RGBA").save(src)
    lines = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = pipeline.inverse_image(src, "indoor1", opt_src="arm", opt_order=["rm", "a"], opt_env_from=2, save_path=str(tmp_path), size=512, spp=64,
                                     num_epochs=60, sync_every=10, log=lines.append, frame_interval=1e9, model_name="pos_mlp")
    out = res["output_dir"]
    cfg = json.load(open(os.path.join(out, "config.json")))
    assert cfg["image_size"] == [512, 512]
    assert cfg["model_name"] == "pos_mlp" and cfg["opt_order"] == ["rm", "a"] and cfg["opt_env_from"] == 2 and cfg["spp"] == 64
    # the crop: 423 rows kept, columns 2..424 of 427 (centre), resized with align_corners=True -- the corners of the target are the crop's corners
    gt = read_exr(os.path.join(out, "gt_image.exr"))
    assert gt.shape == (512, 512, 3)
    srgb = lambda u: u ** 2.2                                          # the reference's srgb_to_linear (myutils/misc.py:163-166)
    crop = rgba[:, 2:425, :3].astype(np.float64) / 255.0
    for (i, j), (ci, cj) in (((0, 0), (0, 0)), ((511, 511), (422, 422)), ((0, 511), (0, 422))):
        assert np.allclose(gt[i, j], srgb(crop[ci, cj]), atol=2e-3), (i, j, gt[i, j], srgb(crop[ci, cj]))
    # the schedule of --opt_env_from 2 (inverse_img_w_mi.py:211-235,343-345)
    tr = res["trace"]
    shown = [(t.loop, t.phase, t.part, t.epoch, t.stop) for t in tr]
    assert [(t.loop, t.phase, t.part) for t in tr if t.phase != "end"][:6] == [(1, "env", ""), (1, "brdf", "rm"), (1, "brdf", "a"), (2, "env", ""), (2, "brdf", "rm"), (2, "brdf", "a")], shown
    assert tr[0].epoch == 0, shown                                   # loop 1 < opt_env_from: a single env epoch
    assert tr[2].stop == "skip 'a' in loop 1", shown
    assert [t.epoch for t in tr if t.phase == "brdf" and t.loop == 1 and t.part == "rm"] == [59], shown      # 60 epochs, no early stop from the flat prior
    assert np.isfinite(res["best_loss"]) and res["psnr"] > 15.0
    # the loss went down: the gamma-space MSE the single env epoch of loop 1 saw (flat prior, uniform light) against SaveBest's at the end
    import re

    first = [float(re.search(r"mse ([0-9.eE+-]+)", l).group(1)) for l in lines if isinstance(l, str) and "loop 1: env phase done" in l]
    assert first and res["best_loss"] < 0.6 * first[0], (first, res["best_loss"])
    for name in ("albedoPred.exr", "normalPred.exr", "roughnessPred.png", "metallicPred.png", "depthPred.exr", "gt_image.exr", "gt_image.png", "config.json",
                 "env.png", "final_envmap.hdr", "opt_env_img.png", "indoor1.ply"):
        assert os.path.exists(os.path.join(out, name)), name
    assert sorted(os.listdir(os.path.join(out, "best_results"))) == ["albedo.exr", "envmap.hdr", "metallic.exr", "normal.exr", "rendered_img.exr", "roughness.exr"]
    best = read_exr(os.path.join(out, "best_results", "rendered_img.exr"))
    assert best.shape == (512, 512, 3) and np.isfinite(best).all()


# ---------------------------------------------------------------------------------------------- configs[2], what N ranks do, on one GPU
def test_rccl_path_of_the_bench_at_world_size_one():
    """`bench.py` launched exactly as the driver launches its N-rank runs (`python -m torch.distributed.run --nproc-per-node ...`), with one
    rank and --force-dist: RCCL `init_process_group("nccl", device_id=...)`, the barrier and the all_gather of the timing protocol
    execute on the GPU, and `world_size` in the JSON line comes from `dist.get_world_size()`.  The C3 command itself: --mode fused
    --images-per-gpu 8, whose headline is image-iterations/s over all ranks."""
    import json
    import socket
    import subprocess
    import sys

    _cuda()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--force-dist", "--gpus", "1", "--mode", "fused", "--images-per-gpu", "8", "--no-extras", "--no-cpu-baseline",
           "--steps", "300", "--warmup", "10"]           # the default line's `fused_b8` protocol (a phase's first iterations re-sample the most)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)      # a child process: never an exec from a GPU-initialised one
    assert res.returncode == 0, res.stderr[-2000:]
    line = res.stdout.splitlines()[-1]                   # the driver parses the LAST stdout line: compact, contract keys only
    assert len(line) < 4096
    brief = json.loads(line)
    out = json.load(open(os.path.join(root, brief["detail"])))          # the full record of the same run
    assert brief["value"] == pytest.approx(out["value"], rel=1e-5) and brief["config"]["images_per_gpu"] == 8 and brief["world_size"] == 1
    assert out["world_size"] == 1 and out["n_gpus"] == 1 and out["collective_backend"] == "nccl"
    assert out["config"]["images_per_gpu"] == 8 and out["config"]["mode"] == "fused"
    assert out["value"] == pytest.approx(300 * 8 / (out["ms_per_step"] * 300 * 1e-3), rel=1e-6)     # image-iterations/s over all ranks
    assert len(out["ranks"]) == 1 and out["value"] > 1000
    assert out["cold_first_process_it_per_s"] > 0
    # the same workload is `modes.fused_b8` of the default line (one rank, no RCCL): the two agree up to the run-to-run spread of a box
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-relight", "--steps", "5", "--warmup", "2"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    brief = json.loads(res.stdout.splitlines()[-1])
    assert len(res.stdout.splitlines()[-1]) < 4096 and 0 < brief["roofline"]["frac"] < 1 and brief["modes_it_per_s"]["fused_b8"] > 0
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"} <= set(brief["roofline"])
    default = json.load(open(os.path.join(root, brief["detail"])))
    b8 = default["modes"]["fused_b8"]
    # structure and consistency only: throughput and roofline thresholds are the perf gate's business (tools/final_r06.sh), not a test's --
    # the boxes of the pool are not all alike
    assert b8["images_per_gpu"] == 8 and b8["it_per_s"] > 0 and b8["it_per_s"] == pytest.approx(8e3 / b8["ms_per_step"], rel=1e-3)
    assert 0 < default["roofline"]["frac"] < 1 and default["cold_first_process_it_per_s"] > 0


@pytest.mark.parametrize("model_name,epochs", [("none", 5000), ("pos_mlp", 150)])
def test_the_pipeline_is_reproducible(model_name, epochs, tmp_path):
    """The reference's sample photograph end to end (`tools/real_image.py`: --opt_src a --opt_order rm a --opt_env_from 2, spp 64), twice in
    ONE process and once in a FRESH process: `final_envmap.hdr` and every file of `best_results/` are the same bytes.  Round 3 ended at
    27.8-33.9 dB depending on the history of the process (the env phase's head and optimiser ran as framework kernels picked on first use);
    with hot loop A on the C ABI, fixed-order reductions everywhere and order-free fixed-point sums where atomics are used (the walk
    queue of the folded step), nothing in an inversion depends on timing or on what ran before -- in `pos_mlp` mode for a fixed torch seed
    (the networks' random initial weights are the one source of run-to-run variation the reference has too)."""
    import json
    import subprocess
    import sys

    _cuda()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tests", "golden", "indoor2.npz")):
        pytest.skip("sample photograph fixture missing")
    runs = []
    for k, repeat in enumerate((2, 1)):
        res = subprocess.run([sys.executable, os.path.join(root, "tools", "pipeline_hashes.py"), "--model_name", model_name, "--out", str(tmp_path / f"p{k}"),
                              "--num_epochs", str(epochs), "--repeat", str(repeat), "--seed", "11"], capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-2000:]
        runs += json.loads([l for l in res.stdout.splitlines() if l.startswith("[")][-1])
    assert len(runs) == 3 and len(runs[0]["hashes"]) >= 6

    def first_difference(x, y):
        for (sx, dx), (sy, dy) in zip(x, y):
            if (sx, dx) != (sy, dy):
                return f"first stage that differs: {sx!r} ({dx} against {dy})"
        return "stage digests agree" if len(x) == len(y) else f"{len(x)} against {len(y)} stages"

    for r in runs[1:]:
        assert r["hashes"] == runs[0]["hashes"], (runs[0]["psnr"], r["psnr"], first_difference(runs[0]["stage_digests"], r["stage_digests"]))
        assert r["log"][1:] == runs[0]["log"][1:]            # the same losses, iteration counts and stop reasons, line by line
    # across runs, processes AND boxes: the digests committed for this arithmetic (regenerate with tools/pipeline_hashes.py --write-golden after a
    # change that is meant to move the numbers; a difference found here names the first stage of the schedule at which a run went its own way)
    gpath = os.path.join(root, "tests", "golden", f"indoor2_digests_{model_name}.json")
    assert os.path.exists(gpath), "commit the stage digests: python tools/pipeline_hashes.py --model_name ... --write-golden " + gpath
    gold = json.load(open(gpath))
    assert gold["num_epochs"] == epochs and gold["seed"] == 11
    got = runs[0]["stage_digests"]
    assert [list(d) for d in got] == gold["stage_digests"], (first_difference(gold["stage_digests"], got), gold["psnr"], runs[0]["psnr"])
    assert runs[0]["hashes"] == gold["hashes"]


def test_run_batch_takes_predictions_and_runs_a_shard_of_photographs_as_one_batch(golden_dir, tmp_path):
    """run_batch.py on two photographs with MaterialNet predictions given as files (--pred_dir): no flat-prior warning, and in
    --model_name none mode the rank's shard is ONE batch in the kernels' batch dimension; every image gets the reference's output
    directory (best_results/*.exr, envmap.hdr, final_envmap.hdr, config.json, .ply)."""
    import json
    import subprocess
    import sys
    import warnings

    from PIL import Image

    from materialist_amd.imageio_exr import write_exr

    _cuda()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(golden_dir, "jinjya256.npz"))
    imgs, preds = tmp_path / "imgs", tmp_path / "preds"
    imgs.mkdir()
    for k, name in enumerate(("shrine_a", "shrine_b")):
        gt = z["gt_linear_f16"].astype(np.float32)
        if k:
            gt = gt[:, ::-1].copy()                        # a second, different photograph: the mirror image
        write_exr(str(imgs / f"{name}.exr"), gt)
        d = preds / name
        d.mkdir(parents=True)
        flip = (lambda x: x[:, ::-1].copy()) if k else (lambda x: x)
        write_exr(str(d / "albedoPred.exr"), flip(z["albedo_pred_f16"].astype(np.float32)))
        H, W = z["depth_pred_f32"].shape
        up = np.zeros((H, W, 3), np.float32)
        up[..., 2] = 1
        write_exr(str(d / "normalPred.exr"), up)
        Image.fromarray(flip(z["roughness_pred_u8"])).save(str(d / "roughnessPred.png"))
        Image.fromarray(flip(z["metallic_pred_u8"])).save(str(d / "metallicPred.png"))
        write_exr(str(d / "depthPred.exr"), flip(z["depth_pred_f32"]))
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(root, "run_batch.py"), "--images", str(imgs / "shrine_a.exr"), str(imgs / "shrine_b.exr"),
           "--pred_dir", str(preds), "--save_path", str(out), "--model_name", "none", "--opt_src", "a", "--opt_order", "rm", "a", "--opt_env_from", "2",
           "--size", "256", "--num_epochs", "60"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "FLAT prior" not in res.stderr
    rows = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert [os.path.basename(r["image"]) for r in rows] == ["shrine_a.exr", "shrine_b.exr"] and all(r["error"] is None for r in rows)
    assert rows[0]["iterations"] == rows[1]["iterations"]          # one batch: the shard's images share the enqueued iterations
    assert all(np.isfinite(r["psnr_db"]) and r["psnr_db"] > 15 for r in rows)
    for name in ("shrine_a", "shrine_b"):
        d = out / name
        for f in ("config.json", f"{name}.ply", "final_envmap.hdr", "albedoPred.exr", "gt_image.png", "best_results/albedo.exr", "best_results/roughness.exr",
                  "best_results/metallic.exr", "best_results/rendered_img.exr", "best_results/normal.exr", "best_results/envmap.hdr"):
            assert (d / f).exists(), (name, f)
