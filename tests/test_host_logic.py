"""CPU tests of the host side: C-ABI exports, loss / schedule helpers against the reference goldens, SH maps,
RGBE codec, scene parameter face, error behaviour without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from materialist_amd import _lib

    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "matpbr.h")).read()                        # the shading hot path (b)
    mlp = open(os.path.join(ROOT, "include", "matpbr_mlp.h")).read()                        # row f2: the coordinate MLP's entry points (included by matpbr.h)
    experimental = open(os.path.join(ROOT, "include", "matpbr_experimental.h")).read()      # measurement / A-B entry points: exported, not the boundary
    hot = set(re.findall(r"\b(matpbr_\w+)\s*\(", header))
    net = set(re.findall(r"\b(matpbr_\w+)\s*\(", mlp))
    assert len(hot) <= 60 and not (hot & net) and '#include "matpbr_mlp.h"' in header, (len(hot), hot & net)
    core = hot | net
    extra = set(re.findall(r"^int (matpbr_\w+)\s*\(", experimental, re.M))
    assert extra == {"matpbr_brdf_phase_stages_timed", "matpbr_mlp_set_lds_dma"} and not (core & extra)
    declared = core | extra
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.matpbr_version() == 311
    assert b"spp" in lib.matpbr_strerror(-2)
    assert lib.matpbr_shade_bwd_workspace_bytes(512, 512, 1, 25) == 512 * 75 * 4
    assert lib.matpbr_brdf_loss_workspace_bytes(2) > 0
    # argument checks that return before anything touches a device: the deferred-fold entry point (at most 16 records, kinds it knows)
    assert lib.matpbr_mlp_reduce_jobs(None, 0, None) != 0
    jobs = (_lib.ReduceJob * 17)()
    import ctypes
    assert lib.matpbr_mlp_reduce_jobs(ctypes.cast(jobs, ctypes.c_void_p), 17, None) != 0
    jobs[0].kind = 7
    assert lib.matpbr_mlp_reduce_jobs(ctypes.cast(jobs, ctypes.c_void_p), 1, None) != 0
    for k in range(3):
        jobs[k].kind = -1                                   # MATPBR_REDUCE_NONE: nothing to fold, nothing launched
    assert lib.matpbr_mlp_reduce_jobs(ctypes.cast(jobs, ctypes.c_void_p), 3, None) == 0


def test_python_mirror_of_the_abi_struct_and_flags_matches_the_header(tmp_path):
    """`_lib.MatpbrBrdfPhase` (ctypes) and the flag constants of `ops.py` against include/matpbr.h itself: a C program that includes the
    header prints sizeof / offsetof / the macros (gcc; the header is plain C), the Python side must agree field for field."""
    import ctypes
    import json
    import subprocess

    from materialist_amd import _lib, ops

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [name for name, _ in _lib.MatpbrBrdfPhase._fields_]
    src = tmp_path / "abi.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "matpbr.h"', 'int main(void) {',
             '  printf("{\\"sizeof\\": %zu", sizeof(MatpbrBrdfPhase));']
    lines += [f'  printf(", \\"{f}\\": %zu", offsetof(MatpbrBrdfPhase, {f}));' for f in fields]
    nfields = [name for name, _ in _lib.MatpbrNormalStep._fields_]
    lines += ['  printf(", \\"n_sizeof\\": %zu", sizeof(MatpbrNormalStep));']
    lines += [f'  printf(", \\"n_{f}\\": %zu", offsetof(MatpbrNormalStep, {f}));' for f in nfields]
    rfields = [name for name, _ in _lib.ReduceJob._fields_]
    lines += ['  printf(", \\"r_sizeof\\": %zu", sizeof(MatpbrReduceJob));']
    lines += [f'  printf(", \\"r_{f}\\": %zu", offsetof(MatpbrReduceJob, {f}));' for f in rfields]
    macros = ["FLAG_CLAMP_PARAMS", "FLAG_ATTACHED_SAMPLING", "FLAG_LAZY_FORCE", "FLAG_JAC16", "FLAG_MODELS_READY", "FLAG_ROTATE_BEST",
              "FLAG_GENERIC_STEP", "FLAG_JAC32", "FLAG_SHARE_GPU"]
    lines += [f'  printf(", \\"{m}\\": %u", (unsigned)MATPBR_{m});' for m in macros]
    lines += ['  printf(", \\"PART_N\\": %u", (unsigned)MATPBR_PART_N);']
    lines += ['  printf(", \\"PART_A\\": %u, \\"PART_R\\": %u, \\"PART_M\\": %u, \\"STATS_STRIDE\\": %d}\\n", MATPBR_PART_A, MATPBR_PART_R, MATPBR_PART_M, MATPBR_STATS_STRIDE);',
              '  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)], check=True)
    c = json.loads(subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout)
    assert ctypes.sizeof(_lib.MatpbrBrdfPhase) == c["sizeof"]
    for f in fields:
        assert getattr(_lib.MatpbrBrdfPhase, f).offset == c[f], f
    assert ctypes.sizeof(_lib.MatpbrNormalStep) == c["n_sizeof"]
    for f in nfields:
        assert getattr(_lib.MatpbrNormalStep, f).offset == c["n_" + f], f
    assert ctypes.sizeof(_lib.ReduceJob) == c["r_sizeof"]
    for f in rfields:
        assert getattr(_lib.ReduceJob, f).offset == c["r_" + f], f
    for m in macros:
        assert getattr(ops, m) == c[m], m
    # one word carries flags AND part bits (MatpbrBrdfPhase.flags beside part_mask; the `flags` argument of the brdf_loss_* entry points):
    # every value a single bit, all of them distinct
    # (ATTACHED_SAMPLING = 2 = PART_A on purpose never meet: the first is a flag of matpbr_shade_bwd / MatpbrBrdfPhase.flags, whose part bits
    # travel in `part_mask`; the brdf_loss_* entry points OR the part bits with JAC16 only)
    for word in ([c[m] for m in macros if m != "FLAG_JAC16"], [c["PART_A"], c["PART_R"], c["PART_M"], c["PART_N"], c["FLAG_JAC16"]]):
        assert all(v and not (v & (v - 1)) for v in word) and len(set(word)) == len(word), word
    assert ops.PART_N == c["PART_N"] and ops.part_mask("armn") == c["PART_A"] | c["PART_R"] | c["PART_M"] | c["PART_N"]
    assert c["STATS_STRIDE"] == 16 and ops.STAT_GT_SUM == 15
    from materialist_amd.loop import FusedBrdfPhase
    assert FusedBrdfPhase.PARTS == {"a": c["PART_A"], "r": c["PART_R"], "m": c["PART_M"]}


def test_argument_validation_without_gpu():
    """Entry points validate before touching the device: errors come back as codes, nothing launches."""
    from materialist_amd import _lib

    lib = _lib.load()
    cam = _lib.MatpbrCamera(35.0)
    null = ctypes.c_void_p(None)
    one = ctypes.c_void_p(8)
    assert lib.matpbr_shade_fwd(null, null, null, null, null, 0, 25, null, 8, 8, 1, 8, ctypes.byref(cam), 0, null) == -1
    assert lib.matpbr_shade_fwd(one, one, one, one, one, 0, 25, one, 8, 8, 1, 7, ctypes.byref(cam), 0, null) == -2
    assert lib.matpbr_shade_fwd(one, one, one, one, one, 0, 9, one, 8, 8, 1, 8, ctypes.byref(cam), 0, null) == -1
    assert lib.matpbr_shade_bwd(one, one, one, one, one, 0, 25, one, null, null, null, null, one, null, 0, 8, 8, 1, 8, ctypes.byref(cam), 0,
                                null) == -4
    assert lib.matpbr_adam_step(one, one, one, one, 10, 1e-3, 0.9, 0.999, 1e-8, 0, null) == -1


def test_product_path_has_no_cpu_fallback():
    from materialist_amd import ops
    from materialist_amd._lib import MatpbrError

    a = torch.rand(4, 4, 3)
    with pytest.raises(MatpbrError):
        ops.shade_fwd(a, torch.rand(4, 4, 1), torch.rand(4, 4, 1), a, torch.zeros(25, 3), 8)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "materialist_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


# ---------------------------------------------------------------------------------------- a13
def test_early_stopping_and_save_best_match_reference(golden_dir):
    from materialist_amd.loop import EarlyStopping, SaveBest

    g = np.load(os.path.join(golden_dir, "misc.npz"))
    for name in ("env", "flat", "rise"):
        seq = g[f"seq_{name}"]
        for pat, delta in ((3, 0.01), (2, 0.025), (4, 0.001)):
            es = EarlyStopping(patience=pat, min_delta=delta)
            tr = []
            for v in seq:
                es(float(v))
                tr.append((es.counter, es.best_loss, float(es.early_stop)))
            np.testing.assert_allclose(np.array(tr), g[f"es_{name}_{pat}_{delta}"], rtol=0, atol=0)
        sb = SaveBest()
        keep = []
        for k, v in enumerate(seq):
            sb.update(float(v), torch.full((1,), float(k)), None, None, None, None, None)
            keep.append((sb.best_loss, float(sb.best_albedo[0])))
        np.testing.assert_allclose(np.array(keep), g[f"sb_{name}"], rtol=0, atol=0)


def test_device_save_best_semantics():
    from materialist_amd.loop import DeviceSaveBest

    sb = DeviceSaveBest()
    seq = [0.5, 0.6, 0.4, 0.4, 0.3]
    for k, v in enumerate(seq):
        sb.update(torch.tensor([v, 1.0 - v]), albedo=torch.full((2, 2, 2, 3), float(k)))
    assert sb.best_loss.tolist() == pytest.approx([0.3, 0.4])
    assert sb.best["albedo"][0, 0, 0, 0] == 4 and sb.best["albedo"][1, 0, 0, 0] == 1  # strict <: ties keep the earlier snapshot


# ---------------------------------------------------------------------------------------- a11
def test_gamma_matches_reference(golden_dir):
    from materialist_amd import loss

    g = np.load(os.path.join(golden_dir, "misc.npz"))
    x = torch.from_numpy(g["gamma_x"])
    np.testing.assert_allclose(loss.linear_to_srgb(x).numpy(), g["linear_to_srgb"], rtol=1e-12)
    np.testing.assert_allclose(loss.srgb_to_linear(x).numpy(), g["srgb_to_linear"], rtol=1e-12)


def test_losses_follow_the_cited_formulas():
    """inverse_img_w_mi.py:241-245 and :388-418 written out with plain torch ops on seeded arrays."""
    import torch.nn.functional as NF

    from materialist_amd import loss

    gen = torch.Generator().manual_seed(5)
    pred = torch.rand(16, 16, 3, generator=gen, dtype=torch.float64) + 0.05
    gt = torch.rand(16, 16, 3, generator=gen, dtype=torch.float64) + 0.05
    ps, gs = pred ** (1 / 2.2), gt ** (1 / 2.2)
    l, mse, l1 = loss.env_loss(pred, gt)
    assert float(mse) == pytest.approx(float(NF.mse_loss(ps, gs)), rel=1e-12)
    assert float(l) == pytest.approx(float(NF.mse_loss(ps, gs) + NF.l1_loss(ps, gs)), rel=1e-12)
    a = torch.rand(16, 16, 3, generator=gen, dtype=torch.float64)
    a0 = torch.rand(16, 16, 3, generator=gen, dtype=torch.float64)
    r = torch.rand(16, 16, 1, generator=gen, dtype=torch.float64)
    r0 = torch.rand(16, 16, 1, generator=gen, dtype=torch.float64)
    ratio = gt.mean() / pred.mean()
    ps = (pred * ratio) ** (1 / 2.2)
    mse_, l1_ = NF.mse_loss(ps, gs), NF.l1_loss(ps, gs)
    want = 3 * (l1_ / mse_) * mse_ + l1_ + 0.1 * (NF.l1_loss(a, a0) + NF.l1_loss(r, r0))
    got, gmse, gsrgb, gratio = loss.brdf_loss(pred, gt, {"albedo": a, "roughness": r}, {"albedo": a0, "roughness": r0}, 0.1)
    assert float(got) == pytest.approx(float(want), rel=1e-12)
    assert float(gratio) == pytest.approx(float(ratio), rel=1e-12)
    # a batch is the sum of independent per-image losses
    pb, gb = torch.stack([pred, pred * 0.5 + 0.1]), torch.stack([gt, gt * 0.9])
    lb, mb, _, _ = loss.brdf_loss(pb, gb, {"albedo": torch.stack([a, a])}, {"albedo": torch.stack([a0, a0])}, 0.1)
    l0, m0, _, _ = loss.brdf_loss(pb[0], gb[0], {"albedo": a}, {"albedo": a0}, 0.1)
    l1b, m1, _, _ = loss.brdf_loss(pb[1], gb[1], {"albedo": a}, {"albedo": a0}, 0.1)
    assert float(lb) == pytest.approx(float(l0 + l1b), rel=1e-12)
    assert mb.tolist() == pytest.approx([float(m0), float(m1)], rel=1e-12)
    assert float(loss.psnr(gt, gt)) > 150


# ---------------------------------------------------------------------------------------- a10
def test_sh_host_utilities(oracle64):
    from materialist_amd import sh

    rng = np.random.default_rng(0)
    w = rng.normal(size=(40, 3))
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    np.testing.assert_allclose(sh.sh_basis(w), oracle64.sh_basis_dir(w), atol=1e-13)
    np.testing.assert_allclose(sh.sh_norm(), oracle64.sh_K() * (sh.sh_norm() / oracle64.sh_K()), rtol=1e-12)
    # texel directions invert lookup_envmap (myutils/envmap_utils.py:29-36)
    d = sh.envmap_directions(16, 32)
    phi = np.arctan2(d[..., 0], -d[..., 2]) / (2 * np.pi)
    u = np.clip((phi * 32 + 32) % 32, 0, 31).astype(int)
    v = np.clip(np.arccos(d[..., 1]) / np.pi * 16, 0, 15).astype(int)
    assert (u == np.arange(32)[None, :]).all() and (v == np.arange(16)[:, None]).all()
    assert sh.envmap_solid_angles(16, 32).sum() == pytest.approx(4 * np.pi, rel=1e-12)
    # projection of a band-limited envmap is exact up to the midpoint-rule error of the texel grid
    coef = rng.normal(size=(25, 3)) * 0.2
    coef[0] = 3.0
    env = sh.sh_to_envmap_matrix(64, 128) @ coef
    back = sh.envmap_to_sh_matrix(64, 128) @ env
    np.testing.assert_allclose(back, coef, atol=5e-3)
    # y-rotation: rotating the light by a full texel column equals rolling the envmap (render_final.py:290-298)
    M = sh.rotate_y_matrix(2 * np.pi / 32)
    env16 = (sh.sh_to_envmap_matrix(16, 32) @ coef).reshape(16, 32, 3)
    rot16 = (sh.sh_to_envmap_matrix(16, 32) @ (M @ coef)).reshape(16, 32, 3)
    np.testing.assert_allclose(rot16, np.roll(env16, 1, axis=1), atol=1e-10)
    np.testing.assert_allclose(sh.rotate_y_matrix(0.7) @ sh.rotate_y_matrix(-0.7), np.eye(25), atol=1e-12)


def test_hdr_codec(golden_dir, tmp_path):
    from materialist_amd.imageio_hdr import read_hdr, write_hdr

    g = np.load(os.path.join(golden_dir, "envmaps.npz"))
    for key in ("env0", "indoor", "jinjya"):
        img = g[key]
        assert img.shape == (16, 32, 3)
        p = str(tmp_path / f"{key}.hdr")
        write_hdr(p, img)
        back = read_hdr(p)
        # RGBE holds 8 mantissa bits shared per pixel: values that came out of an .hdr re-encode exactly
        np.testing.assert_allclose(back, img, rtol=0, atol=0)
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 4, (5, 9, 3)).astype(np.float32)
    p = str(tmp_path / "x.hdr")
    write_hdr(p, x)
    assert np.abs(read_hdr(p) - x).max() <= x.max() / 128


def test_synthetic_scenes_are_seeded():
    from materialist_amd import synthetic

    a, b, c = synthetic.make_scene(3, 32, 32), synthetic.make_scene(3, 32, 32), synthetic.make_scene(4, 32, 32)
    assert np.array_equal(a.albedo, b.albedo) and np.array_equal(a.light, b.light)
    assert not np.array_equal(a.albedo, c.albedo)
    assert a.roughness.min() >= 0.07 - 1e-6 and a.roughness.max() <= 1.0 + 1e-6
    from materialist_amd import sh

    assert (sh.sh_to_envmap_matrix(32, 64) @ a.light.astype(np.float64)).min() >= 0.02 - 1e-6


# ---------------------------------------------------------------------------------------- a8 (host part)
def test_scene_parameter_face_on_cpu():
    from materialist_amd import render

    sc = render.Scene(8, 8, "cpu")
    params = render.traverse(sc)
    assert set(params) == set(render.Scene.KEYS)
    assert sc.a.shape == (8, 8, 3) and float(sc.a[0, 0, 0]) == 0.5          # MatDiffBSDF defaults (mi_plugin.py:1238-1241)
    env = torch.rand(16, 32, 3, dtype=torch.float32, requires_grad=True)
    params["emitter.data"] = env
    assert sc.light.shape == (25, 3) and sc.light.requires_grad             # texels -> SH25, differentiable
    params["emitter.data"] = torch.zeros(25, 3)
    assert sc.light.shape == (25, 3)
    with pytest.raises(KeyError):
        params["shape.bsdf.x"] = 1
    with pytest.raises(ValueError):
        params["emitter.data"] = torch.zeros(7)
    params["shape.bsdf.use_mesh_normal"] = False
    assert sc.shading_normal() is sc.n
    params["shape.bsdf.use_mesh_normal"] = True
    assert sc.shading_normal() is sc.geo_normal
    a = torch.rand(8, 8, 3)
    params["shape.bsdf.a"] = a
    assert sc.a is a                                                         # stateful like mi.traverse (:73-78)
    with pytest.raises(Exception):
        sc.render(8)                                                         # CPU tensors: loud failure, no fallback


def test_shard_range_partitions_exactly():
    from materialist_amd.dist import shard_range

    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(64, 8, 3) == (24, 32)   # BASELINE config 3: 8 images per GPU


# ---------------------------------------------------------------------------------------- f1: pipeline head (host part)
def test_center_crop_and_resize_matches_reference(golden_dir):
    from materialist_amd.pipeline import center_crop_and_resize

    g = np.load(os.path.join(golden_dir, "misc_resize.npz"))
    np.testing.assert_allclose(center_crop_and_resize(g["im_u8"], (16, 16)), g["out_u8"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(center_crop_and_resize(g["im_f32"], (24, 24)), g["out_f32"], rtol=0, atol=1e-7)
    with pytest.raises(ValueError):
        center_crop_and_resize(np.zeros((4, 4, 3), np.float64), (2, 2))


def test_output_dir_rules_and_writers(tmp_path):
    from materialist_amd import pipeline
    from materialist_amd.imageio_exr import read_exr, write_exr

    # inverse_img_w_mi.py:82-104
    assert pipeline.get_output_dir("x") == os.path.join(pipeline.OUT_DIR, "x")
    assert pipeline.get_output_dir("/abs/x") == "/abs/x"
    assert pipeline.get_output_dir("x", "sub") == os.path.join(pipeline.OUT_DIR, "sub", "x")
    assert pipeline.get_output_dir("x", "/abs") == "/abs/x"
    rng = np.random.default_rng(2)
    for comp in ("none", "zips", "zip"):
        for shape in ((19, 23, 3), (8, 8), (5, 7, 1)):
            x = rng.normal(size=shape).astype(np.float32)
            p = str(tmp_path / f"{comp}.exr")
            write_exr(p, x, comp)
            y = read_exr(p)
            assert np.array_equal(y if x.ndim == 3 else y[..., 0], x)
    best = {k: torch.rand(6, 6, c) for k, c in (("albedo", 3), ("roughness", 1), ("metallic", 1), ("rendered_img", 3))}
    best["envmap"] = torch.rand(16, 32, 3)
    pipeline.save_results(str(tmp_path / "best_results"), best, torch.rand(6, 6, 3))
    assert sorted(os.listdir(tmp_path / "best_results")) == ["albedo.exr", "envmap.hdr", "metallic.exr", "normal.exr", "rendered_img.exr",
                                                             "roughness.exr"]
    np.testing.assert_array_equal(read_exr(str(tmp_path / "best_results" / "roughness.exr"))[..., 0], best["roughness"][..., 0].numpy())
    pipeline.write_png(str(tmp_path / "a.png"), rng.random((4, 5, 3)), linear=True)
    pr = pipeline.flat_prior(rng.random((6, 6, 3)).astype(np.float32))
    assert pr["normal"][0, 0].tolist() == [0, 0, 1] and pr["depth"].shape == (6, 6)


def test_cli_parses_the_reference_flags():
    import importlib.util

    spec = importlib.util.spec_from_file_location("cli", os.path.join(ROOT, "inverse_img_w_mi.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    a = cli.parse_args(["--img_inverse_path", "x.png", "--save_name", "s", "--opt_src", "arm", "--opt_order", "rm", "a", "--opt_env_from", "2"])
    assert a.opt_order == ["rm", "a"] and a.opt_env_from == 2 and a.model_name == "pos_mlp" and not a.use_mask   # the reference always runs pos_mlp (:782)


def test_hsv_round_trip_and_material_edit_flags():
    """render_final.py:143-181: HSV edit of the albedo and constant roughness / metallic inside the mask; file-name flags."""
    import colorsys

    import torch

    from materialist_amd import relight

    rng = np.random.default_rng(0)
    rgb = rng.random((200, 3))
    rgb[:5] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0], [0, 0, 1]]
    hsv = relight.rgb_to_hsv(rgb)
    np.testing.assert_allclose(hsv, np.array([colorsys.rgb_to_hsv(*c) for c in rgb]), atol=1e-12)
    np.testing.assert_allclose(relight.hsv_to_rgb(hsv), rgb, atol=1e-12)
    mask = torch.zeros(4, 6, dtype=torch.bool)
    mask[1:3, 2:5] = True
    mat = {"albedo": torch.full((4, 6, 3), 0.25), "roughness": torch.full((4, 6, 1), 0.5), "metallic": torch.zeros(4, 6, 1), "mask": mask}
    mat["albedo"][..., 0] = 0.75
    flag = relight.apply_edit(mat, {"albedo": [0.5, 0.0, 0.0], "roughness": 0.2, "metallic": None})
    assert flag == "_a_0.5_r_0.2"
    assert torch.all(mat["roughness"][mask] == 0.2) and torch.all(mat["roughness"][~mask] == 0.5)
    np.testing.assert_allclose(mat["albedo"][1, 2].numpy(), [0.25, 0.75, 0.75], atol=1e-6)      # red shifted by half a turn = cyan
    np.testing.assert_allclose(mat["albedo"][0, 0].numpy(), [0.75, 0.25, 0.25], atol=1e-6)
    assert relight.apply_edit(mat, {"albedo": None, "roughness": None, "metallic": None}) == ""
    del mat["mask"]
    with pytest.raises(FileNotFoundError):
        relight.apply_edit(mat, {"metallic": 1.0})


def test_masked_mean_fill_is_the_reference_in_place_form():
    """--use_mask (inverse_img_w_mi.py:379-381): `x[mask] = x[mask].mean()` on a non-leaf tensor, values and autograd gradients."""
    import torch

    from materialist_amd.loop import masked_mean_fill

    torch.manual_seed(0)
    mask = torch.rand(9, 7) > 0.6
    p_ref = torch.rand(9, 7, 1, dtype=torch.float64, requires_grad=True)
    p_new = p_ref.detach().clone().requires_grad_(True)
    w = torch.rand(9, 7, 1, dtype=torch.float64)
    x = p_ref.clamp(0.07, 1)
    x[mask] = x[mask].mean()                       # the reference's statement
    (x * w).sum().backward()
    y = masked_mean_fill(p_new.clamp(0.07, 1), mask)
    (y * w).sum().backward()
    assert torch.allclose(x, y, atol=1e-14) and torch.allclose(p_ref.grad, p_new.grad, atol=1e-14)
    assert torch.equal(masked_mean_fill(p_new.detach(), torch.zeros(9, 7, dtype=torch.bool)), p_new.detach())


def test_masked_mean_fill_of_a_batch_is_its_images_alone():
    """A batch under --use_mask: one masked mean PER IMAGE (a batch's images are independent runs of the reference), an image without masked pixels
    unchanged."""
    import torch

    from materialist_amd.loop import masked_mean_fill

    torch.manual_seed(1)
    x = torch.rand(3, 9, 7, 1, dtype=torch.float64)
    mask = torch.rand(3, 9, 7) > 0.5
    mask[2] = False
    y = masked_mean_fill(x, mask)
    for b in range(3):
        assert torch.allclose(y[b], masked_mean_fill(x[b], mask[b]), atol=1e-15)
    assert torch.equal(y[2], x[2])
