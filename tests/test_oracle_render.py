"""Image-level oracle: the deterministic render's backward pass is the exact derivative of the forward pass with the
sampling state frozen (finite differences, fp64), and the fp32/OpenMP build agrees with the fp64 build."""
import numpy as np
import pytest


def _scene(H=6, W=7, seed=11):
    from materialist_amd import synthetic

    sc = synthetic.make_scene(seed, 64, 64)
    sl = (slice(10, 10 + H), slice(20, 20 + W))
    rng = np.random.default_rng(seed)
    n = rng.normal(size=(H, W, 3)) * 0.3 + np.array([0, 0, 1.0])
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    n *= rng.uniform(0.5, 2.0, (H, W, 1))   # the render normalises the map
    f64 = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    return f64(sc.albedo[sl]), f64(sc.roughness[sl]), f64(sc.metallic[sl]), f64(n), f64(sc.light)


def test_frozen_forward_equals_forward(oracle64):
    a, r, m, n, light = _scene()
    for spp in (2, 8, 32):
        np.testing.assert_allclose(oracle64.shade_fwd_frozen(a, r, m, n, n, r, light, spp), oracle64.shade_fwd(a, r, m, n, light, spp),
                                   rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("spp", [2, 16])
def test_backward_is_derivative_of_frozen_forward(oracle64, spp):
    a, r, m, n, light = _scene()
    r = np.clip(r, 0.12, 0.95)   # keep r +- eps inside the valid range
    rng = np.random.default_rng(3)
    w = rng.normal(size=a.shape)
    d_a, d_r, d_m, d_n, d_l = oracle64.shade_bwd(a, r, m, n, light, w, spp)
    f = lambda a_=a, r_=r, m_=m, n_=n, l_=light: (oracle64.shade_fwd_frozen(a_, r_, m_, n_, n, r, l_, spp) * w).sum()
    eps = 1e-6

    def check(x, grad, name, make):
        idxs = [tuple(rng.integers(0, s) for s in x.shape) for _ in range(6)]
        for idx in idxs:
            xp, xm = x.copy(), x.copy()
            xp[idx] += eps
            xm[idx] -= eps
            fd = (f(**make(xp)) - f(**make(xm))) / (2 * eps)
            assert fd == pytest.approx(grad[idx], rel=2e-5, abs=1e-7), f"{name}{idx}"

    check(a, d_a, "d_a", lambda v: {"a_": v})
    check(r, d_r, "d_r", lambda v: {"r_": v})
    check(m, d_m, "d_m", lambda v: {"m_": v})
    check(n, d_n, "d_n", lambda v: {"n_": v})
    check(light, d_l, "d_light", lambda v: {"l_": v})


def test_render_is_linear_in_light_and_nonnegative(oracle64):
    a, r, m, n, light = _scene()
    o1 = oracle64.shade_fwd(a, r, m, n, light, 8)
    o2 = oracle64.shade_fwd(a, r, m, n, 2.5 * light, 8)
    np.testing.assert_allclose(o2, 2.5 * o1, rtol=1e-12)
    assert (o1 >= 0).all()


def test_white_furnace_diffuse(oracle64):
    """Constant white light, m = 0, rough surface facing the camera: the estimator returns a bounded albedo-like value
    (energy is not created): rgb <= albedo * (Disney retro-reflection <= 1.25) + specular <= 1."""
    H = W = 4
    a = np.full((H, W, 3), 0.6)
    r = np.full((H, W, 1), 1.0)
    m = np.zeros((H, W, 1))
    n = np.zeros((H, W, 3))
    n[..., 2] = 1
    light = np.zeros((25, 3))
    light[0] = np.sqrt(4 * np.pi)   # radiance 1 in every direction
    out = oracle64.shade_fwd(a, r, m, n, light, 64)
    assert 0.4 < out.mean() < 0.9


def test_f32_port_matches_f64(oracle64, oracle32):
    a, r, m, n, light = _scene(12, 12)
    o64 = oracle64.shade_fwd(a, r, m, n, light, 16)
    o32 = oracle32.shade_fwd(a, r, m, n, light, 16)
    assert np.abs(o32 - o64).max() <= 2e-3 * np.abs(o64).mean()


def test_normals_from_depth_plane(oracle64):
    # a fronto-parallel plane has normal +z (towards the camera at the origin looking down -z)
    n = oracle64.normals_from_depth(np.full((8, 8), 2.0))
    np.testing.assert_allclose(n[2:-2, 2:-2], np.broadcast_to([0, 0, 1.0], (4, 4, 3)), atol=1e-9)
    # depth grows to the right: surface z = -(2 + k x) has normal (k, 0, 1)/|.| -> leans to +x, still faces the camera
    jj = np.arange(32)[None, :] * np.ones((32, 1))
    n = oracle64.normals_from_depth(2.0 + 0.002 * jj)
    assert (n[8:-8, 8:-8, 0] > 0).all() and (n[..., 2] > 0).all()
    np.testing.assert_allclose(np.linalg.norm(n, axis=-1), 1.0, atol=1e-12)


def test_lazy_specification_stays_within_the_parity_bar_on_a_drifting_roughness(oracle64):
    """The specification of the lazy re-sampling path (oracle lazy_refresh_pixel / lazy_eval_pixel, mirrored by csrc/matpbr_lazy.hpp):
    a forced call is the exact render; while every pixel's roughness drifts by up to 3e-4 per step (Adam's bound at the reference's
    learning rate, inverse_img_w_mi.py:359) the lazy render stays within 1e-3 max(|exact|, mean|exact|) of walking every sample, the
    models are rebuilt for a small fraction of the pixels, and the jac it hands to the backward pass is the exact one at refresh."""
    from materialist_amd import synthetic

    H, W, spp, T = 24, 32, 64, 120
    sc = synthetic.make_scene(3, H, W)
    o = oracle64
    f64 = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    a, m, light = f64(sc.albedo), f64(sc.metallic), f64(sc.light)
    n = o.normals_from_depth(f64(sc.depth))
    rng = np.random.default_rng(0)
    n = n + 0.3 * rng.normal(size=n.shape)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    r = np.clip(f64(sc.roughness), 0.07, 1.0)
    state = np.zeros((H, W, o.lazy_nstate()))
    vel = rng.uniform(-1, 1, r.shape)
    worst, frac = 0.0, []
    for t in range(T):
        exact = o.shade_fwd(a, r, m, n, light, spp)
        floor = 0.5 * np.abs(exact).mean()
        lazy, jac, ref = o.lazy_fwd(a, r, m, n, light, state, spp, floor, force=(t == 0))
        err = np.abs(lazy - exact) / np.maximum(np.abs(exact), np.abs(exact).mean())
        if t == 0:
            assert err.max() < 1e-13 and ref.all()
            d_r = o.shade_bwd(a, r, m, n, light, np.ones_like(exact), spp, want_n=False, want_light=False)[1]
            np.testing.assert_allclose(jac[..., 6:9].sum(-1), d_r[..., 0], rtol=1e-10, atol=1e-12)
        else:
            frac.append(ref.mean())
        worst = max(worst, err.max())
        vel = 0.9 * vel + 0.1 * rng.normal(size=r.shape)
        r = np.clip(r + 3e-4 * np.tanh(3 * vel), 0.07, 1.0)
    assert worst < 1e-3, worst
    assert 0.0 < np.mean(frac) < 0.15, np.mean(frac)
    # every interval is positive, no wider than the radius, and the radius inside its bounds
    assert (state[..., 1] > 0).all() and (state[..., 2] > 0).all()
    assert (state[..., 1] <= state[..., 3] + 1e-15).all() and (state[..., 2] <= state[..., 3] + 1e-15).all()
    assert (state[..., 3] >= 2.5e-4).all() and (state[..., 3] <= 3e-2).all()
