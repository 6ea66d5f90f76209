"""Torch-tensor front end of the C ABI: argument validation + pointer/stream plumbing only.
PyTorch is used for device memory and streams; all arithmetic happens in libmatpbr.so."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import MatpbrCamera, MatpbrError

LIGHT_SH25 = 0
NSH = 25
MAX_SPP = 128
FLAG_CLAMP_PARAMS = 1
FLAG_ATTACHED_SAMPLING = 2
FLAG_LAZY_FORCE = 16
FLAG_JAC16 = 32
FLAG_MODELS_READY = 64
FLAG_ROTATE_BEST = 128
FLAG_GENERIC_STEP = 256
FLAG_JAC32 = 512
FLAG_SHARE_GPU = 2048
LAZY_NSTATE = 28
LAZY_PLANES = 30          # 32-bit planes of a lazy model (csrc/matpbr_lazy.hpp kLzPlanes)
STATS_STRIDE = 16
(STAT_RATIO, STAT_MSE, STAT_L1, STAT_SR, STAT_LA, STAT_LR, STAT_LM, STAT_LOSS, STAT_IMPROVED, STAT_BEST, STAT_ES_COUNTER, STAT_ES_BEST,
 STAT_ES_HAS, STAT_STOPPED, STAT_ITERS, STAT_GT_SUM) = range(16)
PART_A, PART_R, PART_M, PART_N = 2, 4, 8, 1024


class KernelTimer:
    """Optional HIP-event timing of the shading launches, on the stream they are enqueued on.
    `with KernelTimer() as t: ...` then `t.summary()` -> {"shade_fwd": (n, mean_ms), "shade_bwd": (n, mean_ms)}."""

    active = None

    def __init__(self):
        self.events = {"shade_fwd": [], "shade_bwd": [], "brdf_phase_step": [], "env_phase_step": [], "relight": []}

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for k, ev in self.events.items():
            if ev:
                ms = [a.elapsed_time(b) for a, b in ev]
                out[k] = (len(ms), sum(ms) / len(ms))
        return out


class _timed:
    def __init__(self, name):
        self.name, self.t = name, KernelTimer.active

    def __enter__(self):
        if self.t is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.t is not None:
            self.e1.record()
            self.t.events[self.name].append((self.e0, self.e1))


def _dev(t: torch.Tensor, name: str, shape_tail=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise MatpbrError(f"{name}: matpbr kernels run on the GPU only (tensor is on {t.device}); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if shape_tail is not None and tuple(t.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise ValueError(f"{name}: expected trailing shape {shape_tail}, got {tuple(t.shape)}")
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _bhw(a: torch.Tensor):
    if a.ndim == 3:
        return 1, a.shape[0], a.shape[1]
    if a.ndim == 4:
        return a.shape[0], a.shape[1], a.shape[2]
    raise ValueError(f"maps must be [H,W,C] or [B,H,W,C], got {tuple(a.shape)}")


def check_spp(spp: int) -> int:
    spp = int(spp)
    if spp < 2 or spp > MAX_SPP or spp % 2:
        raise ValueError(f"spp must be even and in [2, {MAX_SPP}], got {spp}")
    return spp


def workspace_for(a: torch.Tensor, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Device scratch of matpbr_shade_bwd_workspace_bytes() for maps shaped like `a` ([H,W,3] or [B,H,W,3]); reuses `workspace` if it fits."""
    B, H, W = _bhw(a)
    need = int(_lib.load().matpbr_shade_bwd_workspace_bytes(H, W, B, NSH))
    if workspace is not None and workspace.device == a.device and workspace.numel() * workspace.element_size() >= need:
        return workspace
    return torch.empty((need + 3) // 4, dtype=torch.float32, device=a.device)


def plane9(a: torch.Tensor) -> torch.Tensor:
    """Scratch for the per-pixel plane buffers (diffuse cache / jac) of maps shaped like `a`: [9, B, H, W]."""
    B, H, W = _bhw(a)
    return torch.empty((9, B, H, W), dtype=torch.float32, device=a.device)


def shade_fwd_cached(a, m, jac: torch.Tensor, s1: torch.Tensor, clamp_params: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The render for new albedo / metallic maps from the planes a previous `shade_fwd(..., jac=jac, s1=s1)` kept, valid while the
    roughness, the normals and the light of that call are unchanged: bit-identical to rendering again, no samples."""
    lib = _lib.load()
    a = _dev(a, "albedo", (3,))
    B, H, W = _bhw(a)
    m = _dev(m, "metallic").reshape(B, H, W, 1)
    if jac.numel() != 3 * a.numel() or s1.numel() != a.numel() or not (jac.is_contiguous() and s1.is_contiguous()):
        raise ValueError("shade_fwd_cached: jac / s1 must be the contiguous plane buffers of the kept render")
    if out is None:
        out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        code = lib.matpbr_shade_fwd_cached(_ptr(a), _ptr(m), _ptr(jac), _ptr(s1), _ptr(out), H, W, B, FLAG_CLAMP_PARAMS if clamp_params else 0,
                                           _stream(a))
    _lib.check(code, "matpbr_shade_fwd_cached")
    return out


def shade_fwd(a, r, m, n, light, spp: int, fov_x_deg: float = 35.0, clamp_params: bool = False,
              out: Optional[torch.Tensor] = None, dcache: Optional[torch.Tensor] = None, jac: Optional[torch.Tensor] = None,
              s1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Render.  `dcache` = diffuse_cache(n, light, spp) skips the diffuse-lobe samples (valid while n and light are unchanged);
    `jac` ([9,B,H,W], filled) lets shade_bwd_jac / brdf_loss_bwd_jac form the material gradients of this pass; `s1` ([3,B,H,W],
    filled, needs jac) keeps what `shade_fwd_cached` needs besides jac."""
    lib = _lib.load()
    a = _dev(a, "albedo", (3,))
    B, H, W = _bhw(a)
    r = _dev(r, "roughness").reshape(B, H, W, 1)
    m = _dev(m, "metallic").reshape(B, H, W, 1)
    n = _dev(n, "normal", (3,))
    light = _dev(light, "light", (NSH, 3))
    if n.numel() != a.numel() or r.numel() * 3 != a.numel() or light.numel() != B * NSH * 3:
        raise ValueError("shade_fwd: inconsistent map / light shapes")
    for t, k in ((dcache, "dcache"), (jac, "jac")):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != 3 * a.numel()):
            raise ValueError(f"shade_fwd: {k} must be a contiguous fp32 CUDA tensor of 9*B*H*W floats")
    if out is None:
        out = torch.empty_like(a)
    cam = MatpbrCamera(float(fov_x_deg))
    with torch.cuda.device(a.device), _timed("shade_fwd"):
        if s1 is not None:
            if jac is None or s1.numel() != a.numel() or not s1.is_contiguous():
                raise ValueError("shade_fwd: s1 needs jac and must be a contiguous fp32 CUDA tensor of 3*B*H*W floats")
            code = lib.matpbr_shade_fwd_keep(_ptr(a), _ptr(r), _ptr(m), _ptr(n), _ptr(light), LIGHT_SH25, NSH, _ptr(dcache), _ptr(out), _ptr(jac),
                                             _ptr(s1), H, W, B, check_spp(spp), ctypes.byref(cam), FLAG_CLAMP_PARAMS if clamp_params else 0,
                                             _stream(a))
        else:
            code = lib.matpbr_shade_fwd_ex(_ptr(a), _ptr(r), _ptr(m), _ptr(n), _ptr(light), LIGHT_SH25, NSH, _ptr(dcache), _ptr(out), _ptr(jac),
                                           H, W, B, check_spp(spp), ctypes.byref(cam), FLAG_CLAMP_PARAMS if clamp_params else 0, _stream(a))
    _lib.check(code, "matpbr_shade_fwd_ex")
    return out


def lazy_fold(a: torch.Tensor) -> torch.Tensor:
    """Room for a part's folded per-pixel models (`MatpbrBrdfPhase.lazy_fold`, include/matpbr.h), for maps shaped like `a` [B,]H,W,3."""
    B, H, W = _bhw(a)
    return torch.empty(int(_lib.load().matpbr_lazy_fold_bytes(H, W, B)), dtype=torch.uint8, device=a.device)


def lazy_state(a: torch.Tensor) -> torch.Tensor:
    """Storage of the per-pixel local models of `shade_fwd_lazy` for maps shaped like `a` (opaque bytes; zero-filled)."""
    B, H, W = _bhw(a)
    return torch.zeros(int(_lib.load().matpbr_lazy_state_bytes(H, W, B)), dtype=torch.uint8, device=a.device)


def shade_fwd_lazy(a, r, m, n, light, spp: int, dcache: torch.Tensor, state: torch.Tensor, out: Optional[torch.Tensor] = None,
                   jac16: Optional[torch.Tensor] = None, force: bool = False, clamp_params: bool = False, floor: Optional[float] = None,
                   stats: Optional[torch.Tensor] = None, sums: Optional[torch.Tensor] = None, tol: float = 1.0, fov_x_deg: float = 35.0,
                   jac32: bool = False):
    """The render from per-pixel local models in the roughness (include/matpbr.h `matpbr_shade_fwd_lazy`): pixels whose roughness is
    still inside the validity interval of their model are a streaming evaluation, the others are re-sampled and their models rebuilt
    in `state` (from `lazy_state`).  `force=True` on the first call (and whenever light / normals / dcache changed).  Returns
    (out, jac16); jac16 feeds `brdf_loss_bwd_jac(..., jac16=True)` or `jac16_unpack`."""
    lib = _lib.load()
    a = _dev(a, "albedo", (3,))
    B, H, W = _bhw(a)
    r = _dev(r, "roughness").reshape(B, H, W, 1)
    m = _dev(m, "metallic").reshape(B, H, W, 1)
    n = _dev(n, "normal", (3,))
    light = _dev(light, "light", (NSH, 3))
    if n.numel() != a.numel() or r.numel() * 3 != a.numel() or light.numel() != B * NSH * 3:
        raise ValueError("shade_fwd_lazy: inconsistent map / light shapes")
    if dcache is None or not dcache.is_cuda or dcache.dtype != torch.float32 or not dcache.is_contiguous() or dcache.numel() != 3 * a.numel():
        raise ValueError("shade_fwd_lazy: dcache must be diffuse_cache(n, light, spp)")
    if not state.is_cuda or state.dtype != torch.uint8 or state.numel() < int(lib.matpbr_lazy_state_bytes(H, W, B)):
        raise ValueError("shade_fwd_lazy: state must come from lazy_state()")
    if stats is None and not (floor is not None and floor > 0):
        raise ValueError("shade_fwd_lazy: give the mean-radiance floor of the parity scale (or the statistics buffer)")
    if out is None:
        out = torch.empty_like(a)
    if jac16 is None:      # jac32: the nine fp32 planes of `shade_fwd(jac=...)` instead (what `shade_bwd_jac` reads)
        jac16 = torch.empty((9, B, H, W), dtype=torch.float32, device=a.device) if jac32 else torch.empty((5, B, H, W), dtype=torch.int32, device=a.device)
    if sums is not None and sums.numel() < B * int(lib.matpbr_lazy_sums_count(H, W)):
        raise ValueError("shade_fwd_lazy: sums too small")
    cam = MatpbrCamera(float(fov_x_deg))
    flags = (FLAG_CLAMP_PARAMS if clamp_params else 0) | (FLAG_LAZY_FORCE if force else 0) | (FLAG_JAC32 if jac32 else 0)
    with torch.cuda.device(a.device), _timed("shade_fwd"):
        code = lib.matpbr_shade_fwd_lazy(_ptr(a), _ptr(r), _ptr(m), _ptr(n), _ptr(light), LIGHT_SH25, NSH, _ptr(dcache), _ptr(state), _ptr(out),
                                         _ptr(jac16), _ptr(stats), _ptr(sums), H, W, B, check_spp(spp), ctypes.byref(cam), flags,
                                         float(floor or 0.0), float(tol), _stream(a))
    _lib.check(code, "matpbr_shade_fwd_lazy")
    return out, jac16


def lazy_state_unpack(state: torch.Tensor, a: torch.Tensor):
    """(models [B,H,W,28]: r_ref, lo, hi, rho, SD, S1, gSD, gS1, dSD, dS1, eSD, eS1 (rgb each), refreshed [B,H,W] int32: the pixels the last `shade_fwd_lazy` re-sampled)."""
    B, H, W = _bhw(a)
    st = torch.empty((B, H, W, LAZY_NSTATE), dtype=torch.float32, device=a.device)
    ref = torch.empty((B, H, W), dtype=torch.int32, device=a.device)
    with torch.cuda.device(a.device):
        code = _lib.load().matpbr_lazy_state_unpack(_ptr(state), _ptr(st), _ptr(ref), H, W, B, _stream(a))
    _lib.check(code, "matpbr_lazy_state_unpack")
    return st, ref


def jac16_unpack(jac16: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    B, H, W = _bhw(a)
    jac = plane9(a)
    with torch.cuda.device(a.device):
        code = _lib.load().matpbr_jac16_unpack(_ptr(jac16), _ptr(jac), H, W, B, _stream(a))
    _lib.check(code, "matpbr_jac16_unpack")
    return jac


# ---- pixels without geometry (mesh_mask.png, inverse_img_w_mi.py:713-724): their camera ray sees the environment, out = sum_k light[k] Y_k(ray).
# In the terms the fused loops work with (out = a (1-m) P + C0 (S0-S1) + S1, material gradients from P, S0-S1 and d out/d r) that is P = 0,
# S0-S1 = 0, S1 = background, d out/d r = 0: such a pixel renders the background whatever its materials are and hands them no gradient.  The
# helpers below write exactly that into the per-pixel buffers of a single image after the kernels have filled them.
def _half_bits(x: float) -> int:
    import numpy as np

    return int(np.array([x], dtype=np.float16).view(np.uint16)[0])


def background_into_lazy_state(state: torch.Tensor, a: torch.Tensor, bg_mask: torch.Tensor, bg_rgb: torch.Tensor, r: torch.Tensor) -> None:
    """Constant models for the masked pixels ([(B,)H,W] mask): P = SD = 0, S1 = bg_rgb, no slopes, an interval no roughness can leave."""
    B, H, W = _bhw(a)
    P = B * H * W
    planes = state[: LAZY_PLANES * 4 * P].view(torch.int32).view(LAZY_PLANES, P)
    idx = bg_mask.reshape(-1).nonzero().reshape(-1)
    bits = lambda t: t.contiguous().view(torch.int32)
    planes[0, idx] = bits(r.reshape(-1)[idx].clamp(0.07, 1.0).float())
    big = _half_bits(60000.0)
    planes[1, idx] = (big << 16) | big
    planes[2, idx] = bits(torch.full((idx.numel(),), 0.03, device=a.device))
    planes[3:9, idx] = 0
    for c in range(3):
        planes[9 + c, idx] = bits(bg_rgb.reshape(-1, 3)[idx, c].float())
    planes[12:LAZY_PLANES, idx] = 0


def background_index(bg_mask: torch.Tensor) -> torch.Tensor:
    """Flat indices of the masked pixels -- computed ONCE per phase (`nonzero` synchronises with the host): the per-iteration patches below
    then use index_fill_ / index_copy_, which enqueue without a round trip."""
    return bg_mask.reshape(-1).nonzero().reshape(-1)


def background_into_jac(jac: torch.Tensor, s1: Optional[torch.Tensor], bg_mask: torch.Tensor, bg_rgb: torch.Tensor,
                        idx: Optional[torch.Tensor] = None, rows_t: Optional[torch.Tensor] = None) -> None:
    """The same for the fp32 jac planes ([9, 1, H, W]: P, SD, d out/d r) and, if kept, the S1 planes ([3, 1, H, W]).  `idx` =
    background_index(bg_mask) and `rows_t` = bg_rgb[idx].t() (3, n) when the caller patches every iteration."""
    if idx is None:
        idx = background_index(bg_mask)
    jac.view(9, -1).index_fill_(1, idx, 0.0)
    if s1 is not None:
        s1.view(3, -1).index_copy_(1, idx, rows_t if rows_t is not None else bg_rgb.reshape(-1, 3)[idx].t().contiguous())


def background_into_transfer(T: torch.Tensor, H: int, W: int, bg_basis: torch.Tensor) -> None:
    """Radiance transfer (per image tiled [ceil(P/256)][75][256]; bg_basis [P,25] or [B,P,25]): a masked pixel's transfer is the SH
    basis along its camera ray, per channel."""
    P = H * W
    tiles = (P + 255) // 256
    basis = bg_basis.reshape(-1, P, 25)
    B = basis.shape[0]
    Tv = T[: B * tiles * 75 * 256].view(B, tiles, 25, 3, 256)
    Y = torch.zeros((B, tiles * 256, 25), dtype=torch.float32, device=T.device)
    Y[:, :P] = basis
    sel = (Y.abs().sum(2) > 0).view(B, tiles, 1, 1, 256)
    Yv = Y.view(B, tiles, 256, 25).permute(0, 1, 3, 2).unsqueeze(3)     # [B, tiles, 25, 1, 256]
    Tv.copy_(torch.where(sel, Yv.expand(-1, -1, -1, 3, -1), Tv))


def masked_mean_fill(x: torch.Tensor, mask_u8: torch.Tensor, out: Optional[torch.Tensor] = None, lo: float = 0.0, hi: float = 1.0,
                     gate: Optional[torch.Tensor] = None, batch: int = 1) -> torch.Tensor:
    """`--use_mask` on `batch` maps of x.numel() / batch entries each (include/matpbr.h `matpbr_masked_mean_fill`): masked entries become the masked mean of the clamped
    map; with `gate` (the forward's input) the backward form: masked gradients become their mean, through the clamp of the entry's own input.
    mask_u8: uint8, one byte per pixel."""
    lib = _lib.load()
    x = _dev(x, "map")
    if mask_u8.dtype != torch.uint8 or not mask_u8.is_cuda or not mask_u8.is_contiguous() or mask_u8.numel() != x.numel():
        raise ValueError("masked_mean_fill: mask must be a contiguous uint8 CUDA tensor with one byte per map entry")
    if gate is not None and (_dev(gate, "gate").numel() != x.numel()):
        raise ValueError("masked_mean_fill: gate must have the map's shape")
    B = int(batch)
    if out is None:
        out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        code = lib.matpbr_masked_mean_fill(_ptr(x), ctypes.c_void_p(mask_u8.data_ptr()), _ptr(gate), float(lo), float(hi), _ptr(out), x.numel() // B, B,
                                           _stream(x))
    _lib.check(code, "matpbr_masked_mean_fill")
    return out


def diffuse_cache(n, light, spp: int, fov_x_deg: float = 35.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-pixel coefficients of the diffuse lobe, a(1-m)(A0 + r A1 + r^2 A2): [9, B, H, W] planes, constants while the shading
    normals and the light stay fixed (a whole BRDF phase, inverse_img_w_mi.py:317-342)."""
    lib = _lib.load()
    n = _dev(n, "normal", (3,))
    B, H, W = _bhw(n)
    light = _dev(light, "light", (NSH, 3))
    if light.numel() != B * NSH * 3:
        raise ValueError("diffuse_cache: one [25,3] light per image expected")
    if out is None:
        out = plane9(n)
    cam = MatpbrCamera(float(fov_x_deg))
    with torch.cuda.device(n.device):
        code = lib.matpbr_diffuse_cache(_ptr(n), _ptr(light), LIGHT_SH25, NSH, _ptr(out), H, W, B, check_spp(spp), ctypes.byref(cam), _stream(n))
    _lib.check(code, "matpbr_diffuse_cache")
    return out


def shade_bwd_jac(a, r, m, jac, d_out):
    """Material gradients (d_a, d_r, d_m) of the forward pass that filled `jac` (same a, r, m): one streaming pass, no samples."""
    lib = _lib.load()
    a = _dev(a, "albedo", (3,))
    B, H, W = _bhw(a)
    r = _dev(r, "roughness").reshape(B, H, W, 1)
    m = _dev(m, "metallic").reshape(B, H, W, 1)
    d_out = _dev(d_out, "d_out", (3,))
    d_a, d_r, d_m = torch.empty_like(a), torch.empty_like(r), torch.empty_like(m)
    with torch.cuda.device(a.device), _timed("shade_bwd"):
        code = lib.matpbr_shade_bwd_jac(_ptr(a), _ptr(r), _ptr(m), _ptr(jac), _ptr(d_out), _ptr(d_a), _ptr(d_r), _ptr(d_m), H, W, B, _stream(a))
    _lib.check(code, "matpbr_shade_bwd_jac")
    return d_a, d_r, d_m


def shade_bwd(a, r, m, n, light, d_out, spp: int, fov_x_deg: float = 35.0, want_mat=True, want_n=False, want_light=False,
              workspace: Optional[torch.Tensor] = None, attached: bool = False):
    """Returns (d_a, d_r, d_m, d_n, d_light); entries not requested are None.  attached: d_r through the GGX quadrature nodes (the
    live reference's convention, mi_plugin.py:227-230,1335-1341) instead of the stop-gradient default."""
    lib = _lib.load()
    a = _dev(a, "albedo", (3,))
    B, H, W = _bhw(a)
    r = _dev(r, "roughness").reshape(B, H, W, 1)
    m = _dev(m, "metallic").reshape(B, H, W, 1)
    n = _dev(n, "normal", (3,))
    light = _dev(light, "light", (NSH, 3))
    d_out = _dev(d_out, "d_out", (3,))
    if d_out.numel() != a.numel():
        raise ValueError("shade_bwd: d_out shape mismatch")
    d_a = torch.empty_like(a) if want_mat else None
    d_r = torch.empty_like(r) if want_mat else None
    d_m = torch.empty_like(m) if want_mat else None
    d_n = torch.empty_like(n) if want_n else None
    d_l = torch.empty_like(light) if want_light else None
    ws_bytes = 0
    if want_light:
        workspace = workspace_for(a, workspace)
        ws_bytes = workspace.numel() * 4
    cam = MatpbrCamera(float(fov_x_deg))
    with torch.cuda.device(a.device), _timed("shade_bwd"):
        code = lib.matpbr_shade_bwd(_ptr(a), _ptr(r), _ptr(m), _ptr(n), _ptr(light), LIGHT_SH25, NSH, _ptr(d_out), _ptr(d_a),
                                    _ptr(d_r), _ptr(d_m), _ptr(d_n), _ptr(d_l), _ptr(workspace), ws_bytes, H, W, B,
                                    check_spp(spp), ctypes.byref(cam), FLAG_ATTACHED_SAMPLING if attached else 0, _stream(a))
    _lib.check(code, "matpbr_shade_bwd")
    return d_a, d_r, d_m, d_n, d_l


def eval_brdf(wi, wo, n, a, r, m):
    """MatDiffBSDF.eval_pdf over N lanes: returns (f*cos [N,3], pdf [N])."""
    lib = _lib.load()
    wi, wo, n, a = (_dev(t, k, (3,)) for t, k in ((wi, "wi"), (wo, "wo"), (n, "n"), (a, "a")))
    r, m = _dev(r, "r").reshape(-1), _dev(m, "m").reshape(-1)
    N = r.numel()
    f = torch.empty((N, 3), dtype=torch.float32, device=r.device)
    pdf = torch.empty(N, dtype=torch.float32, device=r.device)
    with torch.cuda.device(r.device):
        code = lib.matpbr_eval_brdf(_ptr(wi), _ptr(wo), _ptr(n), _ptr(a), _ptr(r), _ptr(m), _ptr(f), _ptr(pdf), N, _stream(r))
    _lib.check(code, "matpbr_eval_brdf")
    return f, pdf


def eval_brdf_bwd(wi, wo, n, a, r, m, g):
    lib = _lib.load()
    wi, wo, n, a, g = (_dev(t, k, (3,)) for t, k in ((wi, "wi"), (wo, "wo"), (n, "n"), (a, "a"), (g, "g")))
    r, m = _dev(r, "r").reshape(-1), _dev(m, "m").reshape(-1)
    N = r.numel()
    d_a = torch.empty((N, 3), dtype=torch.float32, device=r.device)
    d_n = torch.empty((N, 3), dtype=torch.float32, device=r.device)
    d_r = torch.empty(N, dtype=torch.float32, device=r.device)
    d_m = torch.empty(N, dtype=torch.float32, device=r.device)
    with torch.cuda.device(r.device):
        code = lib.matpbr_eval_brdf_bwd(_ptr(wi), _ptr(wo), _ptr(n), _ptr(a), _ptr(r), _ptr(m), _ptr(g), _ptr(d_a), _ptr(d_r),
                                        _ptr(d_m), _ptr(d_n), N, _stream(r))
    _lib.check(code, "matpbr_eval_brdf_bwd")
    return d_a, d_r, d_m, d_n


def sample_brdf(sample1, sample2, wo, n, a, r, m):
    """MatDiffBSDF.sample over N lanes: returns (wi [N,3], pdf [N], weight [N,3])."""
    lib = _lib.load()
    wo, n, a = (_dev(t, k, (3,)) for t, k in ((wo, "wo"), (n, "n"), (a, "a")))
    sample2 = _dev(sample2, "sample2", (2,))
    sample1, r, m = _dev(sample1, "sample1").reshape(-1), _dev(r, "r").reshape(-1), _dev(m, "m").reshape(-1)
    N = r.numel()
    wi = torch.empty((N, 3), dtype=torch.float32, device=r.device)
    w = torch.empty((N, 3), dtype=torch.float32, device=r.device)
    pdf = torch.empty(N, dtype=torch.float32, device=r.device)
    with torch.cuda.device(r.device):
        code = lib.matpbr_sample_brdf(_ptr(sample1), _ptr(sample2), _ptr(wo), _ptr(n), _ptr(a), _ptr(r), _ptr(m), _ptr(wi), _ptr(pdf),
                                      _ptr(w), N, _stream(r))
    _lib.check(code, "matpbr_sample_brdf")
    return wi, pdf, w


def sample_brdf_dr(sample1, sample2, wo, n, a, r, m):
    """d/dr of `sample_brdf` through the sampled direction and the pdf (the reference's attached convention, mi_plugin.py:227-230,
    1335-1341): returns (d_wi [N,3], d_pdf [N], d_weight [N,3])."""
    lib = _lib.load()
    wo, n, a = (_dev(t, k, (3,)) for t, k in ((wo, "wo"), (n, "n"), (a, "a")))
    sample2 = _dev(sample2, "sample2", (2,))
    sample1, r, m = _dev(sample1, "sample1").reshape(-1), _dev(r, "r").reshape(-1), _dev(m, "m").reshape(-1)
    N = r.numel()
    d_wi = torch.empty((N, 3), dtype=torch.float32, device=r.device)
    d_w = torch.empty((N, 3), dtype=torch.float32, device=r.device)
    d_pdf = torch.empty(N, dtype=torch.float32, device=r.device)
    with torch.cuda.device(r.device):
        code = lib.matpbr_sample_brdf_dr(_ptr(sample1), _ptr(sample2), _ptr(wo), _ptr(n), _ptr(a), _ptr(r), _ptr(m), _ptr(d_wi), _ptr(d_pdf),
                                         _ptr(d_w), N, _stream(r))
    _lib.check(code, "matpbr_sample_brdf_dr")
    return d_wi, d_pdf, d_w


def sh_eval(w, coef):
    lib = _lib.load()
    w = _dev(w, "w", (3,))
    coef = _dev(coef, "coef", (NSH, 3))
    N = w.numel() // 3
    L = torch.empty((N, 3), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        code = lib.matpbr_sh_eval(_ptr(w), _ptr(coef), _ptr(L), N, _stream(w))
    _lib.check(code, "matpbr_sh_eval")
    return L


def normals_from_depth(depth, fov_x_deg: float = 35.0):
    lib = _lib.load()
    depth = _dev(depth, "depth")
    if depth.ndim == 2:
        B, (H, W) = 1, depth.shape
    elif depth.ndim == 3:
        B, H, W = depth.shape
    else:
        raise ValueError("depth must be [H,W] or [B,H,W]")
    out = torch.empty(tuple(depth.shape) + (3,), dtype=torch.float32, device=depth.device)
    cam = MatpbrCamera(float(fov_x_deg))
    with torch.cuda.device(depth.device):
        code = lib.matpbr_normals_from_depth(_ptr(depth), _ptr(out), H, W, B, ctypes.byref(cam), _stream(depth))
    _lib.check(code, "matpbr_normals_from_depth")
    return out


def new_loss_stats(batch: int, device) -> torch.Tensor:
    """Per-image statistics buffer of matpbr_brdf_loss_stats; best_mse starts at +inf (SaveBest.best_loss, misc.py:64)."""
    st = torch.zeros((batch, STATS_STRIDE), dtype=torch.float32, device=device)
    st[:, STAT_BEST] = float("inf")
    return st


def part_mask(optimize_part: str) -> int:
    """MATPBR_PART_* bits of a part of --opt_order ('n': MATPBR_PART_N -- the material maps beside it are then meant literally, 'n' alone = none)."""
    return sum({"a": PART_A, "r": PART_R, "m": PART_M, "n": PART_N}.get(ch, 0) for ch in optimize_part)


def brdf_loss_stats(pred, gt, gt_srgb, pa, pr, pm, a0, r0, m0, scale_delta: float, stats: torch.Tensor,
                    workspace: Optional[torch.Tensor] = None, optimize_part: str = "arm", es_patience: int = -1, es_min_delta: float = 0.0,
                    history: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fills `stats` [B, STATS_STRIDE] in place (ratio, mse, l1, l1/mse, regulariser L1s, loss, improved, best_mse).  es_patience >= 0: the
    EarlyStopping state machine runs in `stats` on the device (include/matpbr.h `matpbr_brdf_loss_stats_es`), loss_mse goes to `history`."""
    lib = _lib.load()
    pred = _dev(pred, "pred", (3,))
    B, H, W = _bhw(pred)
    ts = [_dev(t, k) for t, k in ((gt, "gt"), (gt_srgb, "gt_srgb"), (pa, "pa"), (pr, "pr"), (pm, "pm"), (a0, "a0"), (r0, "r0"), (m0, "m0"))]
    need = int(lib.matpbr_brdf_loss_workspace_bytes(B))
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty(need // 4, dtype=torch.float32, device=pred.device)
    with torch.cuda.device(pred.device):
        code = lib.matpbr_brdf_loss_stats_es(_ptr(pred), *[_ptr(t) for t in ts], float(scale_delta), _ptr(stats), _ptr(workspace),
                                             workspace.numel() * 4, H, W, B, part_mask(optimize_part), int(es_patience), float(es_min_delta),
                                             _ptr(history), 0 if history is None else int(history.shape[0]), _stream(pred))
    _lib.check(code, "matpbr_brdf_loss_stats")
    return stats


def brdf_loss_dpred(pred, gt_srgb, stats, d_pred) -> torch.Tensor:
    """d loss / d pred of the BRDF-phase loss from the statistics `brdf_loss_stats` left (include/matpbr.h `matpbr_brdf_loss_dpred`), into `d_pred`."""
    lib = _lib.load()
    pred = _dev(pred, "pred", (3,))
    B, H, W = _bhw(pred)
    with torch.cuda.device(pred.device):
        code = lib.matpbr_brdf_loss_dpred(_ptr(pred), _ptr(_dev(gt_srgb, "gt_srgb", (3,))), _ptr(stats), _ptr(_dev(d_pred, "d_pred", (3,))), H, W, B,
                                          _stream(pred))
    _lib.check(code, "matpbr_brdf_loss_dpred")
    return d_pred


def brdf_loss_bwd_jac(pa, pr, pm, jac, pred, gt_srgb, stats, a0, r0, m0, scale_delta: float, d_a, d_r, d_m,
                      best_a=None, best_r=None, best_m=None, best_img=None, optimize_part: str = "arm", jac16: bool = False) -> None:
    """Backward of the fused BRDF-phase loss into preallocated d_a/d_r/d_m from the jac planes of the forward pass that rendered
    pa/pr/pm with clamp_params=True (see include/matpbr.h)."""
    lib = _lib.load()
    pa = _dev(pa, "pa", (3,))
    B, H, W = _bhw(pa)
    with torch.cuda.device(pa.device), _timed("shade_bwd"):
        code = lib.matpbr_brdf_loss_bwd_jac(_ptr(pa), _ptr(pr), _ptr(pm), _ptr(jac), _ptr(pred), _ptr(gt_srgb), _ptr(stats), _ptr(a0), _ptr(r0),
                                            _ptr(m0), float(scale_delta), _ptr(d_a), _ptr(d_r), _ptr(d_m), _ptr(best_a), _ptr(best_r),
                                            _ptr(best_m), _ptr(best_img), H, W, B, part_mask(optimize_part) | (FLAG_JAC16 if jac16 else 0), _stream(pa))
    _lib.check(code, "matpbr_brdf_loss_bwd_jac")


def adam_step(p, g, m, v, lr: float, step: int, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """In-place torch.optim.Adam update of `p` (state m, v) with gradient g; `step` is 1-based."""
    lib = _lib.load()
    with torch.cuda.device(p.device):
        code = lib.matpbr_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                    int(step), _stream(p))
    _lib.check(code, "matpbr_adam_step")


_colsum_ws = {}


def column_sum(x: torch.Tensor) -> torch.Tensor:
    """Sum over the rows of a contiguous [M, N] fp32 CUDA tensor -> [N] (bias gradients of the PosMLP layers)."""
    lib = _lib.load()
    x = _dev(x, "x")
    if x.ndim != 2:
        raise ValueError("column_sum expects [M, N]")
    M, N = x.shape
    key = (x.device, N)
    if key not in _colsum_ws:
        _colsum_ws[key] = torch.empty(int(lib.matpbr_column_sum_workspace_bytes(N)) // 4, dtype=torch.float32, device=x.device)
    ws = _colsum_ws[key]
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        code = lib.matpbr_column_sum(_ptr(x), _ptr(out), M, N, _ptr(ws), ws.numel() * 4, _stream(x))
    _lib.check(code, "matpbr_column_sum")
    return out


def shade_transfer(a, r, m, n, spp: int, fov_x_deg: float = 35.0) -> torch.Tensor:
    """Per-pixel radiance transfer of the materials (opaque tiled buffer, matpbr_transfer_bytes): `relight(T, lights, H, W)`
    reproduces shade_fwd for every light without re-evaluating the BRDF."""
    lib = _lib.load()
    a = _dev(a, "albedo", (3,))
    B, H, W = _bhw(a)
    r, m, n = _dev(r, "roughness").reshape(B, H, W, 1), _dev(m, "metallic").reshape(B, H, W, 1), _dev(n, "normal", (3,))
    T = torch.empty(int(lib.matpbr_transfer_bytes(H, W, B)) // 4, dtype=torch.float32, device=a.device)
    cam = MatpbrCamera(float(fov_x_deg))
    with torch.cuda.device(a.device):
        code = lib.matpbr_shade_transfer(_ptr(a), _ptr(r), _ptr(m), _ptr(n), _ptr(T), H, W, B, check_spp(spp), ctypes.byref(cam), 0, _stream(a))
    _lib.check(code, "matpbr_shade_transfer")
    return T


def relight(T, lights, H: int, W: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """T = shade_transfer(...) of ONE [H,W] image, lights [F,25,3] -> [F,H,W,3]."""
    lib = _lib.load()
    T = _dev(T, "T")
    lights = _dev(lights, "lights", (NSH, 3)).reshape(-1, NSH, 3)
    if T.numel() * 4 < int(lib.matpbr_transfer_bytes(H, W, 1)):
        raise ValueError("relight: T is smaller than matpbr_transfer_bytes(H, W, 1)")
    F_ = lights.shape[0]
    if out is None:
        out = torch.empty((F_, H, W, 3), dtype=torch.float32, device=T.device)
    with torch.cuda.device(T.device), _timed("relight"):
        code = lib.matpbr_relight(_ptr(T), _ptr(lights), _ptr(out), H, W, F_, _stream(T))
    _lib.check(code, "matpbr_relight")
    return out


def sin_bwd(d_y: torch.Tensor, pre: torch.Tensor) -> torch.Tensor:
    """d_y * cos(pre) for [M, n] fp32 CUDA tensors whose rows may be strided views (stride(1) == 1); contiguous result."""
    lib = _lib.load()
    if d_y.shape != pre.shape or d_y.ndim != 2 or d_y.stride(1) != 1 or pre.stride(1) != 1 or not d_y.is_cuda or d_y.dtype != torch.float32:
        raise ValueError("sin_bwd expects matching [M, n] fp32 CUDA tensors with unit column stride")
    M, n = d_y.shape
    out = torch.empty((M, n), dtype=torch.float32, device=d_y.device)
    with torch.cuda.device(d_y.device):
        code = lib.matpbr_sin_bwd(_ptr(d_y), d_y.stride(0), _ptr(pre), pre.stride(0), _ptr(out), M, n, _stream(d_y))
    _lib.check(code, "matpbr_sin_bwd")
    return out


def _mat2(t: torch.Tensor, name: str) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float32 and t.ndim == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0):
        raise ValueError(f"{name}: expected a [rows, cols] fp32 CUDA matrix with unit column stride, a row stride that is a multiple "
                         f"of 4 floats and a 16-byte aligned base (got shape {tuple(t.shape)}, strides {t.stride()})")
    return t


_mlp_ws: Dict[tuple, torch.Tensor] = {}


def _mlp_workspace(kind: str, M: int, device, nbytes: int) -> torch.Tensor:
    key = (kind, M, device)
    if key not in _mlp_ws:
        _mlp_ws[key] = torch.empty(max(int(nbytes) // 4, 1), dtype=torch.float32, device=device)
    return _mlp_ws[key]


def mlp_layer_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, s_out: torch.Tensor, c_out: Optional[torch.Tensor], K: int,
                  tail: Optional[torch.Tensor] = None, packed: bool = False) -> None:
    """s_out[:, :N] = sin(x[:, :K] w[:, :K]^T + bias), c_out[:, :N] = cos(same) (c_out None: no activation).  x [M, >=K], w [N, >=K];
    s_out / c_out are [M, >=N] with the same row stride.  One MFMA kernel (posmlp_kernels.hip).  packed: sines that carry the sign of
    their cosine in the last mantissa bit, no cosines (include/matpbr.h `matpbr_mlp_layer_fwd_sgn`; thin first layer at image size)."""
    lib = _lib.load()
    x, w, s_out = _mat2(x, "x"), _mat2(w, "w"), _mat2(s_out, "s_out")
    M, N = x.shape[0], w.shape[0]
    if packed:
        with torch.cuda.device(x.device):
            code = lib.matpbr_mlp_layer_fwd_sgn(_ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(bias.contiguous()), _ptr(s_out), s_out.stride(0),
                                                _ptr(tail) if tail is not None else None, tail.stride(0) if tail is not None else 0, M, N, K, _stream(x))
        _lib.check(code, "matpbr_mlp_layer_fwd_sgn")
        return
    if c_out is not None and (c_out.stride(0) != s_out.stride(0) or not c_out.is_cuda):
        raise ValueError("c_out must share s_out's row stride")
    with torch.cuda.device(x.device):                          # tail: x0 of a skip layer, stored into columns N.. with the outputs
        code = lib.matpbr_mlp_layer_fwd_tail(_ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(bias.contiguous()), _ptr(s_out),
                                             _ptr(c_out) if c_out is not None else None, s_out.stride(0),
                                             _ptr(tail) if tail is not None else None, tail.stride(0) if tail is not None else 0, M, N, K, _stream(x))
    _lib.check(code, "matpbr_mlp_layer_fwd")


def mlp_layer_bwd_input(g: torch.Tensor, wt: torch.Tensor, c_prev: torch.Tensor, g_prev: torch.Tensor, n_prev: int, n_red: int,
                        d_bias_prev: Optional[torch.Tensor], packed: bool = False) -> None:
    """g_prev[:, :n_prev] = (g[:, :n_red] wt[:n_prev, :n_red]^T) * c_prev[:, :n_prev]; d_bias_prev = column sums of g_prev.
    packed: `c_prev` holds the sign-carrying SINES of the layer below (the cosine is rebuilt in the epilogue)."""
    lib = _lib.load()
    g, wt = _mat2(g, "g"), _mat2(wt, "wt")
    M = g.shape[0]
    if c_prev.stride(0) != g_prev.stride(0):
        raise ValueError("c_prev and g_prev must share their row stride")
    ws = _mlp_workspace("bwd_input", M, g.device, lib.matpbr_mlp_bwd_input_workspace_bytes(M))
    with torch.cuda.device(g.device):
        code = (lib.matpbr_mlp_layer_bwd_input_sgn if packed else lib.matpbr_mlp_layer_bwd_input)(_ptr(g), g.stride(0), _ptr(wt), wt.stride(0), _ptr(c_prev), _ptr(g_prev), g_prev.stride(0),
                                              _ptr(d_bias_prev) if d_bias_prev is not None else None, _ptr(ws), ws.numel() * 4, M, n_prev,
                                              n_red, _stream(g))
    _lib.check(code, "matpbr_mlp_layer_bwd_input")


WSPLIT_TRANSPOSED, WSPLIT_F16X2 = 1, 2


def mlp_split_weights(w: torch.Tensor, N: int, K: int, out: Optional[torch.Tensor] = None, transposed: bool = False, f16: bool = False) -> torch.Tensor:
    """w[:N, :K] (row-major, unit column stride) split into three bf16 pieces in the operand order of the bx kernels (opaque bytes).
    transposed: the operand is w[:K, :N]^T (the forward weight serving the backward product).  f16: two f16 pieces of 256 w, the operand of
    the forward layers with nprod = 3 (include/matpbr.h `matpbr_mlp_split_weights_fmt`)."""
    lib = _lib.load()
    if not (w.is_cuda and w.dtype == torch.float32 and w.ndim == 2 and w.stride(1) == 1):
        raise ValueError("mlp_split_weights: expected a [rows, cols] fp32 CUDA matrix with unit column stride")
    need = int(lib.matpbr_mlp_wsplit_bytes(K))
    if out is None or out.numel() < need:
        out = torch.empty(need, dtype=torch.uint8, device=w.device)
    flags = (WSPLIT_TRANSPOSED if transposed else 0) | (WSPLIT_F16X2 if f16 else 0)
    with torch.cuda.device(w.device):
        code = lib.matpbr_mlp_split_weights_fmt(_ptr(w), w.stride(0), N, K, flags, _ptr(out), _stream(w))
    _lib.check(code, "matpbr_mlp_split_weights")
    return out


def mlp_layer_fwd_bx(x: torch.Tensor, wsplit: torch.Tensor, bias: torch.Tensor, s_out: torch.Tensor, c_out: Optional[torch.Tensor], N: int, K: int,
                     nprod: int = 6, tail: Optional[torch.Tensor] = None) -> None:
    """mlp_layer_fwd on the bf16 matrix pipe with split operands (nprod 6 or 9 partial products per f32 product; nprod 3: two f16 pieces,
    `wsplit` from mlp_split_weights(..., f16=True)).  c_out None: the sines carry the sign of their cosine in the last mantissa bit and no
    cosines are written."""
    lib = _lib.load()
    x, s_out = _mat2(x, "x"), _mat2(s_out, "s_out")
    if c_out is not None and _mat2(c_out, "c_out").stride(0) != s_out.stride(0):
        raise ValueError("c_out must share s_out's row stride")
    with torch.cuda.device(x.device):
        code = lib.matpbr_mlp_layer_fwd_bx_tail(_ptr(x), x.stride(0), _ptr(wsplit), _ptr(bias.contiguous()), _ptr(s_out),
                                                _ptr(c_out) if c_out is not None else None, s_out.stride(0),
                                                _ptr(tail) if tail is not None else None, tail.stride(0) if tail is not None else 0,
                                                x.shape[0], N, K, int(nprod), _stream(x))
    _lib.check(code, "matpbr_mlp_layer_fwd_bx")


def mlp_layer_fwd_bx_head(x: torch.Tensor, wsplit: torch.Tensor, bias: torch.Tensor, s_out: torch.Tensor, c_out: Optional[torch.Tensor], K: int, nprod: int,
                          w_out: torch.Tensor, bias_out: torch.Tensor, start: torch.Tensor, th: torch.Tensor, map_a: Optional[torch.Tensor],
                          map_r: Optional[torch.Tensor], map_m: Optional[torch.Tensor]) -> None:
    """mlp_layer_fwd_bx of the last sine layer (256 outputs) whose epilogue also forms the output layer and the 'arm' head for the
    rows it holds (= mlp_arm_head_fwd on s_out without reading s_out back)."""
    lib = _lib.load()
    x, s_out, w_out = _mat2(x, "x"), _mat2(s_out, "s_out"), _mat2(w_out, "w_out")
    P = lambda t: _ptr(t) if t is not None else None
    with torch.cuda.device(x.device):
        code = lib.matpbr_mlp_layer_fwd_bx_head(_ptr(x), x.stride(0), _ptr(wsplit), _ptr(bias.contiguous()), _ptr(s_out), P(c_out), s_out.stride(0),
                                                _ptr(w_out), w_out.stride(0), _ptr(bias_out), _ptr(start), start.stride(0), _ptr(th), P(map_a),
                                                P(map_r), P(map_m), x.shape[0], K, int(nprod), _stream(x))
    _lib.check(code, "matpbr_mlp_layer_fwd_bx_head")


def mlp_layer_bwd_input_bx(g: torch.Tensor, wtsplit: torch.Tensor, c_prev: torch.Tensor, g_prev: torch.Tensor, n_prev: int, n_red: int,
                           d_bias_prev: Optional[torch.Tensor], nprod: int = 6, packed: bool = False) -> None:
    """packed: `c_prev` holds the sign-carrying sines of the layer below."""
    lib = _lib.load()
    g = _mat2(g, "g")
    M = g.shape[0]
    if c_prev.stride(0) != g_prev.stride(0):
        raise ValueError("c_prev and g_prev must share their row stride")
    ws = _mlp_workspace("bwd_input", M, g.device, lib.matpbr_mlp_bwd_input_workspace_bytes(M))
    with torch.cuda.device(g.device):
        code = (lib.matpbr_mlp_layer_bwd_input_bx_sgn if packed else lib.matpbr_mlp_layer_bwd_input_bx)(_ptr(g), g.stride(0), _ptr(wtsplit), _ptr(c_prev), _ptr(g_prev), g_prev.stride(0),
                                                 _ptr(d_bias_prev) if d_bias_prev is not None else None, _ptr(ws), ws.numel() * 4, M, n_prev, n_red,
                                                 int(nprod), _stream(g))
    _lib.check(code, "matpbr_mlp_layer_bwd_input_bx")


def mlp_first_layer_bwd_bx(g: torch.Tensor, wtsplit: torch.Tensor, c_prev: torch.Tensor, x0: torch.Tensor, d_w0: torch.Tensor, d0: int, n0: int, n_red: int,
                           d_bias0: Optional[torch.Tensor], nprod: int = 6, packed: bool = False) -> None:
    """The backward pass into the first layer without its pre-activation gradient in memory (include/matpbr.h `matpbr_mlp_first_layer_bwd_bx`):
    d_w0 [n0, >= d0] (row n, column k) and d_bias0 [n0] from g = dL/d pre of the second layer.  x0 [M, >= 16] zero beyond d0."""
    lib = _lib.load()
    g, c_prev, x0 = _mat2(g, "g"), _mat2(c_prev, "c_prev"), _mat2(x0, "x0")
    M = g.shape[0]
    ws = _mlp_workspace("bwd_input", M, g.device, lib.matpbr_mlp_bwd_input_workspace_bytes(M))
    ws2 = _mlp_workspace("skinny2", 0, g.device, lib.matpbr_mlp_skinny_workspace_bytes(16))
    with torch.cuda.device(g.device):
        code = lib.matpbr_mlp_first_layer_bwd_bx(_ptr(g), g.stride(0), _ptr(wtsplit), _ptr(c_prev), c_prev.stride(0), 1 if packed else 0, _ptr(x0), x0.stride(0),
                                                 _ptr(d_w0), 1, d_w0.stride(0), int(d0), _ptr(d_bias0), _ptr(ws), ws.numel() * 4, _ptr(ws2), ws2.numel() * 4,
                                                 M, int(n0), int(n_red), int(nprod), _stream(g))
    _lib.check(code, "matpbr_mlp_first_layer_bwd_bx")


def mlp_layer_bwd_weight(g: torch.Tensor, x: torch.Tensor, N: int, K: int) -> torch.Tensor:
    """d_w [N, K] = g[:, :N]^T x[:, :K] over all rows (deterministic slab partials)."""
    lib = _lib.load()
    g, x = _mat2(g, "g"), _mat2(x, "x")
    M = g.shape[0]
    ws = _mlp_workspace("bwd_weight", M, g.device, lib.matpbr_mlp_bwd_weight_workspace_bytes(M))
    d_w = torch.empty((N, K), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        code = lib.matpbr_mlp_layer_bwd_weight(_ptr(g), g.stride(0), _ptr(x), x.stride(0), _ptr(d_w), K, _ptr(ws), ws.numel() * 4, M, N, K,
                                               _stream(g))
    _lib.check(code, "matpbr_mlp_layer_bwd_weight")
    return d_w


def mlp_layer_bwd_weight_bx(g: torch.Tensor, x: torch.Tensor, N: int, K: int, nprod: int = 6, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """As mlp_layer_bwd_weight with split operands (three bf16 pieces per f32, `nprod` bf16 MFMA products, f32 accumulate): both
    operands 256 columns wide in memory, rows a multiple of 16.  out: a [N, >= K] matrix to receive the gradient."""
    lib = _lib.load()
    g, x = _mat2(g, "g"), _mat2(x, "x")
    M = g.shape[0]
    ws = _mlp_workspace("bwd_weight", M, g.device, lib.matpbr_mlp_bwd_weight_workspace_bytes(M))
    d_w = out if out is not None else torch.empty((N, K), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        code = lib.matpbr_mlp_layer_bwd_weight_bx(_ptr(g), g.stride(0), _ptr(x), x.stride(0), _ptr(d_w), d_w.stride(0), _ptr(ws), ws.numel() * 4,
                                                  M, N, K, nprod, _stream(g))
    _lib.check(code, "matpbr_mlp_layer_bwd_weight_bx")
    return d_w


def mlp_skinny_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, K: int) -> None:
    """out[:, :J] = x[:, :K] w[:J, :K]^T + bias for the J in {3, 5, 8} outputs of the network at image size: one streaming pass."""
    lib = _lib.load()
    x, w = _mat2(x, "x"), _mat2(w, "w")
    with torch.cuda.device(x.device):
        code = lib.matpbr_mlp_skinny_fwd(_ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(out), out.stride(0), x.shape[0], w.shape[0], K,
                                         _stream(x))
    _lib.check(code, "matpbr_mlp_skinny_fwd")


def mlp_arm_head_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, start: torch.Tensor, th: torch.Tensor, map_a: Optional[torch.Tensor],
                     map_r: Optional[torch.Tensor], map_m: Optional[torch.Tensor], K: int) -> None:
    """Output layer (5 channels) + the 'arm' head (mymodels/mlps.py:233-236, inverse_img_w_mi.py:493-496): th [M,8] = tanh(x w^T + b),
    maps from the straight-through clamp of u = 1.3 th + start, (clamp(u, 0, 1) + u) - u; a map passed as None is not written."""
    lib = _lib.load()
    x, w = _mat2(x, "x"), _mat2(w, "w")
    with torch.cuda.device(x.device):
        code = lib.matpbr_mlp_arm_head_fwd(_ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(start), start.stride(0), _ptr(th),
                                           _ptr(map_a) if map_a is not None else None, _ptr(map_r) if map_r is not None else None,
                                           _ptr(map_m) if map_m is not None else None, x.shape[0], K, _stream(x))
    _lib.check(code, "matpbr_mlp_arm_head_fwd")


def mlp_arm_head_bwd(g_a: Optional[torch.Tensor], g_r: Optional[torch.Tensor], g_m: Optional[torch.Tensor], th: torch.Tensor, d_x: torch.Tensor) -> None:
    """d_x [M,8] = the map gradients chained through the 'arm' head; None = that map is not being optimised (zero columns)."""
    lib = _lib.load()
    with torch.cuda.device(th.device):
        code = lib.matpbr_mlp_arm_head_bwd(_ptr(g_a) if g_a is not None else None, _ptr(g_r) if g_r is not None else None,
                                           _ptr(g_m) if g_m is not None else None, _ptr(th), _ptr(d_x), th.shape[0], _stream(th))
    _lib.check(code, "matpbr_mlp_arm_head_bwd")


def mlp_skinny_bwd_weight(s: torch.Tensor, b: torch.Tensor, d_w: torch.Tensor, J: int, C: int, d_bias: Optional[torch.Tensor] = None,
                          transposed_out: bool = False) -> None:
    """d_w[j, c] (transposed_out: d_w[c, j]) = sum_m s[m, j] b[m, c] for j < J <= 16, c < C <= 256; d_bias[j] = sum_m s[m, j].
    s rows padded to a multiple of 8 floats, b 256 columns wide in memory."""
    lib = _lib.load()
    b = _mat2(b, "b")
    M = b.shape[0]
    ws = _mlp_workspace("skinny%d" % ((J + 7) // 8), 0, b.device, lib.matpbr_mlp_skinny_workspace_bytes(J))
    ld_j, ld_c = (1, d_w.stride(0)) if transposed_out else (d_w.stride(0), 1)
    with torch.cuda.device(b.device):
        code = lib.matpbr_mlp_skinny_bwd_weight(_ptr(s), s.stride(0), _ptr(b), b.stride(0), _ptr(d_w), ld_j, ld_c,
                                                _ptr(d_bias) if d_bias is not None else None, _ptr(ws), ws.numel() * 4, M, J, C, _stream(b))
    _lib.check(code, "matpbr_mlp_skinny_bwd_weight")


def mlp_out_layer_bwd(d_x: torch.Tensor, s_prev: torch.Tensor, c_prev: Optional[torch.Tensor], w_out: torch.Tensor, g_prev: torch.Tensor,
                      d_w: torch.Tensor, d_bias: Optional[torch.Tensor], d_bias_prev: Optional[torch.Tensor], J: int, n_prev: int) -> None:
    """The output layer's backward pass in one pass over the sines of the last sine layer (include/matpbr.h `matpbr_mlp_out_layer_bwd`):
    the layer's weight / bias gradient, the pre-activation gradient of the layer below and that layer's bias gradient.  c_prev None: s_prev
    holds the sign-carrying sines."""
    lib = _lib.load()
    s_prev, g_prev = _mat2(s_prev, "s_prev"), _mat2(g_prev, "g_prev")
    M = s_prev.shape[0]
    if c_prev is not None and _mat2(c_prev, "c_prev").stride(0) != s_prev.stride(0):
        raise ValueError("mlp_out_layer_bwd: c_prev and s_prev must share their row stride")
    ws = _mlp_workspace("skinny%d" % ((J + 7) // 8), 0, s_prev.device, lib.matpbr_mlp_skinny_workspace_bytes(J))
    with torch.cuda.device(s_prev.device):
        code = lib.matpbr_mlp_out_layer_bwd(_ptr(d_x), d_x.stride(0), _ptr(s_prev), _ptr(c_prev), s_prev.stride(0), _ptr(w_out), w_out.stride(0),
                                            _ptr(g_prev), g_prev.stride(0), _ptr(d_w), d_w.stride(0), 1, _ptr(d_bias), _ptr(d_bias_prev), _ptr(ws),
                                            ws.numel() * 4, M, int(J), int(n_prev), _stream(s_prev))
    _lib.check(code, "matpbr_mlp_out_layer_bwd")


# ---- the backward products on two f16 pieces under one exponent per 128-row tile (include/matpbr.h, "BACKWARD products ... two f16 pieces") ----
def mlp_tile_max(M: int, device) -> torch.Tensor:
    """A zeroed [M / 128] array of f32 bit patterns for the tile maxima of one gradient matrix (zero it again before every producer launch)."""
    if M % 128:
        raise ValueError("the block-scaled products take row counts that are multiples of 128")
    return torch.zeros(M // 128, dtype=torch.int32, device=device)


def mlp_out_layer_bwd_tmax(d_x: torch.Tensor, s_prev: torch.Tensor, w_out: torch.Tensor, g_prev: torch.Tensor, g_tile_max: torch.Tensor,
                           d_w: torch.Tensor, d_bias: Optional[torch.Tensor], d_bias_prev: Optional[torch.Tensor], J: int, n_prev: int, defer=None) -> None:
    """mlp_out_layer_bwd on sign-carrying sines that also fills `g_tile_max` (zeroed by the caller) for the g_prev it writes.
    defer: (slot of a `_lib.ReduceJob` array, workspace tag) -- the fold is left as a record for `mlp_reduce_jobs`, the partial sums in a workspace
    of that tag."""
    lib = _lib.load()
    s_prev, g_prev = _mat2(s_prev, "s_prev"), _mat2(g_prev, "g_prev")
    M = s_prev.shape[0]
    ws = _mlp_workspace("skinny%d%s" % ((J + 7) // 8, defer[1] if defer else ""), 0, s_prev.device, lib.matpbr_mlp_skinny_workspace_bytes(J))
    with torch.cuda.device(s_prev.device):
        code = lib.matpbr_mlp_out_layer_bwd_tmax(_ptr(d_x), d_x.stride(0), _ptr(s_prev), None, s_prev.stride(0), _ptr(w_out), w_out.stride(0),
                                                 _ptr(g_prev), g_prev.stride(0), _ptr(g_tile_max), _ptr(d_w), d_w.stride(0), 1, _ptr(d_bias),
                                                 _ptr(d_bias_prev), _ptr(ws), ws.numel() * 4, M, int(J), int(n_prev), defer[0] if defer else None, _stream(s_prev))
    _lib.check(code, "matpbr_mlp_out_layer_bwd_tmax")


def mlp_layer_bwd_input_blk(g: torch.Tensor, g_tile_max: torch.Tensor, wtsplit: torch.Tensor, s_prev: torch.Tensor, g_prev: torch.Tensor, n_prev: int,
                            n_red: int, d_bias_prev: Optional[torch.Tensor], out_tile_max: Optional[torch.Tensor], defer=None) -> None:
    """mlp_layer_bwd_input_bx(packed=True) on two f16 pieces: `wtsplit` from mlp_split_weights(..., transposed=True, f16=True).  defer: as
    mlp_out_layer_bwd_tmax."""
    lib = _lib.load()
    g = _mat2(g, "g")
    M = g.shape[0]
    if s_prev.stride(0) != g_prev.stride(0):
        raise ValueError("s_prev and g_prev must share their row stride")
    ws = _mlp_workspace("bwd_input" + (defer[1] if defer else ""), M, g.device, lib.matpbr_mlp_bwd_input_workspace_bytes(M))
    with torch.cuda.device(g.device):
        code = lib.matpbr_mlp_layer_bwd_input_blk(_ptr(g), g.stride(0), _ptr(g_tile_max), _ptr(wtsplit), _ptr(s_prev), _ptr(g_prev), g_prev.stride(0),
                                                  _ptr(out_tile_max) if out_tile_max is not None else None,
                                                  _ptr(d_bias_prev) if d_bias_prev is not None else None, _ptr(ws), ws.numel() * 4, M, n_prev, n_red,
                                                  defer[0] if defer else None, _stream(g))
    _lib.check(code, "matpbr_mlp_layer_bwd_input_blk")


def mlp_first_layer_bwd_blk(g: torch.Tensor, g_tile_max: torch.Tensor, wtsplit: torch.Tensor, s_prev: torch.Tensor, x0: torch.Tensor, d_w0: torch.Tensor,
                            d0: int, n0: int, n_red: int, d_bias0: Optional[torch.Tensor], defer=None) -> None:
    """mlp_first_layer_bwd_bx(packed=True) on two f16 pieces.  defer: as mlp_out_layer_bwd_tmax, the slot being the first of TWO records."""
    lib = _lib.load()
    g, s_prev, x0 = _mat2(g, "g"), _mat2(s_prev, "s_prev"), _mat2(x0, "x0")
    M = g.shape[0]
    ws = _mlp_workspace("bwd_input" + (defer[1] if defer else ""), M, g.device, lib.matpbr_mlp_bwd_input_workspace_bytes(M))
    ws2 = _mlp_workspace("skinny2" + (defer[1] if defer else ""), 0, g.device, lib.matpbr_mlp_skinny_workspace_bytes(16))
    with torch.cuda.device(g.device):
        code = lib.matpbr_mlp_first_layer_bwd_blk(_ptr(g), g.stride(0), _ptr(g_tile_max), _ptr(wtsplit), _ptr(s_prev), s_prev.stride(0), _ptr(x0), x0.stride(0),
                                                  _ptr(d_w0), 1, d_w0.stride(0), int(d0), _ptr(d_bias0), _ptr(ws), ws.numel() * 4, _ptr(ws2), ws2.numel() * 4,
                                                  M, int(n0), int(n_red), defer[0] if defer else None, _stream(g))
    _lib.check(code, "matpbr_mlp_first_layer_bwd_blk")


def mlp_layer_bwd_weight_blk(g: torch.Tensor, g_tile_max: torch.Tensor, x: torch.Tensor, N: int, K: int, out: Optional[torch.Tensor] = None,
                             defer=None) -> torch.Tensor:
    """mlp_layer_bwd_weight_bx on two f16 pieces (x as it is, g under one exponent per slab of rows).  defer: as mlp_out_layer_bwd_tmax (the 256
    slabs of partial sums stay in a workspace of their own: 67 MB per deferred layer at 512 x 512)."""
    lib = _lib.load()
    g, x = _mat2(g, "g"), _mat2(x, "x")
    M = g.shape[0]
    ws = _mlp_workspace("bwd_weight" + (defer[1] if defer else ""), M, g.device, lib.matpbr_mlp_bwd_weight_workspace_bytes(M))
    d_w = out if out is not None else torch.empty((N, K), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        code = lib.matpbr_mlp_layer_bwd_weight_blk(_ptr(g), g.stride(0), _ptr(g_tile_max), _ptr(x), x.stride(0), _ptr(d_w), d_w.stride(0), _ptr(ws),
                                                   ws.numel() * 4, M, N, K, defer[0] if defer else None, _stream(g))
    _lib.check(code, "matpbr_mlp_layer_bwd_weight_blk")
    return d_w


def mlp_reduce_jobs(jobs, n: int, like: torch.Tensor) -> None:
    """Every deferred fold of an iteration's backward pass in one launch (include/matpbr.h `matpbr_mlp_reduce_jobs`); jobs: a `_lib.ReduceJob` array."""
    with torch.cuda.device(like.device):
        code = _lib.load().matpbr_mlp_reduce_jobs(ctypes.byref(jobs), int(n), _stream(like))
    _lib.check(code, "matpbr_mlp_reduce_jobs")


LIGHT_SH9, LIGHT_ENV_TEXELS = 1, 2


class _LightToSh25(torch.autograd.Function):
    @staticmethod
    def forward(ctx, light, kind):
        lib = _lib.load()
        light = _dev(light, "light")
        B, n = light.shape[0], light.shape[1]
        out = torch.empty((B, NSH, 3), dtype=torch.float32, device=light.device)
        with torch.cuda.device(light.device):
            code = lib.matpbr_light_to_sh25(_ptr(light), int(kind), n, _ptr(out), B, _stream(light))
        _lib.check(code, "matpbr_light_to_sh25")
        ctx.kind, ctx.n = int(kind), n
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        d_out = d_out.contiguous()
        B = d_out.shape[0]
        d_light = torch.empty((B, ctx.n, 3), dtype=torch.float32, device=d_out.device)
        with torch.cuda.device(d_out.device):
            code = lib.matpbr_light_to_sh25_bwd(_ptr(d_out), ctx.kind, ctx.n, _ptr(d_light), B, _stream(d_out))
        _lib.check(code, "matpbr_light_to_sh25_bwd")
        return d_light, None


def light_to_sh25(light: torch.Tensor, kind: int) -> torch.Tensor:
    """[B, n, 3] light of kind LIGHT_SH9 (n = 9) or LIGHT_ENV_TEXELS (n = He * 2He equirectangular texels, the reference scene's
    `emitter.data`) -> the [B, 25, 3] SH coefficients the shading kernels take; differentiable."""
    squeeze = light.ndim == 2
    out = _LightToSh25.apply(light.reshape((1,) + tuple(light.shape)) if squeeze else light, kind)
    return out[0] if squeeze else out


def adamw_step_dev(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, hyper: torch.Tensor, weight_decay: float = 0.01,
                   beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.AdamW on one flat buffer; hyper = [lr, steps done] in device memory (the count is advanced by the call)."""
    lib = _lib.load()
    with torch.cuda.device(p.device):
        code = lib.matpbr_adamw_step_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(hyper), float(beta1), float(beta2), float(eps),
                                         float(weight_decay), _stream(p))
    _lib.check(code, "matpbr_adamw_step_dev")


def brdf_terms(cos1, cos2, r, f0) -> torch.Tensor:
    """[N,4] = D_GGX(cos1, r), G1_GGX_Schlick(cos1, r), G_Smith(cos1, cos2, r), fresnelSchlick(cos1, f0) over N lanes."""
    lib = _lib.load()
    cos1, cos2, r, f0 = (_dev(t, k).reshape(-1) for t, k in ((cos1, "cos1"), (cos2, "cos2"), (r, "r"), (f0, "f0")))
    N = cos1.numel()
    out = torch.empty((N, 4), dtype=torch.float32, device=cos1.device)
    with torch.cuda.device(cos1.device):
        code = lib.matpbr_brdf_terms(_ptr(cos1), _ptr(cos2), _ptr(r), _ptr(f0), _ptr(out), N, _stream(cos1))
    _lib.check(code, "matpbr_brdf_terms")
    return out
