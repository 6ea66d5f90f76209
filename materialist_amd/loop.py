"""Per-iteration bodies of the optimisation loop (hot loops A and B of SURVEY.md 3.1) on top of the HIP
render, plus SaveBest / EarlyStopping with the reference's semantics (myutils/misc.py:37-111).

`BrdfPhase.step()` is hot loop B in `--model_name none` mode (inverse_img_w_mi.py:347-468): clamp the
parameter maps, render, scale by mean(gt)/mean(pred), gamma-2.2 MSE+L1 with the L1/MSE re-weighting, L1
regularisers towards the initial maps, backward, Adam step.  `EnvPhase.step()` is hot loop A (:236-254)
with the light parameterised directly (SH coefficients or texels) instead of through the envmap MLP.
Everything stays on the device: the best-so-far snapshot is kept with torch.where, so no `.item()`
round trip is needed per iteration (the reference syncs three times per epoch, :247,250,255).
"""
from __future__ import annotations

import os
import copy
from typing import Dict, Optional

import torch

from . import loss as _loss
from . import render as _render


class EarlyStopping:
    """Patience on the relative improvement of a loss, with the behaviour of the reference's helper (myutils/misc.py:37-60, pinned by
    tests/golden/misc.npz): a call counts as a miss when the loss is above `(1 - min_delta)` times the loss of the last call that was
    not a miss; `patience` misses since that call raise `early_stop`, which then stays up; the first call only sets the level."""

    def __init__(self, patience: int = 10, min_delta: float = 0.0):
        self.patience, self.min_delta = patience, min_delta
        self._level, self._misses, self._fired = None, 0, False

    counter = property(lambda self: self._misses)
    best_loss = property(lambda self: self._level)
    early_stop = property(lambda self: self._fired)

    def __call__(self, val_loss: float) -> None:
        miss = self._level is not None and val_loss > self._level * (1 - self.min_delta)
        if miss:
            self._misses += 1
            self._fired = self._fired or self._misses >= self.patience
        else:
            if self._level is not None:
                self._misses = 0
            self._level = val_loss


class SaveBest:
    """The best-so-far snapshot of the reference's helper (myutils/misc.py:62-97) as one table: `update` replaces every slot with a
    detached copy when the loss is STRICTLY below the best one seen, which is global across phases and never reset.
    Slots are read as `best_<slot>` (`rendered_img` without the prefix, as the reference names it) or through `get_best()`."""

    SLOTS = ("albedo", "roughness", "metallic", "normal", "envmap", "rendered_img")

    def __init__(self):
        self.best_loss = float("inf")
        self._kept = dict.fromkeys(self.SLOTS)
        self.best_brdfnet_weight = None

    def update(self, loss, albedo, roughness, metallic, normal, envmap, rendered_img, brdfnet_weights=None) -> bool:
        if not loss < self.best_loss:
            return False
        self.best_loss = loss
        given = dict(zip(self.SLOTS, (albedo, roughness, metallic, normal, envmap, rendered_img)))
        self._kept = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else copy.deepcopy(v)) for k, v in given.items()}
        if brdfnet_weights is not None:
            self.best_brdfnet_weight = copy.deepcopy(brdfnet_weights)
        return True

    def __getattr__(self, name):
        slot = name[5:] if name.startswith("best_") else name
        kept = self.__dict__.get("_kept")
        if kept is not None and slot in kept:
            return kept[slot]
        raise AttributeError(name)

    def get_best(self):
        return dict(self._kept)


class DeviceSaveBest:
    """SaveBest without the host round trip: the strict `<` test and the snapshot copy run on the GPU
    (per image for a batch).  `best_loss` is a device tensor; read it with .item()/.tolist() when needed."""

    def __init__(self):
        self.best_loss: Optional[torch.Tensor] = None
        self.best: Dict[str, torch.Tensor] = {}

    def update(self, loss: torch.Tensor, **maps: torch.Tensor) -> None:
        loss = loss.detach()
        if self.best_loss is None:
            self.best_loss = torch.full_like(loss, float("inf"))
        better = loss < self.best_loss
        self.best_loss = torch.where(better, loss, self.best_loss)
        for k, v in maps.items():
            v = v.detach()
            sel = better.reshape(better.shape + (1,) * (v.ndim - better.ndim)) if better.ndim else better
            if k not in self.best:
                self.best[k] = v.clone()
            else:
                self.best[k] = torch.where(sel, v, self.best[k])


def _make_adam(params, lr):
    try:
        return torch.optim.Adam(params, lr=lr, fused=True)
    except (RuntimeError, TypeError, ValueError):
        return torch.optim.Adam(params, lr=lr)


def masked_mean_fill(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """`x[mask] = x[mask].mean()` of `--use_mask` (inverse_img_w_mi.py:379-381,509-511), out of place and without boolean
    indexing (no host synchronisation): every masked entry becomes the masked mean, so its gradient is the mean of the masked
    gradients, as autograd gives for the reference's in-place form.  x [H,W,C], mask [H,W] bool; a batch x [B,H,W,C] with its masks
    [B,H,W] is its images alone: one mean per image."""
    w = mask.to(x.dtype).reshape(mask.shape + (1,) * (x.ndim - mask.ndim))
    if x.ndim == 4:
        mean = (x * w).sum(dim=(1, 2, 3), keepdim=True) / (w.sum(dim=(1, 2, 3), keepdim=True) * (x.shape[-1] // w.shape[-1])).clamp_min(1.0)
    else:
        mean = (x * w).sum() / (w.sum() * (x.numel() // w.numel())).clamp_min(1.0)
    return x * (1.0 - w) + mean * w


def _make_adamw(params, lr):
    params = list(params)
    try:
        return torch.optim.AdamW(params, lr=lr, fused=True)
    except (RuntimeError, TypeError, ValueError):
        return torch.optim.AdamW(params, lr=lr)


class BrdfPhase:
    """Hot loop B, `model_name == 'none'` (inverse_img_w_mi.py:347-468)."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, albedo: torch.Tensor, roughness: torch.Tensor, metallic: torch.Tensor,
                 normal: Optional[torch.Tensor] = None, optimize_part: str = "arm", spp: int = 64, lr: float = 3e-4,
                 scale_delta: float = 0.1, saver: Optional[DeviceSaveBest] = None, mask: Optional[torch.Tensor] = None,
                 originals: Optional[Dict[str, torch.Tensor]] = None):
        """`originals`: the regulariser anchors albedo_ori / roughness_ori / metallic_ori / normal_ori, captured once before the
        loops (inverse_img_w_mi.py:189-201); default: this phase's start maps."""
        self.scene, self.gt, self.spp, self.scale_delta = scene, gt_image, spp, scale_delta
        self.mask = mask
        self.part = optimize_part
        self.maps = {"albedo": albedo, "roughness": roughness, "metallic": metallic, "normal": normal}
        self.originals = {k: (originals[k] if originals is not None and k in originals else v).detach().clone()
                          for k, v in self.maps.items() if v is not None}
        self.gt_srgb = _loss.linear_to_srgb(gt_image)
        keys = {"a": "albedo", "r": "roughness", "m": "metallic", "n": "normal"}
        self.opt_keys = [keys[c] for c in optimize_part if c in keys and not (c == "n" and scene.use_mesh_normal)]
        self.params = {k: torch.nn.Parameter(self.maps[k].detach().clone()) for k in self.opt_keys}
        self.opt = _make_adam(list(self.params.values()), lr)
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=100, gamma=0.8)   # :363-365
        self.saver = saver if saver is not None else DeviceSaveBest()
        self.last = {}

    def current_maps(self) -> Dict[str, torch.Tensor]:
        m = dict(self.maps)
        p = self.params
        if "albedo" in p:
            m["albedo"] = p["albedo"].clamp(0, 1)                    # :373
        if "roughness" in p:
            m["roughness"] = p["roughness"].clamp(0.07, 1)           # :375
        if "metallic" in p:
            m["metallic"] = p["metallic"].clamp(0, 1)                # :377
        if "normal" in p:
            m["normal"] = torch.nn.functional.normalize(p["normal"], p=2, dim=-1)   # :379
        if self.mask is not None:                                    # :379-381
            m["roughness"] = masked_mean_fill(m["roughness"], self.mask)
            m["metallic"] = masked_mean_fill(m["metallic"], self.mask)
        return m

    def step(self) -> torch.Tensor:
        m = self.current_maps()
        normal = None if self.scene.use_mesh_normal else m["normal"]
        pred = _render.render_w_brdf(self.scene, m["albedo"], m["roughness"], m["metallic"], normal, self.spp)   # :384-386
        parts = {k: m[k] for k in self.opt_keys}
        loss, loss_mse, pred_srgb, _ = _loss.brdf_loss(pred, self.gt, parts, self.originals, self.scale_delta, self.gt_srgb)
        loss.backward()                                                                                          # :420
        extra = {} if self.scene.use_mesh_normal else {"normal": m["normal"]}
        self.saver.update(loss_mse, albedo=m["albedo"], roughness=m["roughness"], metallic=m["metallic"], rendered_img=pred_srgb, **extra)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        if self.opt.param_groups[0]["lr"] > 1.5e-4:                                                              # :431-432
            self.sched.step()
        self.last = {"loss": loss.detach(), "loss_mse": loss_mse.detach()}
        return loss_mse.detach()


class EnvPhase:
    """Hot loop A (inverse_img_w_mi.py:236-254) with the light as the optimised tensor."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, light_init: torch.Tensor, spp: int = 64, lr: float = 1e-3,
                 saver: Optional[DeviceSaveBest] = None):
        self.scene, self.gt, self.spp = scene, gt_image, spp
        self.light = torch.nn.Parameter(light_init.detach().clone())
        self.opt = _make_adam([self.light], lr)
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=100, gamma=0.8)   # :226-227
        self.saver = saver if saver is not None else DeviceSaveBest()

    def step(self) -> torch.Tensor:
        pred = _render.render_envmap(self.scene, self.light, self.spp)
        loss, loss_mse, _ = _loss.env_loss(pred, self.gt)
        loss.backward()
        self.saver.update(loss_mse, envmap=self.light, rendered_img=pred)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.sched.step()
        return loss_mse.detach()


class EnvHeadPhase:
    """Hot loop A composed on the operator face (inverse_img_w_mi.py:236-254): `head()` -> `render_envmap` -> MSE + L1 on the
    gamma-2.2 images -> autograd -> optimiser.  Used when the scene has pixels without geometry (`Scene.set_mesh_mask`), which the
    fused `matpbr_env_phase_step` does not model; same SaveBest / EarlyStopping semantics, checked on the host per epoch."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, head, optimizer: torch.optim.Optimizer, spp: int = 64,
                 saver: Optional[DeviceSaveBest] = None):
        self.scene, self.gt, self.head, self.opt, self.spp = scene, gt_image, head, optimizer, int(spp)
        self.saver = saver if saver is not None else DeviceSaveBest()
        self.pred = None

    def step(self) -> torch.Tensor:
        data = self.head()
        pred = _render.render_envmap(self.scene, data, self.spp)                          # :240
        loss, loss_mse, _ = _loss.env_loss(pred, self.gt)                                 # :241-245
        loss.backward()
        self.saver.update(loss_mse, envmap=data, rendered_img=pred)                       # :247
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.pred = pred.detach()
        return loss_mse.detach()


class FusedBrdfPhase:
    """Hot loop B in `model_name == 'none'` mode (inverse_img_w_mi.py:347-468), one `matpbr_brdf_phase_step` call per
    iteration: render (clamp folded in; GGX-lobe samples only, the diffuse-lobe coefficients are cached at phase start because
    light and geometric normals are fixed, :317-342), loss statistics, SaveBest and EarlyStopping decisions, streaming loss
    backward with regularisers / clamp gating / best-so-far snapshot / Adam -- all on the device.  Same arithmetic as `BrdfPhase`
    (which composes the step from torch ops and is its parity reference).  `originals` are the regulariser anchors
    albedo_ori / roughness_ori / metallic_ori (:189-201), captured once before the loops; default: the start maps.

    EarlyStopping (`patience` > 0) lives in the statistics buffer: after an image has stopped every kernel skips it, so
    `run(k)` may enqueue k iterations blindly and `poll()` (one host sync) tells how far each image really got."""

    PARTS = {"a": 2, "r": 4, "m": 8}
    LAZY = True
    ROTATE_BEST = True      # SaveBest without copies in the lazy loop (MATPBR_FLAG_ROTATE_BEST); False: the copying step
    FOLD = True             # the folded, persistent step where the part has one (MatpbrBrdfPhase.lazy_fold); False: the generic step

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, albedo: torch.Tensor, roughness: torch.Tensor, metallic: torch.Tensor,
                 optimize_part: str = "arm", spp: int = 64, lr: float = 3e-4, scale_delta: float = 0.1, patience: int = 0,
                 min_delta: float = 0.0, best_mse: Optional[torch.Tensor] = None, history_len: int = 5000,
                 originals: Optional[Dict[str, torch.Tensor]] = None, keep_grads: bool = False, lazy: Optional[bool] = None,
                 lazy_tol: float = 1.0, attached_sampling: bool = False, rotate_best: Optional[bool] = None, fold: Optional[bool] = None,
                 share_gpu: bool = False):
        """`lazy` (default `FusedBrdfPhase.LAZY`): in parts that move the roughness, render from per-pixel local models in r and walk
        the GGX samples only of the pixels that left their model's validity interval (include/matpbr.h `matpbr_shade_fwd_lazy`).
        `attached_sampling` (lazy parts only): the roughness gradient through the GGX sample directions, the live reference's convention
        (mi_plugin.py:227-230,1335-1341), instead of the stop-gradient default.
        `rotate_best` (default: with `lazy`): SaveBest without copies -- MATPBR_FLAG_ROTATE_BEST, include/matpbr.h; False: the step
        kernel copies the snapshot in every improving iteration (the same values, bit for bit).
        `fold` (default `FusedBrdfPhase.FOLD`, lazy parts only): parts of r / m and part 'a' run the folded, persistent step
        (csrc/matpbr_pstep.hpp: the maps the part leaves alone are folded into the per-pixel models; two launches per iteration, the listed
        pixels walked inside the step launch); with `keep_grads` it forms the gradients of the maps the part moves only."""
        import ctypes

        from . import _lib, ops

        # the shading normals are constants of the phase: the geometric normals or, with use_mesh_normal False, the predicted normal map
        # (`scene.shading_normal()`, :335-340); a part that moves them ('n') is BrdfPhase's
        if "n" in optimize_part:
            raise NotImplementedError("FusedBrdfPhase optimises a/r/m under fixed shading normals; use BrdfPhase for 'n'")
        self._ct, self._libmod, self.ops = ctypes, _lib, ops
        self.scene, self.spp, self.scale_delta, self.part = scene, int(spp), float(scale_delta), optimize_part
        self.gt = gt_image.contiguous()
        dev = self.gt.device
        B = self.gt.shape[0] if self.gt.ndim == 4 else 1
        self.B, self.H, self.W = B, self.gt.shape[-3], self.gt.shape[-2]
        self.gt_srgb = _loss.linear_to_srgb(self.gt).contiguous()
        c = lambda t: t.detach().clone().contiguous()
        self._p = {"albedo": c(albedo), "roughness": c(roughness), "metallic": c(metallic)}
        self.orig = {k: c(originals[k] if originals is not None and k in originals else v).reshape(v.shape) for k, v in self._p.items()}
        self.g = {k: torch.empty_like(v) for k, v in self._p.items()} if keep_grads else None
        self.m = {k: torch.zeros_like(v) for k, v in self._p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self._p.items()}
        self._best = {k: c(v) for k, v in self._p.items()}
        self._best_img = torch.zeros_like(self.gt)
        self._pred = torch.empty_like(self.gt)
        self._dirty = False
        self.stats = ops.new_loss_stats(B, dev)
        if best_mse is not None:   # SaveBest.best_loss is global across phases and never reset (F11)
            self.stats[:, ops.STAT_BEST] = best_mse.to(dev).reshape(-1)
        # summed in double: the fp32 result then does not depend on how torch tiles the reduction (a batch row vs the image alone)
        self.stats[:, ops.STAT_GT_SUM] = self.gt.reshape(B, -1).double().sum(dim=1).float()
        lib = _lib.load()
        self.ws = torch.empty(int(lib.matpbr_brdf_phase_workspace_bytes(self.H, self.W, B)) // 4 + 1, dtype=torch.float32, device=dev)
        self.hist = torch.zeros((history_len, B), dtype=torch.float32, device=dev)
        self.n = scene.shading_normal().contiguous()
        light = scene.light.detach()
        self.light = (light.unsqueeze(0).expand(B, -1, -1) if (B > 1 and light.ndim == 2) else light).contiguous()
        self.base_lr, self.t = float(lr), 0
        # light and shading normals do not change during the phase: the diffuse lobe reduces to 9 coefficients per pixel
        self.dcache = ops.diffuse_cache(self.n, self.light, self.spp, scene.fov)
        self.jac = ops.plane9(self._p["albedo"])
        ph = _lib.MatpbrBrdfPhase()
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        ph.pa, ph.pr, ph.pm = P(self._p["albedo"]), P(self._p["roughness"]), P(self._p["metallic"])
        ph.n, ph.light, ph.gt_srgb = P(self.n), P(self.light), P(self.gt_srgb)
        ph.a0, ph.r0, ph.m0 = P(self.orig["albedo"]), P(self.orig["roughness"]), P(self.orig["metallic"])
        ph.dcache, ph.pred, ph.jac = P(self.dcache), P(self._pred), P(self.jac)
        self.lazy = bool(self.LAZY if lazy is None else lazy)
        # the folded step has the gradients of the maps its part moves; a caller that wants all three keeps the generic step
        self.fold = self.lazy and bool(self.FOLD if fold is None else fold) and optimize_part in ("r", "m", "rm", "mr", "a")
        if self.g is not None:
            if self.fold:
                for k in self.g:
                    self.g[k].zero_()
                if "a" in optimize_part:
                    ph.d_a = P(self.g["albedo"])
                else:
                    ph.d_r, ph.d_m = P(self.g["roughness"]), P(self.g["metallic"])
            else:
                ph.d_a, ph.d_r, ph.d_m = P(self.g["albedo"]), P(self.g["roughness"]), P(self.g["metallic"])
        for i, k in enumerate(("albedo", "roughness", "metallic")):
            ph.adam_m[i], ph.adam_v[i] = self.m[k].data_ptr(), self.v[k].data_ptr()
        ph.best_a, ph.best_r, ph.best_m, ph.best_img = P(self._best["albedo"]), P(self._best["roughness"]), P(self._best["metallic"]), P(self._best_img)
        ph.stats, ph.history, ph.workspace = P(self.stats), P(self.hist), P(self.ws)
        ph.workspace_bytes = self.ws.numel() * 4
        ph.H, ph.W, ph.batch, ph.spp = self.H, self.W, B, self.spp
        ph.fov_x_deg, ph.scale_delta = scene.fov, self.scale_delta
        ph.part_mask = sum(self.PARTS[ch] for ch in optimize_part)
        ph.es_patience, ph.es_min_delta, ph.hist_len = int(patience), float(min_delta), int(history_len)
        # parts that leave the roughness alone: the specular sums of every pixel are constants of the part (kept from its first render)
        # lazy (default): every part renders from the per-pixel models -- in a part that leaves the roughness alone no pixel ever leaves its
        # model's interval, so its iteration is the same two launches with no re-sampling at all
        self.s1cache = None if ("r" in optimize_part or self.lazy) else torch.empty((3,) + tuple(self.jac.shape[1:]), dtype=torch.float32, device=self.jac.device)
        ph.s1cache = P(self.s1cache) if self.s1cache is not None else None
        self.lazy_state = ops.lazy_state(self._p["albedo"]) if self.lazy else None
        ph.lazy_state = P(self.lazy_state) if self.lazy else None
        ph.lazy_tol = float(lazy_tol)
        self.lazy_fold = ops.lazy_fold(self._p["albedo"]) if self.fold else None
        ph.lazy_fold = P(self.lazy_fold) if self.fold else None
        if attached_sampling and not self.lazy:
            raise ValueError("attached_sampling needs the lazy path")
        ph.flags = (ops.FLAG_ATTACHED_SAMPLING if attached_sampling else 0) | (0 if self.fold else ops.FLAG_GENERIC_STEP) | \
            (ops.FLAG_SHARE_GPU if share_gpu else 0)             # share_gpu: a group of a PipelinedBrdfPhase (the step leaves room on every CU)
        # pixels without geometry (Scene.set_mesh_mask): build what the first step would build, give those pixels constant models
        # (they render the environment along their camera ray and receive no material gradient), and tell the steps so
        self.bg_mask = scene.bg_mask
        if self.bg_mask is not None:
            bg_rgb = scene.background_radiance(self.light)
            pc = (self._p["albedo"], self._p["roughness"], self._p["metallic"])
            if self.lazy:
                ops.shade_fwd_lazy(*pc, self.n, self.light, self.spp, self.dcache, self.lazy_state, out=self._pred, jac16=self.jac.view(torch.int32)[:5],
                                   force=True, clamp_params=True, stats=self.stats, fov_x_deg=scene.fov)      # per-image parity floor from the statistics row
                ops.background_into_lazy_state(self.lazy_state, self._p["albedo"], self.bg_mask, bg_rgb, self._p["roughness"])
            elif self.s1cache is not None:
                ops.shade_fwd(*pc, self.n, self.light, self.spp, scene.fov, clamp_params=True, out=self._pred, dcache=self.dcache, jac=self.jac, s1=self.s1cache)
                ops.background_into_jac(self.jac, self.s1cache, self.bg_mask, bg_rgb)
            else:
                raise NotImplementedError("pixels without geometry need the lazy path in parts that optimise the roughness")
            ph.flags |= ops.FLAG_MODELS_READY
        # lazy: the step's last launch also renders the next iterate (into pred_next); the two render buffers swap roles every step
        self._pred_bufs = [self._pred, torch.empty_like(self.gt)] if self.lazy else None
        self._pred_cur = 0
        ph.pred_next = P(self._pred_bufs[1]) if self.lazy else None
        # lazy: SaveBest without copies (MATPBR_FLAG_ROTATE_BEST): the live maps and the render rotate between two buffers each on the
        # device; `p`, `best`, `best_img` and `pred` are resolved (one launch) when they are next read
        self.rotate = (self.lazy and self.ROTATE_BEST) if rotate_best is None else (bool(rotate_best) and self.lazy)
        if self.rotate:
            ph.flags |= ops.FLAG_ROTATE_BEST
        self._ph, self._lib = ph, lib
        if os.environ.get("MATPBR_POISON"):      # debugging aid: every buffer the kernels are expected to WRITE before they read it starts as garbage
            for buf in (self.ws, self.jac, self._pred, self._pred_bufs[1] if self.lazy else None):
                if buf is not None:
                    buf.fill_(float(os.environ["MATPBR_POISON"]))
            if self.lazy_fold is not None:
                self.lazy_fold.fill_(255)          # (bytes: every word of the folded planes, the walk sums, counters and queue a NaN pattern / a huge count)

    def lr_at(self, t0: int) -> float:
        """Learning rate of the iteration with 0-based index t0: StepLR(100, 0.8) stepped only while lr > 1.5e-4 (:363-365,431-432)."""
        w = t0 // 100                                            # (the same value for a hundred iterations: remembered, the loop below runs once per window)
        cached = getattr(self, "_lr_window", (-1, 0.0))      # (other phase classes borrow this method)
        if cached[0] == w:
            return cached[1]
        lr, k = self.base_lr, 0
        while lr > 1.5e-4 and (k + 1) * 100 <= t0:
            lr *= 0.8
            k += 1
        self._lr_window = (w, lr)
        return lr

    def step(self) -> None:
        ct = self._ct
        dev = self.gt.device
        if self.ops.KernelTimer.active is None and torch.cuda.current_device() == dev.index:
            # the loop's own path: no context managers around the one call (a rank's iteration is enqueued by one Python thread: bench.py host_enqueue)
            code = self._lib.matpbr_brdf_phase_step(ct.byref(self._ph), self.t + 1, self.lr_at(self.t), ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        else:
            with torch.cuda.device(dev), self.ops._timed("brdf_phase_step"):
                code = self._lib.matpbr_brdf_phase_step(ct.byref(self._ph), self.t + 1, self.lr_at(self.t),
                                                        ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        self._libmod.check(code, "matpbr_brdf_phase_step")
        self._advance()

    def _resolve(self) -> None:
        if self._dirty:
            self._dirty = False
            with torch.cuda.device(self.gt.device):
                code = self._lib.matpbr_brdf_phase_resolve(self._ct.byref(self._ph), self.t,
                                                           self._ct.c_void_p(torch.cuda.current_stream(self.gt.device).cuda_stream))
            self._libmod.check(code, "matpbr_brdf_phase_resolve")

    @property
    def p(self) -> Dict[str, torch.Tensor]:
        """The raw parameters after the iterations run so far."""
        self._resolve()
        return self._p

    @property
    def best(self) -> Dict[str, torch.Tensor]:
        """SaveBest's maps (clamped), as of the best iteration so far."""
        self._resolve()
        return self._best

    @property
    def best_img(self) -> torch.Tensor:
        self._resolve()
        return self._best_img

    @property
    def pred(self) -> torch.Tensor:
        """Lazy mode (with either form of SaveBest): the render of the CURRENT parameters (what the next iteration will judge; the last launch
        of a step renders it).  `lazy=False`: the render the last iteration judged."""
        self._resolve()
        return self._pred

    def _advance(self) -> None:
        ct = self._ct
        self.t += 1
        if self.rotate or self.fold:
            self._dirty = True                    # (a folded phase stores no render: matpbr_brdf_phase_resolve forms it when somebody reads it)
        if not self.rotate and self._pred_bufs is not None:
            # the buffer the step's last launch rendered the next iterate into becomes `pred` of the next step -- and `self.pred`, as in the
            # rotating form: the render of the current parameters
            self._pred_cur ^= 1
            self._pred = self._pred_bufs[self._pred_cur]
            self._ph.pred = ct.c_void_p(self._pred_bufs[self._pred_cur].data_ptr())
            self._ph.pred_next = ct.c_void_p(self._pred_bufs[self._pred_cur ^ 1].data_ptr())

    def launch_stage(self, stages: int) -> None:
        """Enqueue some stages of the NEXT iteration without advancing the phase (1 render, 2 statistics, 4 backward + Adam, 8 the lazy
        mode's resampling launch; kernel timing)."""
        ct = self._ct
        with torch.cuda.device(self.gt.device):
            code = self._lib.matpbr_brdf_phase_stages(ct.byref(self._ph), self.t + 1, self.lr_at(self.t), int(stages),
                                                      ct.c_void_p(torch.cuda.current_stream(self.gt.device).cuda_stream))
        self._libmod.check(code, "matpbr_brdf_phase_stages")

    def step_timed(self, events: list) -> None:
        """`step()` with HIP events on its backward launch (the backward pass + Adam, in the lazy mode also the next render) and behind the
        resampling launch that follows it, appended to `events` as (begin, end, after the resampling launch[, before it]): the in-loop durations
        for bench.py's roofline."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.launch_stage(1 | 2)
        if self.fold:
            # the folded step: the kernel's own begin / end timestamps (matpbr_brdf_phase_stages_timed) -- events recorded around the launch add the
            # stream's dispatch latency on both sides (2-5 us on a 50 us kernel: the rocprofv3 trace and the events disagreed by that much)
            ct = self._ct
            e0.record(); e1.record()         # (creates the underlying hipEvents)
            with torch.cuda.device(self.gt.device):
                code = self._lib.matpbr_brdf_phase_stages_timed(ct.byref(self._ph), self.t + 1, self.lr_at(self.t), 4, ct.c_void_p(e0.cuda_event),
                                                                ct.c_void_p(e1.cuda_event), ct.c_void_p(torch.cuda.current_stream(self.gt.device).cuda_stream))
            self._libmod.check(code, "matpbr_brdf_phase_stages_timed")
            ew = torch.cuda.Event(enable_timing=True)
            ew.record()
        else:
            e0.record()
            self.launch_stage(4)
            e1.record()
            ew = e1
        self.launch_stage(8)             # the resampling launch of the lazy mode (nothing otherwise)
        e2 = torch.cuda.Event(enable_timing=True)
        e2.record()
        events.append((e0, e1, e2) if ew is e1 else (e0, e1, e2, ew))
        self._advance()

    def run(self, n: int) -> None:
        for _ in range(n):
            self.step()

    def poll(self) -> Dict[str, torch.Tensor]:
        """One host synchronisation: per-image stop flags, iterations actually executed, best / last MSE."""
        st = self.stats.cpu()
        o = self.ops
        return {"stopped": st[:, o.STAT_STOPPED] > 0.5, "iters": st[:, o.STAT_ITERS].to(torch.int64), "best_mse": st[:, o.STAT_BEST],
                "mse": st[:, o.STAT_MSE], "loss": st[:, o.STAT_LOSS]}

    def history(self) -> torch.Tensor:
        """[iterations enqueued so far, B] loss_mse of every executed iteration (device tensor; rows past an image's stop are 0)."""
        return self.hist[: self.t]

    def current_maps(self) -> Dict[str, torch.Tensor]:
        p = self.p
        return {"albedo": p["albedo"].clamp(0, 1), "roughness": p["roughness"].clamp(0.07, 1), "metallic": p["metallic"].clamp(0, 1)}


class _SceneGroup:
    """What FusedBrdfPhase reads of a Scene, for a slice of its batch."""

    def __init__(self, scene: _render.Scene, sl: slice):
        self.use_mesh_normal, self.fov = scene.use_mesh_normal, scene.fov
        self._n = scene.shading_normal()[sl]
        light = scene.light
        self.light = light if light.ndim == 2 else light[sl]
        self.bg_mask = None if scene.bg_mask is None else scene.bg_mask[sl]
        self._bg_basis = None if scene.bg_mask is None else scene.bg_basis[sl]

    def shading_normal(self) -> torch.Tensor:
        return self._n

    def background_radiance(self, light: torch.Tensor) -> torch.Tensor:
        if light.ndim == 2:
            light = light.unsqueeze(0).expand(self._bg_basis.shape[0], -1, -1)
        return (self._bg_basis @ light).reshape(self.bg_mask.shape + (3,))


class _SceneImage(_SceneGroup):
    """One image of a batched Scene, as the single-image phases read a Scene."""

    def __init__(self, scene: _render.Scene, i: int):
        self.use_mesh_normal, self.fov = scene.use_mesh_normal, scene.fov
        self._n = scene.shading_normal()[i]
        light = scene.light
        self.light = light if light.ndim == 2 else light[i]
        self.bg_mask = None if scene.bg_mask is None else scene.bg_mask[i]
        self._bg_basis = None if scene.bg_mask is None else scene.bg_basis[i]

    def background_radiance(self, light: torch.Tensor) -> torch.Tensor:
        return (self._bg_basis @ light).reshape(self.bg_mask.shape + (3,))


class _BatchOfOne:
    """A single-image phase seen as a batch of one image: what PipelinedBrdfPhase concatenates."""

    def __init__(self, ph):
        self._ph = ph

    def __getattr__(self, name):
        return getattr(self._ph, name)

    p = property(lambda self: {k: v.unsqueeze(0) for k, v in self._ph.p.items()})
    best = property(lambda self: {k: v.unsqueeze(0) for k, v in self._ph.best.items()})
    best_img = property(lambda self: self._ph.best_img.unsqueeze(0))
    pred = property(lambda self: self._ph.pred.unsqueeze(0))

    def current_maps(self) -> Dict[str, torch.Tensor]:
        return {k: v.unsqueeze(0) for k, v in self._ph.current_maps().items()}


_GROUP_STREAMS: Dict[tuple, list] = {}


def _group_streams(dev: torch.device, groups: int) -> list:
    """The streams the groups of a PipelinedBrdfPhase step on: ONE set per device and group count for the whole process.  A set per phase
    object (through round 5) meant new HIP streams for every part of every loop of a run, and HIP multiplexes a process's streams onto a
    few hardware queues: once two groups' streams share a queue their launches are serialised again -- the overlap the groups exist for
    is gone without a trace in the results (bench.py --mode fused --images-per-gpu 8, whose timed phase was the third one built in its
    process: 70 k image-iterations/s against the 96-112 k of the same phase built first).  Phases that share the streams are ordered
    against each other by them, which is what consecutive parts of a run want anyway."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(groups))
    if key not in _GROUP_STREAMS:
        _GROUP_STREAMS[key] = [torch.cuda.Stream(dev) for _ in range(groups)]
    return _GROUP_STREAMS[key]


class PipelinedBrdfPhase:
    """A batch of images as GROUPS of images, each a `FusedBrdfPhase` stepping on a stream of its own.  An image's iteration does not depend on the
    images beside it (a batch is its images alone, bit for bit: tests/test_gpu_lazy.py), so the groups are independent -- and the walk and
    statistics launches of one group (latency-bound: a handful of waves behind a chain of round trips, a third of the iteration) run under the
    streaming step of the other, whose 512 workgroups leave room on every CU for them (MATPBR_FLAG_SHARE_GPU).  Same results as one
    FusedBrdfPhase over the whole batch, bit for bit; same interface.  8 x 512 x 512, part 'rm': 80.6 -> 75 us per iteration of the batch."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, albedo: torch.Tensor, roughness: torch.Tensor, metallic: torch.Tensor,
                 groups: int = 2, best_mse: Optional[torch.Tensor] = None, originals: Optional[Dict[str, torch.Tensor]] = None, make_phase=None,
                 **kw):
        if gt_image.ndim != 4 or gt_image.shape[0] % groups or (groups < 2 and make_phase is None):
            raise ValueError("PipelinedBrdfPhase: a batch [B,H,W,3] whose size the number of groups divides")
        B, dev = gt_image.shape[0], gt_image.device
        per = B // groups
        self.B, self.groups = B, groups
        self.streams = _group_streams(dev, groups)
        self.phases = []
        here = torch.cuda.current_stream(dev)
        if make_phase is None:
            scene.shading_normal()                                   # computed (and cached) on the CALLER's stream: every group's stream waits for it below
        for gi, st in enumerate(self.streams):
            sl = slice(gi * per, (gi + 1) * per)
            st.wait_stream(here)                                     # the caller's tensors are ready on the caller's stream
            with torch.cuda.stream(st):                              # the group's buffers and its set-up launches belong to its stream
                group_kw = dict(best_mse=None if best_mse is None else best_mse.reshape(-1)[sl],
                                originals=None if originals is None else {k: v[sl] for k, v in originals.items()}, **kw)
                if make_phase is not None:
                    self.phases.append(make_phase(sl, group_kw))
                else:
                    self.phases.append(FusedBrdfPhase(_SceneGroup(scene, sl), gt_image[sl], albedo[sl], roughness[sl], metallic[sl], share_gpu=True,
                                                      **group_kw))
        self.ops = self.phases[0].ops

    def step(self) -> None:
        for ph, st in zip(self.phases, self.streams):
            with torch.cuda.stream(st):
                ph.step()

    def run(self, n: int) -> None:
        for _ in range(n):
            self.step()

    def _join(self) -> None:
        here = torch.cuda.current_stream(self.phases[0].gt.device)
        for st in self.streams:
            here.wait_stream(st)

    def _release(self) -> None:
        """The groups' streams wait for the caller's: what the caller's stream has just read of the groups' buffers (they belong to the groups'
        streams in the allocator's books) is read before those streams touch -- or, after the phase is gone, re-use -- that memory."""
        here = torch.cuda.current_stream(self.phases[0].gt.device)
        for st in self.streams:
            st.wait_stream(here)

    def _cat(self, get, dim: int = 0) -> torch.Tensor:
        self._join()
        out = torch.cat([get(ph) for ph in self.phases], dim=dim)
        self._release()
        return out

    @property
    def t(self) -> int:
        return self.phases[0].t

    @property
    def stats(self) -> torch.Tensor:
        return self._cat(lambda ph: ph.stats)

    def _dict(self, name: str) -> Dict[str, torch.Tensor]:
        self._join()
        parts = []
        for ph, st in zip(self.phases, self.streams):
            with torch.cuda.stream(st):                              # (`p`, `best` may enqueue the resolving launch of the rotating SaveBest)
                parts.append(getattr(ph, name))
        self._join()
        out = {k: torch.cat([d[k] for d in parts], dim=0) for k in parts[0]}
        self._release()
        return out

    p = property(lambda self: self._dict("p"))
    best = property(lambda self: self._dict("best"))

    def _img(self, name: str) -> torch.Tensor:
        self._join()
        parts = []
        for ph, st in zip(self.phases, self.streams):
            with torch.cuda.stream(st):
                parts.append(getattr(ph, name))
        self._join()
        out = torch.cat(parts, dim=0)
        self._release()
        return out

    best_img = property(lambda self: self._img("best_img"))
    pred = property(lambda self: self._img("pred"))

    def lr_at(self, t0: int) -> float:
        return self.phases[0].lr_at(t0)

    def poll(self) -> Dict[str, torch.Tensor]:
        self._join()
        polls = [ph.poll() for ph in self.phases]                     # (host copies: a synchronisation each)
        return {k: torch.cat([p_[k] for p_ in polls], dim=0) for k in polls[0]}

    def history(self) -> torch.Tensor:
        return self._cat(lambda ph: ph.history(), dim=1)

    def current_maps(self) -> Dict[str, torch.Tensor]:
        p = self.p
        return {"albedo": p["albedo"].clamp(0, 1), "roughness": p["roughness"].clamp(0.07, 1), "metallic": p["metallic"].clamp(0, 1)}


class NormalBrdfPhase:
    """Hot loop B in `model_name == 'none'` mode for a part of --opt_order that MOVES THE NORMAL MAP ('n', 'armn', ...; use_mesh_normal False:
    inverse_img_w_mi.py:356-432), launch by launch on the C ABI: the render under the current maps, the loss statistics with SaveBest /
    EarlyStopping on the device, d loss / d pred, the backward render (the normal gradient: `shade_bwd_nl_kernel` walks both lobes' directions per
    pixel; the material gradients: closed forms of the nine planes the forward render leaves), and one kernel for the regularisers (L1(normal, normal_ori) among them), the clamp gating, NF.normalize's backward,
    the snapshot of an improving iteration and Adam (include/matpbr.h `MatpbrNormalStep`): nine launches per iteration, no autograd, no
    framework losses, no host synchronisation per epoch.  Same arithmetic as `BrdfPhase` (its parity reference: tests/test_gpu_parity.py);
    same interface as `FusedBrdfPhase` (`run`, `poll`, `history`, `p`, `best`, `best_img`, `lr_at`)."""

    KEYS = ("albedo", "roughness", "metallic", "normal")

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, albedo: torch.Tensor, roughness: torch.Tensor, metallic: torch.Tensor,
                 normal: torch.Tensor, optimize_part: str = "armn", spp: int = 64, lr: float = 3e-4, scale_delta: float = 0.1, patience: int = 0,
                 min_delta: float = 0.0, best_mse: Optional[torch.Tensor] = None, history_len: int = 5000,
                 originals: Optional[Dict[str, torch.Tensor]] = None):
        import ctypes

        from . import _lib, ops

        if scene.use_mesh_normal or "n" not in optimize_part:
            raise NotImplementedError("NormalBrdfPhase runs the parts that move the normal map (use_mesh_normal False, 'n' in the part); "
                                      "FusedBrdfPhase runs the others")
        self._ct, self._libmod, self.ops = ctypes, _lib, ops
        self.scene, self.spp, self.scale_delta, self.part = scene, int(ops.check_spp(spp)), float(scale_delta), optimize_part
        self.gt = gt_image.contiguous()
        dev = self.gt.device
        B = self.gt.shape[0] if self.gt.ndim == 4 else 1
        self.B, self.H, self.W = B, self.gt.shape[-3], self.gt.shape[-2]
        self.gt_srgb = _loss.linear_to_srgb(self.gt).contiguous()
        c = lambda t: t.detach().clone().contiguous()
        self.p = {"albedo": c(albedo), "roughness": c(roughness), "metallic": c(metallic), "normal": c(normal)}
        self.orig = {k: c(originals[k] if originals is not None and k in originals else v).reshape(v.shape) for k, v in self.p.items()}
        live = {"albedo": "a" in optimize_part, "roughness": "r" in optimize_part, "metallic": "m" in optimize_part, "normal": True}
        # what the renders take: the maps of the part clamped / normalised (:372-379; the step kernel rewrites them from the new parameters), the
        # others as they are
        lims = {"albedo": (0.0, 1.0), "roughness": (0.07, 1.0), "metallic": (0.0, 1.0)}
        self.c = {k: (self.p[k].clamp(*lims[k]) if live[k] else self.p[k].clone()) for k in lims}
        self.c["normal"] = torch.nn.functional.normalize(self.p["normal"], p=2, dim=-1).contiguous()
        self.g = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.best = {k: c(v) for k, v in self.c.items()}
        self.best_img = torch.zeros_like(self.gt)
        self.pred = torch.empty_like(self.gt)
        self.d_pred = torch.zeros_like(self.gt)
        self.jac = ops.plane9(self.p["albedo"]) if any(live[k] for k in ("albedo", "roughness", "metallic")) else None
        self.stats = ops.new_loss_stats(B, dev)
        if best_mse is not None:   # SaveBest.best_loss is global across phases and never reset (F11)
            self.stats[:, ops.STAT_BEST] = best_mse.to(dev).reshape(-1)
        lib = self._lib = _lib.load()
        self.ws = torch.empty(int(lib.matpbr_brdf_loss_workspace_bytes(B)) // 4 + 1, dtype=torch.float32, device=dev)
        self.hist = torch.zeros((history_len, B), dtype=torch.float32, device=dev)
        self.ln_part = torch.zeros((B, (self.H * self.W + 255) // 256), dtype=torch.float32, device=dev)
        light = scene.light.detach()
        self.light = (light.unsqueeze(0).expand(B, -1, -1) if (B > 1 and light.ndim == 2) else light).contiguous()
        self.base_lr, self.t = float(lr), 0
        self.patience, self.min_delta, self.hist_len = int(patience), float(min_delta), int(history_len)
        self.mask = ops.part_mask(optimize_part)
        self.cam = ops.MatpbrCamera(float(scene.fov))
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        ns = _lib.MatpbrNormalStep()
        for i, (k, ch) in enumerate(zip(self.KEYS, "armn")):
            setattr(ns, "p" + ch, P(self.p[k]))
            setattr(ns, "c" + ch, P(self.c[k]))
            setattr(ns, "d_" + ch, P(self.g[k]) if live[k] else None)
            setattr(ns, ch + "0", P(self.orig[k]))
            setattr(ns, "best_" + ch, P(self.best[k]) if live[k] else None)      # a map the part leaves alone: its snapshot is the map
            ns.adam_m[i], ns.adam_v[i] = self.m[k].data_ptr(), self.v[k].data_ptr()
        ns.best_img, ns.pred, ns.stats, ns.ln_part = P(self.best_img), P(self.pred), P(self.stats), P(self.ln_part)
        ns.H, ns.W, ns.batch, ns.part_mask, ns.scale_delta = self.H, self.W, B, self.mask, self.scale_delta
        self._ns, self._live = ns, live

    lr_at = FusedBrdfPhase.lr_at

    def step(self) -> None:
        ct, lib, o, P = self._ct, self._lib, self.ops, (lambda t: self._ct.c_void_p(t.data_ptr()) if t is not None else None)
        H, W, B, c, p, g = self.H, self.W, self.B, self.c, self.p, self.g
        want_mat = any(self._live[k] for k in ("albedo", "roughness", "metallic"))
        with torch.cuda.device(self.gt.device):
            st = ct.c_void_p(torch.cuda.current_stream(self.gt.device).cuda_stream)
            chk = self._libmod.check
            # the render; with material maps in the part it also leaves the nine planes their gradients are closed forms of (a second walk of
            # the samples in the backward pass otherwise: 30 of the iteration's 87 us)
            chk(lib.matpbr_shade_fwd_ex(P(c["albedo"]), P(c["roughness"]), P(c["metallic"]), P(c["normal"]), P(self.light), o.LIGHT_SH25, o.NSH, None,
                                        P(self.pred), P(self.jac) if want_mat else None, H, W, B, self.spp, ct.byref(self.cam), 0, st), "matpbr_shade_fwd_ex")
            chk(lib.matpbr_brdf_loss_stats_es(P(self.pred), P(self.gt), P(self.gt_srgb), P(p["albedo"]), P(p["roughness"]), P(p["metallic"]),
                                              P(self.orig["albedo"]), P(self.orig["roughness"]), P(self.orig["metallic"]), self.scale_delta,
                                              P(self.stats), P(self.ws), self.ws.numel() * 4, H, W, B, self.mask, self.patience, self.min_delta,
                                              P(self.hist), self.hist_len, st), "matpbr_brdf_loss_stats_es")
            chk(lib.matpbr_brdf_loss_dpred(P(self.pred), P(self.gt_srgb), P(self.stats), P(self.d_pred), H, W, B, st), "matpbr_brdf_loss_dpred")
            if want_mat:
                chk(lib.matpbr_shade_bwd_jac(P(c["albedo"]), P(c["roughness"]), P(c["metallic"]), P(self.jac), P(self.d_pred), P(g["albedo"]),
                                             P(g["roughness"]), P(g["metallic"]), H, W, B, st), "matpbr_shade_bwd_jac")
            chk(lib.matpbr_shade_bwd(P(c["albedo"]), P(c["roughness"]), P(c["metallic"]), P(c["normal"]), P(self.light), o.LIGHT_SH25, o.NSH,
                                     P(self.d_pred), None, None, None, P(g["normal"]), None, None, 0, H, W, B, self.spp, ct.byref(self.cam), 0, st),
                "matpbr_shade_bwd")
            chk(lib.matpbr_brdf_normal_step(ct.byref(self._ns), self.t + 1, self.lr_at(self.t), st), "matpbr_brdf_normal_step")
        self.t += 1

    def run(self, n: int) -> None:
        for _ in range(n):
            self.step()

    poll = FusedBrdfPhase.poll
    history = FusedBrdfPhase.history

    def loss(self) -> torch.Tensor:
        """The last iteration's loss with the normal regulariser (`stats` carries the three material regularisers): [B]."""
        return self.stats[:, self.ops.STAT_LOSS] + self.scale_delta * self.ln_part.sum(dim=1) / (3.0 * self.H * self.W)

    def current_maps(self) -> Dict[str, torch.Tensor]:
        p = self.p
        lims = {"albedo": (0.0, 1.0), "roughness": (0.07, 1.0), "metallic": (0.0, 1.0)}
        out = {k: (p[k].clamp(*lims[k]) if self._live[k] else p[k]) for k in lims}
        out["normal"] = torch.nn.functional.normalize(p["normal"], p=2, dim=-1)
        return out


class MaskedBrdfPhase:
    """Hot loop B, `--model_name none` under `--use_mask` (inverse_img_w_mi.py:347-468 with :379-381), launch by launch on the C ABI.
    Inside the mask the roughness and the metallic the render sees are the masked means of the clamped maps: two image-wide reductions per
    iteration sit between the optimiser step and the next render (the mean of the new parameters) and between the render's backward pass and
    the optimiser step (the mean of the masked gradients), so the one-launch step of `FusedBrdfPhase` does not apply; the iteration is
    fill x2, render (+ jac planes), statistics with the SaveBest / EarlyStopping commit on the device, streaming backward from the jac
    planes (+ snapshots), masked gradient means, one Adam launch over the flat parameter buffer.  One image, geometric normals.
    Same interface as `FusedBrdfPhase` (run, poll, lr_at, current_maps, stats, best, best_img, pred, history)."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, albedo: torch.Tensor, roughness: torch.Tensor, metallic: torch.Tensor,
                 mask: torch.Tensor, optimize_part: str = "arm", spp: int = 64, lr: float = 3e-4, scale_delta: float = 0.1, patience: int = 0,
                 min_delta: float = 0.0, best_mse: Optional[torch.Tensor] = None, history_len: int = 5000,
                 originals: Optional[Dict[str, torch.Tensor]] = None):
        from . import _lib, ops

        if gt_image.ndim != 3 or not scene.use_mesh_normal or "n" in optimize_part:
            raise NotImplementedError("MaskedBrdfPhase: one image, geometric normals, parts of a / r / m")
        self.ops, self._libmod, self.scene, self.part = ops, _lib, scene, optimize_part
        self.spp, self.scale_delta, self.base_lr = int(spp), float(scale_delta), float(lr)
        self.gt = gt_image.contiguous()
        dev = self.gt.device
        H, W = self.H, self.W = self.gt.shape[0], self.gt.shape[1]
        n = H * W
        self.gt_srgb = _loss.linear_to_srgb(self.gt).contiguous()
        keys = {"a": "albedo", "r": "roughness", "m": "metallic"}
        self.live = [keys[c] for c in "arm" if c in optimize_part]
        start = {"albedo": albedo.reshape(H, W, 3), "roughness": roughness.reshape(H, W, 1), "metallic": metallic.reshape(H, W, 1)}
        sizes = {"albedo": 3 * n, "roughness": n, "metallic": n}
        total = sum(sizes[k] for k in self.live)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)               # the optimised maps, one buffer: one Adam launch
        self.gflat = torch.zeros_like(self.flat)
        self.adam_m, self.adam_v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.p, self.g, off = {}, {}, 0
        for k in ("albedo", "roughness", "metallic"):
            if k in self.live:
                self.p[k] = self.flat[off:off + sizes[k]].view(start[k].shape)
                self.p[k].copy_(start[k])
                self.g[k] = self.gflat[off:off + sizes[k]].view(start[k].shape)
                off += sizes[k]
            else:
                self.p[k] = start[k].detach().clone().contiguous()
                self.g[k] = torch.zeros_like(self.p[k])
        self.orig = {k: (originals[k] if originals is not None and k in originals else start[k]).detach().reshape(start[k].shape).clone().contiguous()
                     for k in start}
        self.mask_u8 = mask.to(dev).reshape(H, W).to(torch.uint8).contiguous()
        self.fed = {k: torch.empty(H, W, 1, dtype=torch.float32, device=dev) for k in ("roughness", "metallic")}
        self.bounds = {"roughness": (0.07, 1.0), "metallic": (0.0, 1.0)}                  # the clamps of :375,377, taken before the mean
        self.hyper = torch.tensor([self.base_lr, 0.0], dtype=torch.float32, device=dev)   # lr, Adam's step count (advanced on the device)
        self._lr = self.base_lr
        self.stats = ops.new_loss_stats(1, dev)
        if best_mse is not None:
            self.stats[:, ops.STAT_BEST] = best_mse.to(dev).reshape(-1)
        self.patience, self.min_delta = int(patience), float(min_delta)
        self.hist = torch.zeros((history_len, 1), dtype=torch.float32, device=dev)
        self.ws = torch.empty(int(_lib_ws(1)) // 4, dtype=torch.float32, device=dev)
        self.pred = torch.empty_like(self.gt)
        self.best = {k: v.detach().clone() for k, v in self.p.items()}
        self.best_img = torch.zeros_like(self.gt)
        self._n = scene.shading_normal().contiguous()
        self._light = scene.light.detach().contiguous()
        self.dcache = ops.diffuse_cache(self._n, self._light, self.spp, scene.fov)
        self.jac = ops.plane9(self.gt)
        self.s1 = None if "roughness" in self.live else torch.empty((3, 1, H, W), dtype=torch.float32, device=dev)
        self._bg_mask = scene.bg_mask
        if self._bg_mask is not None:
            self._bg_rgb = scene.background_radiance(self._light).reshape(-1, 3).contiguous()
            self._bg_idx = ops.background_index(self._bg_mask)          # once: boolean-mask indexing would synchronise with the host every iteration
            self._bg_rows = self._bg_rgb[self._bg_idx].contiguous()
            self._bg_rows_t = self._bg_rows.t().contiguous()
        self.t = 0

    lr_at = FusedBrdfPhase.lr_at

    def _fed_maps(self) -> Dict[str, torch.Tensor]:
        d = dict(self.p)
        for k in ("roughness", "metallic"):                     # a map the part does not optimise is taken as it is (no clamp, :372-381)
            lo, hi = self.bounds[k] if k in self.live else (-3.0e38, 3.0e38)
            d[k] = self.ops.masked_mean_fill(self.p[k], self.mask_u8, out=self.fed[k], lo=lo, hi=hi)
        return d

    def step(self) -> None:
        o, sc = self.ops, self.scene
        lr = self.lr_at(self.t)
        if lr != self._lr:
            self._lr = lr
            self.hyper[0:1].fill_(lr)
        d = self._fed_maps()
        if self.s1 is not None and self.t > 0:                  # the roughness is fixed: the specular sums are constants of the part
            o.shade_fwd_cached(d["albedo"], d["metallic"], self.jac, self.s1, clamp_params=True, out=self.pred)
        else:
            o.shade_fwd(d["albedo"], d["roughness"], d["metallic"], self._n, self._light, self.spp, sc.fov, clamp_params=True, out=self.pred,
                        dcache=self.dcache, jac=self.jac, s1=self.s1)
        if self._bg_mask is not None and not (self.s1 is not None and self.t > 0):
            self.pred.view(-1, 3).index_copy_(0, self._bg_idx, self._bg_rows)
            o.background_into_jac(self.jac, self.s1, self._bg_mask, self._bg_rgb, self._bg_idx, self._bg_rows_t)
        o.brdf_loss_stats(self.pred, self.gt, self.gt_srgb, d["albedo"], d["roughness"], d["metallic"], self.orig["albedo"],
                          self.orig["roughness"], self.orig["metallic"], self.scale_delta, self.stats, self.ws, optimize_part=self.part,
                          es_patience=self.patience, es_min_delta=self.min_delta, history=self.hist)
        o.brdf_loss_bwd_jac(d["albedo"], d["roughness"], d["metallic"], self.jac, self.pred, self.gt_srgb, self.stats,
                            self.orig["albedo"], self.orig["roughness"], self.orig["metallic"], self.scale_delta,
                            self.g["albedo"], self.g["roughness"], self.g["metallic"], self.best["albedo"], self.best["roughness"],
                            self.best["metallic"], self.best_img, optimize_part=self.part)
        for k in ("roughness", "metallic"):
            if k in self.live:                                  # the mean of the masked gradients, through each entry's own clamp (:375-381)
                lo, hi = self.bounds[k]
                o.masked_mean_fill(self.g[k], self.mask_u8, out=self.g[k], lo=lo, hi=hi, gate=self.p[k])
        lib = self._libmod.load()
        with torch.cuda.device(self.gt.device):                 # torch.optim.Adam (:359): decoupled decay 0; rests once EarlyStopping has fired
            self._libmod.check(lib.matpbr_adamw_step_snapshot_dev(o._ptr(self.flat), o._ptr(self.gflat), o._ptr(self.adam_m), o._ptr(self.adam_v),
                                                                  self.flat.numel(), o._ptr(self.hyper), 0.9, 0.999, 1e-8, 0.0, None,
                                                                  o._ptr(self.stats), o._stream(self.flat)), "matpbr_adamw_step_snapshot_dev")
        self.t += 1

    def run(self, n: int) -> None:
        for _ in range(n):
            self.step()

    poll = FusedBrdfPhase.poll

    def history(self) -> torch.Tensor:
        return self.hist[: self.t]

    def current_maps(self) -> Dict[str, torch.Tensor]:
        d = self._fed_maps()
        return {"albedo": d["albedo"].clamp(0, 1), "roughness": d["roughness"].clamp(0.07, 1), "metallic": d["metallic"].clamp(0, 1)}


class MaskedBatchPhase(PipelinedBrdfPhase):
    """`--use_mask` on a batch of images: a batch is its images alone, so it is one `MaskedBrdfPhase` per image (the masked means, SaveBest and
    EarlyStopping are per image), each stepping on a stream of its own -- an image's launches are small next to the chip, and the means between
    them are chains of round trips that run under the other images' renders.  Same interface as `PipelinedBrdfPhase`; same results as the
    images run alone, bit for bit (tests/test_gpu_parity.py)."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, albedo: torch.Tensor, roughness: torch.Tensor, metallic: torch.Tensor,
                 mask: torch.Tensor, **kw):
        if gt_image.ndim != 4 or mask.shape != gt_image.shape[:3]:
            raise ValueError("MaskedBatchPhase: images [B,H,W,3] with their masks [B,H,W]")

        def make(sl: slice, group_kw: dict):
            i = sl.start
            one = dict(group_kw)
            if one.get("originals") is not None:
                one["originals"] = {k: v[0] for k, v in one["originals"].items()}
            return _BatchOfOne(MaskedBrdfPhase(_SceneImage(scene, i), gt_image[i], albedo[i], roughness[i], metallic[i], mask[i], **one))

        super().__init__(scene, gt_image, albedo, roughness, metallic, groups=gt_image.shape[0], make_phase=make, **kw)

    def current_maps(self) -> Dict[str, torch.Tensor]:
        self._join()
        parts = []
        for ph, st in zip(self.phases, self.streams):
            with torch.cuda.stream(st):
                parts.append(ph.current_maps())
        self._join()
        out = {k: torch.cat([d[k] for d in parts], dim=0) for k in parts[0]}
        self._release()
        return out


class FusedEnvPhase:
    """Hot loop A (inverse_img_w_mi.py:236-254).  Materials and normals are fixed while the light is optimised (:216-220) and the
    render is linear in the light, so the phase computes the per-pixel radiance transfer once (`matpbr_shade_transfer`) and
    every iteration is one `matpbr_env_phase_step`: render = T.light, loss statistics, SaveBest / EarlyStopping decisions and
    d loss / d light in a single HBM-bound pass over T.  `head()` produces `emitter.data` ([He,We,3] texels or [25,3] SH
    coefficients) from its own torch parameters (the envmap MLP in the reference, `envmap_net(start_envmap)`, :238-239);
    its backward and the optimiser step stay in torch, fed with d loss / d light from the device."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, head, optimizer: torch.optim.Optimizer, spp: int = 64,
                 patience: int = 0, min_delta: float = 0.0, best_mse: Optional[torch.Tensor] = None, history_len: int = 5000,
                 use_graph: bool = False, keep_pred: bool = True):
        """`use_graph`: capture one whole iteration (head forward, SH projection, the two kernels of matpbr_env_phase_step,
        head backward, optimiser step) into a hipGraph after three eager iterations and replay it afterwards.  The envmap
        MLP works on 512 points, so the eager iteration is bound by kernel launches from Python; the optimiser must then be
        built by `capturable_adam` (tensor learning rate, set with `set_lr`)."""
        import ctypes

        from . import _lib, ops

        self.use_graph, self._graph, self._warm = bool(use_graph), None, 0
        self._ct, self._libmod, self.ops = ctypes, _lib, ops
        self.scene, self.head, self.opt, self.spp = scene, head, optimizer, int(spp)
        self.gt = gt_image.contiguous()
        dev = self.gt.device
        self.B = self.gt.shape[0] if self.gt.ndim == 4 else 1
        self.H, self.W = self.gt.shape[-3], self.gt.shape[-2]
        self.gt_srgb = _loss.linear_to_srgb(self.gt).contiguous()
        self.pred = torch.empty_like(self.gt) if keep_pred else None
        self.best_env: Optional[torch.Tensor] = None
        self.stats = ops.new_loss_stats(self.B, dev)
        if best_mse is not None:
            self.stats[:, ops.STAT_BEST] = best_mse.to(dev).reshape(-1)
        self.lib = _lib.load()
        self.ws = torch.empty(int(self.lib.matpbr_env_phase_workspace_bytes(self.H, self.W, self.B)) // 4 + 1, dtype=torch.float32, device=dev)
        self.hist = torch.zeros((history_len, self.B), dtype=torch.float32, device=dev)
        self.d_light = torch.zeros((self.B, 25, 3) if self.B > 1 else (25, 3), dtype=torch.float32, device=dev)
        self.patience, self.min_delta, self.t = int(patience), float(min_delta), 0
        shp = (self.B, self.H, self.W) if self.B > 1 else (self.H, self.W)
        sc = scene
        self.T = ops.shade_transfer(sc.a.contiguous(), sc.r.reshape(shp + (1,)).contiguous(), sc.m.reshape(shp + (1,)).contiguous(),
                                    sc.shading_normal().contiguous(), self.spp, sc.fov)
        if sc.bg_mask is not None:       # pixels without geometry see the environment along their camera ray: their transfer is the SH basis there
            ops.background_into_transfer(self.T, self.H, self.W, sc.bg_basis)

    def step(self) -> None:
        if not self.use_graph:
            self._body()
        elif self._graph is not None:
            self._graph.replay()
        elif self._warm < 3:                      # eager iterations first: allocator warm-up, lazy initialisations
            self._body()
            self.opt.zero_grad(set_to_none=True)
            self._warm += 1
        else:
            self.opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._body()
            self._graph = graph                   # the capture itself does not execute: replay it for this iteration
            graph.replay()
        self.t += 1

    def _body(self) -> None:
        ct, sc, ops = self._ct, self.scene, self.ops
        data = self.head()
        light = sc.light_from_emitter(data)
        lc = light.detach()
        if self.B > 1 and lc.ndim == 2:
            lc = lc.unsqueeze(0).expand(self.B, -1, -1)
        lc = lc.contiguous()
        P = lambda t: None if t is None else ct.c_void_p(t.data_ptr())
        with torch.cuda.device(self.gt.device), ops._timed("env_phase_step"):
            code = self.lib.matpbr_env_phase_step(P(self.T), P(lc), P(self.gt_srgb), P(self.pred), P(self.d_light), P(self.stats), P(self.hist),
                                                  self.hist.shape[0], self.patience, self.min_delta, P(self.ws), self.ws.numel() * 4, self.H,
                                                  self.W, self.B, ct.c_void_p(torch.cuda.current_stream(self.gt.device).cuda_stream))
        self._libmod.check(code, "matpbr_env_phase_step")
        # SaveBest keeps the envmap of the best iteration (:247): device-side select on the improved flag, no host sync.  The flag
        # is cleared for images whose EarlyStopping fired earlier, and their d_light is zero.
        improved = self.stats[:, ops.STAT_IMPROVED] > 0.5
        d = data.detach()
        if self.best_env is None:
            self.best_env = d.clone()
        else:
            sel = improved.reshape((self.B,) + (1,) * (d.ndim - 1)) if (self.B > 1 and d.ndim == self.best_env.ndim and d.shape[0] == self.B) else improved.any()
            self.best_env.copy_(torch.where(sel, d, self.best_env))
        g = self.d_light if light.shape == self.d_light.shape else self.d_light.sum(0)
        light.backward(g)
        # after an image has stopped its d_light is zero and its `improved` flag stays clear, so best_env keeps the light of the
        # best iteration whatever the optimiser's momentum does to the head until the host polls and leaves the loop
        self.opt.step()
        if not self.use_graph:
            self.opt.zero_grad(set_to_none=True)

    @property
    def best_img(self) -> torch.Tensor:
        """Linear render under the best-so-far light (SaveBest.rendered_img of the env phase, :247): T . light(best_env)."""
        light = self.scene.light_from_emitter(self.best_env).detach()
        if self.B == 1:
            return self.ops.relight(self.T, light.reshape(1, 25, 3), self.H, self.W)[0]
        lights = light if light.ndim == 3 else light.unsqueeze(0).expand(self.B, -1, -1)
        per = self.T.numel() // self.B
        return torch.stack([self.ops.relight(self.T[b * per:(b + 1) * per], lights[b:b + 1].contiguous(), self.H, self.W)[0] for b in range(self.B)])

    def poll(self) -> Dict[str, torch.Tensor]:
        st, o = self.stats.cpu(), self.ops
        return {"stopped": st[:, o.STAT_STOPPED] > 0.5, "iters": st[:, o.STAT_ITERS].to(torch.int64), "best_mse": st[:, o.STAT_BEST],
                "mse": st[:, o.STAT_MSE], "loss": st[:, o.STAT_LOSS]}

    def history(self) -> torch.Tensor:
        return self.hist[: self.t]


class PosMlpBrdfPhase:
    """Hot loop B in the reference's default `pos_mlp` mode (inverse_img_w_mi.py:470-590): the material maps are the output of
    a residual coordinate MLP (`brdf_net(start_arm)`, :493-506); everything downstream of the maps -- render, loss
    statistics, loss backward with regularisers / clamp gating, SaveBest decision -- runs in libmatpbr.so, and the
    gradients w.r.t. the maps are handed back to torch to traverse the MLP.  AdamW + StepLR as in :470-471,553-554."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, net: torch.nn.Module, start_arm: torch.Tensor, fixed: Dict[str, torch.Tensor],
                 optimize_part: str = "arm", spp: int = 64, lr: float = 3e-4, scale_delta: float = 0.1, patience: int = 0,
                 min_delta: float = 0.0, best_mse: Optional[torch.Tensor] = None, history_len: int = 5000,
                 mask: Optional[torch.Tensor] = None):
        from . import ops

        if not scene.use_mesh_normal or "n" in optimize_part:
            raise NotImplementedError("PosMlpBrdfPhase covers output_type 'arm' (geometric normals)")
        self.ops, self.scene, self.net, self.part = ops, scene, net, optimize_part
        self.mask = mask
        self.spp, self.scale_delta = int(spp), float(scale_delta)
        self.gt = gt_image.contiguous()
        self.H, self.W = self.gt.shape[0], self.gt.shape[1]
        self.gt_srgb = _loss.linear_to_srgb(self.gt).contiguous()
        self.start_arm = start_arm.detach()
        # maps that this part does not optimise keep the values they had when the part started (:497-504)
        self.fixed = {k: v.detach().contiguous() for k, v in fixed.items()}
        # regulariser anchors: albedo_ori / roughness_ori / metallic_ori (:189-201)
        self.orig = {"albedo": self.start_arm[:, 0:3].reshape(self.H, self.W, 3).contiguous(),
                     "roughness": (self.start_arm[:, 3:4]).reshape(self.H, self.W, 1).contiguous(),
                     "metallic": self.start_arm[:, 4:5].reshape(self.H, self.W, 1).contiguous()}
        self.opt = _make_adamw(net.parameters(), lr)       # one fused launch instead of eight foreach passes
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=100, gamma=0.8)
        dev = self.gt.device
        self.stats = ops.new_loss_stats(1, dev)
        if best_mse is not None:
            self.stats[:, ops.STAT_BEST] = best_mse.to(dev).reshape(-1)
        self.es = EarlyStopping(patience, min_delta) if patience > 0 else None
        self.pred = torch.empty_like(self.gt)
        self.g = {"albedo": torch.empty(self.H, self.W, 3, device=dev), "roughness": torch.empty(self.H, self.W, 1, device=dev),
                  "metallic": torch.empty(self.H, self.W, 1, device=dev)}
        self.best = {k: v.clone() for k, v in self.fixed.items()}
        self.best_img = torch.zeros_like(self.gt)
        # all parameters as views of one flat buffer: the SaveBest snapshot of the weights is one select instead of one per tensor
        self._names = [k for k, _ in net.named_parameters()]
        params = [p for _, p in net.named_parameters()]
        al = lambda n: (n + 3) // 4 * 4                               # every view 16-byte aligned (the MFMA kernels load float4)
        self._flat = torch.zeros(sum(al(p.numel()) for p in params), dtype=torch.float32, device=dev)
        off = 0
        self._shapes = []
        for p in params:
            view = self._flat[off:off + p.numel()].view_as(p)
            view.copy_(p.detach())
            p.data = view
            self._shapes.append((p.shape, p.numel(), off))
            off += al(p.numel())
        self._best_flat = self._flat.clone()
        self.hist = torch.zeros((history_len, 1), dtype=torch.float32, device=dev)
        self.ws = None
        self.t = 0
        # light and geometric normals are fixed during the phase: cached diffuse-lobe coefficients + jac scratch
        self._n = scene.shading_normal().contiguous()
        self._light = scene.light.detach().contiguous()
        self.dcache = ops.diffuse_cache(self._n, self._light, self.spp, scene.fov)
        self.jac = ops.plane9(self.gt)

    def maps_from_net(self):
        arm = self.net(self.start_arm)                                                   # :493
        H, W = self.H, self.W
        raw = {"albedo": arm[:, 0:3].reshape(H, W, 3), "roughness": (arm[:, 3:4] * 0.93 + 0.07).reshape(H, W, 1),
               "metallic": arm[:, 4:5].reshape(H, W, 1)}                                 # :494-496 (the clamps run inside the kernels)
        keys = {"a": "albedo", "r": "roughness", "m": "metallic"}
        live = [keys[c] for c in self.part if c in keys]
        maps = {k: (raw[k].contiguous() if k in live else self.fixed[k]) for k in raw}
        if self.mask is not None:                                                        # :509-511, after the clamps of :493-495
            for k in ("roughness", "metallic"):                                          # (a map the part does not optimise: as it is)
                maps[k] = masked_mean_fill(maps[k].clamp(0, 1) if k in live else maps[k], self.mask).contiguous()
            live = [k for k in live]                                                     # gradients reach the net through the fill
        return maps, live

    def step(self) -> None:
        ops, sc = self.ops, self.scene
        maps, live = self.maps_from_net()
        d = {k: v.detach() for k, v in maps.items()}
        ops.shade_fwd(d["albedo"], d["roughness"], d["metallic"], self._n, self._light, self.spp, sc.fov, clamp_params=True, out=self.pred,
                      dcache=self.dcache, jac=self.jac)
        if self.ws is None:
            self.ws = torch.empty(int(_lib_ws(1)) // 4, dtype=torch.float32, device=self.gt.device)
        ops.brdf_loss_stats(self.pred, self.gt, self.gt_srgb, d["albedo"], d["roughness"], d["metallic"], self.orig["albedo"],
                            self.orig["roughness"], self.orig["metallic"], self.scale_delta, self.stats, self.ws, optimize_part=self.part)
        ops.brdf_loss_bwd_jac(d["albedo"], d["roughness"], d["metallic"], self.jac, self.pred, self.gt_srgb, self.stats,
                              self.orig["albedo"], self.orig["roughness"], self.orig["metallic"], self.scale_delta,
                              self.g["albedo"], self.g["roughness"], self.g["metallic"], self.best["albedo"], self.best["roughness"],
                              self.best["metallic"], self.best_img, optimize_part=self.part)
        torch.autograd.backward([maps[k] for k in live], [self.g[k] for k in live])      # :544
        improved = self.stats[0, ops.STAT_IMPROVED] > 0.5                                 # SaveBest keeps the weights too (:546-547)
        self._best_flat = torch.where(improved, self._flat, self._best_flat)
        if self.t < self.hist.shape[0]:
            self.hist[self.t].copy_(self.stats[:, ops.STAT_MSE])
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        if self.opt.param_groups[0]["lr"] > 1.5e-4:                                       # :553-554
            self.sched.step()
        self.t += 1

    @property
    def best_weights(self) -> Dict[str, torch.Tensor]:
        return {name: self._best_flat[off:off + n].view(shape).clone() for name, (shape, n, off) in zip(self._names, self._shapes)}

    def step_and_check(self) -> bool:
        """One iteration followed by the host EarlyStopping check of the reference (:550); True when the part should stop."""
        self.step()
        if self.es is None:
            return False
        self.es(float(self.stats[0, self.ops.STAT_MSE]))
        return self.es.early_stop


def pos_mlp_brdf_phase(scene, gt_image, net, start_arm, fixed, optimize_part="arm", mask=None, **kw):
    """The phase object of a `pos_mlp` part: the launch-by-launch `armhead.ArmMlpPhase` where it applies (one image of at least
    8192 pixels, 'arm' network with 256-wide layers; `--use_mask` included), the autograd composition `PosMlpBrdfPhase` otherwise."""
    from .armhead import ArmMlpPhase

    if ArmMlpPhase.supported(scene, gt_image, net, optimize_part, mask):
        return ArmMlpPhase(scene, gt_image, net, start_arm, fixed, optimize_part=optimize_part, mask=mask, **kw)
    return PosMlpBrdfPhase(scene, gt_image, net, start_arm, fixed, optimize_part=optimize_part, mask=mask, **kw)


class PosMlpNormalPhase:
    """Hot loop B in `pos_mlp` mode with output_type 'armn' (inverse_img_w_mi.py:165-172,493-506,516-554): the coordinate MLP
    also predicts the shading normal (`'n'` in --opt_order, predicted normals instead of geometric ones).  Maps from the net
    (clamps of :493-496, `normalize` of :497), the autograd render, the torch-composed loss with the L1 anchor on every live part
    (:522-537), AdamW + StepLR.  On the GPU (no `--use_mask`) the render, the loss statistics, SaveBest's decision and the gradients of the
    maps run on the C ABI (`_step_device`: the launches of `NormalBrdfPhase`); autograd only carries those gradients back through the clamps,
    the normalisation and the MLP (its layers are the HIP kernels of `posmlp._PosMlpHipFn`).  Otherwise: the operator face (`render_w_brdf`)
    and the torch-composed loss."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, net: torch.nn.Module, start_armn: torch.Tensor,
                 fixed: Dict[str, torch.Tensor], optimize_part: str = "armn", spp: int = 64, lr: float = 3e-4, scale_delta: float = 0.1,
                 saver: Optional[DeviceSaveBest] = None, mask: Optional[torch.Tensor] = None):
        self.scene, self.gt, self.net, self.part = scene, gt_image, net, optimize_part
        self.spp, self.scale_delta, self.mask = int(spp), float(scale_delta), mask
        self.H, self.W = gt_image.shape[0], gt_image.shape[1]
        self.gt_srgb = _loss.linear_to_srgb(gt_image)
        self.start = start_armn.detach()
        self.fixed = {k: v.detach() for k, v in fixed.items()}
        H, W = self.H, self.W
        self.armn = self.start.shape[1] >= 8               # 'arm' networks (5 channels) run here too: scenes with a mesh mask
        self.orig = {"albedo": self.start[:, 0:3].reshape(H, W, 3), "roughness": self.start[:, 3:4].reshape(H, W, 1),
                     "metallic": self.start[:, 4:5].reshape(H, W, 1)}
        if self.armn:
            self.orig["normal"] = self.start[:, 5:8].reshape(H, W, 3)
        self.saver = saver if saver is not None else DeviceSaveBest()
        self._t, self.ops = 0, None          # (`ops` and the device buffers: set up by the first `_step_device`)
        # round 6: the network launch by launch on the C ABI (armhead.MlpEngine: forward, backward products, AdamW with SaveBest's weight snapshot --
        # no autograd graph, no framework optimiser); the head (tanh / residual / clamps / normalize) and its backward are element-wise passes here
        self.engine = None
        if self.ENGINE and self.DEVICE_LOSS and self.armn and mask is None and gt_image.is_cuda:
            from .armhead import MlpEngine

            if MlpEngine.why_not(net, self.start.shape[0], gt_image.device) is None:
                self.engine = MlpEngine(net, self.start, lr=lr)
        if self.engine is not None:
            import types

            self.opt, self.sched = types.SimpleNamespace(param_groups=[{"lr": float(lr)}]), None      # what the callers read back
            self._sched_epoch, self._bw = 0, None
        else:
            self.opt = _make_adamw(net.parameters(), lr)
            self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=100, gamma=0.8)
            self._bw = {k: v.detach().clone() for k, v in net.state_dict().items()}

    ENGINE = True           # False: the network through autograd (`posmlp._PosMlpHipFn`) and torch.optim.AdamW

    @property
    def best_weights(self) -> Dict[str, torch.Tensor]:
        """SaveBest's copy of the network's weights (:546-547)."""
        return self.engine.best_weights if self.engine is not None else self._bw

    def maps_from_net(self):
        arm = self.net(self.start)                                                       # :493
        H, W = self.H, self.W
        raw = {"albedo": arm[:, 0:3].clamp(0, 1).reshape(H, W, 3), "roughness": (arm[:, 3:4] * 0.93 + 0.07).clamp(0, 1).reshape(H, W, 1),
               "metallic": arm[:, 4:5].clamp(0, 1).reshape(H, W, 1)}
        if self.armn:
            raw["normal"] = torch.nn.functional.normalize(arm[:, 5:8], p=2, dim=1).reshape(H, W, 3)   # :494-497
        keys = {"a": "albedo", "r": "roughness", "m": "metallic", "n": "normal"}
        live = [keys[c] for c in self.part if c in keys and keys[c] in raw]
        maps = {k: (raw[k] if k in live else self.fixed[k]) for k in raw}
        maps.setdefault("normal", None)
        if self.mask is not None:                                                        # :509-511
            maps["roughness"] = masked_mean_fill(maps["roughness"], self.mask)
            maps["metallic"] = masked_mean_fill(maps["metallic"], self.mask)
        return maps, live

    def _maps_from_engine(self):
        """`maps_from_net` without autograd: the raw outputs of armhead.MlpEngine through the 'armn' head (mymodels/mlps.py:236-244) and the clamps /
        normalisation of inverse_img_w_mi.py:493-497, the same expressions in the same order; what the head's backward needs is kept."""
        H, W = self.H, self.W
        with torch.no_grad():
            x = self.engine.forward_raw()
            t5 = torch.tanh(x[:, 0:5])
            u = 1.3 * t5 + self.start[:, 0:5]
            arm = (u.clamp(0, 1) + u) - u                                                  # x.clamp(0, 1).detach() + x - x.detach()
            nraw = torch.tanh(x[:, 5:8] + self.start[:, 5:8])
            nlen = nraw.norm(dim=1, keepdim=True).clamp_min(1e-12)                         # torch.nn.functional.normalize's eps
            r_pre = arm[:, 3:4] * 0.93 + 0.07
            raw = {"albedo": arm[:, 0:3].clamp(0, 1).reshape(H, W, 3), "roughness": r_pre.clamp(0, 1).reshape(H, W, 1),
                   "metallic": arm[:, 4:5].clamp(0, 1).reshape(H, W, 1), "normal": (nraw / nlen).reshape(H, W, 3)}
            # the clamps of :493-496 pass a gradient where their ARGUMENT lies inside [0, 1] (bounds included): the straight-through value
            # (clamp(u) + u) - u of a saturated u is 1 to rounding -- a hair above it for about half of them, and those entries get none
            gate = torch.cat([(arm[:, 0:3] >= 0) & (arm[:, 0:3] <= 1), (r_pre >= 0) & (r_pre <= 1), (arm[:, 4:5] >= 0) & (arm[:, 4:5] <= 1)], dim=1)
        keys = {"a": "albedo", "r": "roughness", "m": "metallic", "n": "normal"}
        live = [keys[c] for c in self.part if c in keys]
        maps = {k: (raw[k] if k in live else self.fixed[k]) for k in raw}
        self._head = (t5, nraw, nlen, gate)
        return maps, live

    def _engine_backward(self, grads: Dict[str, torch.Tensor], live) -> None:
        """The head's backward (the clamps' gates as recorded by the forward; `normalize`; tanh) -> d loss / d (raw
        outputs) -> the network's backward products (armhead.MlpEngine.backward_raw)."""
        t5, nraw, nlen, gate = self._head
        M = t5.shape[0]
        with torch.no_grad():
            d_arm = torch.zeros(M, 5, device=t5.device)
            if "albedo" in live:
                d_arm[:, 0:3] = grads["albedo"].reshape(M, 3)
            if "roughness" in live:
                d_arm[:, 3:4] = grads["roughness"].reshape(M, 1) * 0.93
            if "metallic" in live:
                d_arm[:, 4:5] = grads["metallic"].reshape(M, 1)
            d_x = torch.zeros(M, 8, device=t5.device)
            d_x[:, 0:5] = torch.where(gate, d_arm, torch.zeros_like(d_arm)) * (1.3 * (1.0 - t5 * t5))
            if "normal" in live:
                g = grads["normal"].reshape(M, 3)
                y = nraw / nlen
                d_x[:, 5:8] = ((g - y * (y * g).sum(dim=1, keepdim=True)) / nlen) * (1.0 - nraw * nraw)
        self.engine.backward_raw(d_x)

    DEVICE_LOSS = True      # False: the autograd render and the torch-composed loss in every case

    def _device_setup(self) -> None:
        from . import ops

        dev, H, W = self.gt.device, self.H, self.W
        self.ops = ops
        self.stats = ops.new_loss_stats(1, dev)
        self.ws = torch.empty(int(_lib_ws(1)) // 4, dtype=torch.float32, device=dev)
        self.pred, self.d_pred = torch.empty_like(self.gt), torch.empty_like(self.gt)
        self.jac = ops.plane9(self.gt)
        self._light = self.scene.light.detach().contiguous()
        E = lambda c: torch.empty(H, W, c, device=dev)
        self.g = {"albedo": E(3), "roughness": E(1), "metallic": E(1)}
        # the snapshot buffers start as the incoming fixed maps (what a reader of `saver.best` sees until an iteration improves on the carried-over
        # best loss: never uninitialised memory)
        start = {"albedo": self.fixed["albedo"], "roughness": self.fixed["roughness"], "metallic": self.fixed["metallic"],
                 "normal": self.fixed.get("normal", self.scene.shading_normal())}
        self.best = {k: v.detach().to(dev, torch.float32).reshape(H, W, -1).clone() for k, v in start.items()}
        self.best["rendered_img"] = torch.zeros_like(self.gt)
        self.saver.best = self.best        # the runner reads the snapshot through the saver

    def _step_device(self) -> torch.Tensor:
        """The iteration with the render, the loss, SaveBest's decision and the gradients of the maps on the C ABI (no autograd render node,
        no framework losses): the maps the net produced, detached -> render with its nine planes -> statistics -> material gradients with their
        regularisers and the snapshot (matpbr_brdf_loss_bwd_jac) -> d loss / d pred -> the normal gradient (matpbr_shade_bwd) + its L1 anchor;
        autograd carries those gradients back through the clamps, the normalisation and the MLP."""
        if self.ops is None:
            self._device_setup()
        o, sc = self.ops, self.scene
        if self.saver.best_loss is not None and self._t == 0:
            self.stats[:, o.STAT_BEST] = self.saver.best_loss.to(self.gt.device).reshape(-1)   # SaveBest.best_loss is global across phases (F11)
        maps, live = self._maps_from_engine() if self.engine is not None else self.maps_from_net()
        d = {k: maps[k].detach().contiguous() for k in ("albedo", "roughness", "metallic", "normal")}
        o.shade_fwd(d["albedo"], d["roughness"], d["metallic"], d["normal"], self._light, self.spp, sc.fov, out=self.pred, jac=self.jac)
        og = self.orig
        o.brdf_loss_stats(self.pred, self.gt, self.gt_srgb, d["albedo"], d["roughness"], d["metallic"], og["albedo"].contiguous(),
                          og["roughness"].contiguous(), og["metallic"].contiguous(), self.scale_delta, self.stats, self.ws, optimize_part=self.part)
        o.brdf_loss_bwd_jac(d["albedo"], d["roughness"], d["metallic"], self.jac, self.pred, self.gt_srgb, self.stats, og["albedo"].contiguous(),
                            og["roughness"].contiguous(), og["metallic"].contiguous(), self.scale_delta, self.g["albedo"], self.g["roughness"],
                            self.g["metallic"], self.best["albedo"], self.best["roughness"], self.best["metallic"], self.best["rendered_img"],
                            optimize_part=self.part)
        grads = {k: self.g[k] for k in ("albedo", "roughness", "metallic")}
        improved = self.stats[0, o.STAT_IMPROVED] > 0.5
        if "normal" in live:
            o.brdf_loss_dpred(self.pred, self.gt_srgb, self.stats, self.d_pred)
            g_n = o.shade_bwd(d["albedo"], d["roughness"], d["metallic"], d["normal"], self._light, self.d_pred, self.spp, sc.fov, want_mat=False,
                              want_n=True)[3]
            g_n.add_(torch.sign(d["normal"] - og["normal"]), alpha=self.scale_delta / (3.0 * self.H * self.W))      # L1(normal, normal_ori), :533-535
            grads["normal"] = g_n
        self.best["normal"] = torch.where(improved, d["normal"], self.best["normal"])
        self.saver.best = self.best
        if self.engine is not None:
            self._engine_backward(grads, live)                                           # :544
            self.saver.best_loss = self.stats[0, o.STAT_BEST].clone()
            self.engine.adamw_step(self.stats)                                           # AdamW + SaveBest's copy of the weights (:546-547) in one launch
            lr_now = self.opt.param_groups[0]["lr"]
            if lr_now > 1.5e-4:                                                          # StepLR(100, 0.8), stepped while lr > 1.5e-4 (:553-554)
                self._sched_epoch += 1
                if self._sched_epoch % 100 == 0:
                    self.engine.set_lr(lr_now * 0.8)
                    self.opt.param_groups[0]["lr"] = lr_now * 0.8
            self._t += 1
            return self.stats[0, o.STAT_MSE].clone()
        torch.autograd.backward([maps[k] for k in live], [grads[k].reshape(maps[k].shape) for k in live])      # :544
        for k, v in self.net.state_dict().items():                                       # SaveBest keeps the weights too (:546-547)
            self._bw[k] = torch.where(improved, v.detach(), self._bw[k])
        self.saver.best_loss = self.stats[0, o.STAT_BEST].clone()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        if self.opt.param_groups[0]["lr"] > 1.5e-4:                                      # :553-554
            self.sched.step()
        self._t += 1
        return self.stats[0, o.STAT_MSE].clone()

    def step(self) -> torch.Tensor:
        if self.DEVICE_LOSS and self.armn and self.mask is None and self.gt.is_cuda:
            return self._step_device()
        maps, live = self.maps_from_net()
        pred = _render.render_w_brdf(self.scene, maps["albedo"], maps["roughness"], maps["metallic"], maps["normal"], self.spp)   # :515
        loss, loss_mse, pred_srgb, _ = _loss.brdf_loss(pred, self.gt, {k: maps[k] for k in live}, self.orig, self.scale_delta, self.gt_srgb)
        loss.backward()                                                                  # :544
        before = self.saver.best_loss.clone() if self.saver.best_loss is not None else torch.full_like(loss_mse.detach(), float("inf"))
        extra = {"normal": maps["normal"]} if maps["normal"] is not None else {}
        self.saver.update(loss_mse, albedo=maps["albedo"], roughness=maps["roughness"], metallic=maps["metallic"], rendered_img=pred_srgb, **extra)
        flag = (self.saver.best_loss < before).reshape(())
        for k, v in self.net.state_dict().items():                                       # SaveBest keeps the weights too (:546-547)
            self._bw[k] = torch.where(flag, v.detach(), self._bw[k])
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        if self.opt.param_groups[0]["lr"] > 1.5e-4:                                      # :553-554
            self.sched.step()
        return loss_mse.detach()


def _lib_ws(batch: int) -> int:
    from . import _lib

    return _lib.load().matpbr_brdf_loss_workspace_bytes(int(batch))


def capturable_adam(params, lr: float) -> torch.optim.Optimizer:
    """Adam whose step can live inside a hipGraph: device-side step counter and a tensor learning rate (see `set_lr`)."""
    params = list(params)
    dev = params[0].device
    return torch.optim.Adam(params, lr=torch.tensor(float(lr), device=dev), capturable=True, foreach=True)


def set_lr(opt: torch.optim.Optimizer, lr: float) -> None:
    for gp in opt.param_groups:
        if isinstance(gp["lr"], torch.Tensor):
            gp["lr"].fill_(float(lr))
        else:
            gp["lr"] = float(lr)
