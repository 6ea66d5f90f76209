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

import copy
from typing import Dict, Optional

import torch

from . import loss as _loss
from . import render as _render


class EarlyStopping:
    """myutils/misc.py:37-60: relative-improvement patience; `min_delta` is a fraction of the best loss."""

    def __init__(self, patience: int = 10, min_delta: float = 0.0):
        self.patience = patience
        self.min_delta = min_delta
        self.counter = 0
        self.best_loss = None
        self.early_stop = False

    def __call__(self, val_loss: float) -> None:
        if self.best_loss is None:
            self.best_loss = val_loss
        elif val_loss > self.best_loss * (1 - self.min_delta):
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            self.best_loss = val_loss
            self.counter = 0


class SaveBest:
    """myutils/misc.py:62-97: keep a detached clone of every map whenever `loss` is strictly below the best
    seen so far; the best loss is global across phases and never reset (F11)."""

    FIELDS = ("albedo", "roughness", "metallic", "normal", "envmap", "rendered_img")

    def __init__(self):
        self.best_loss = float("inf")
        self.best_albedo = self.best_roughness = self.best_metallic = None
        self.best_envmap = self.rendered_img = self.best_normal = None
        self.best_brdfnet_weight = None

    @staticmethod
    def _detach_and_clone(t):
        return t.detach().clone() if isinstance(t, torch.Tensor) else copy.deepcopy(t)

    def update(self, loss, albedo, roughness, metallic, normal, envmap, rendered_img, brdfnet_weights=None) -> bool:
        if loss < self.best_loss:
            self.best_loss = loss
            self.best_albedo = self._detach_and_clone(albedo)
            self.best_roughness = self._detach_and_clone(roughness)
            self.best_metallic = self._detach_and_clone(metallic)
            self.best_envmap = self._detach_and_clone(envmap)
            self.rendered_img = self._detach_and_clone(rendered_img)
            self.best_normal = self._detach_and_clone(normal)
            if brdfnet_weights is not None:
                self.best_brdfnet_weight = copy.deepcopy(brdfnet_weights)
            return True
        return False

    def get_best(self):
        return {"envmap": self.best_envmap, "albedo": self.best_albedo, "roughness": self.best_roughness,
                "metallic": self.best_metallic, "normal": self.best_normal, "rendered_img": self.rendered_img}


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


class BrdfPhase:
    """Hot loop B, `model_name == 'none'` (inverse_img_w_mi.py:347-468)."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, albedo: torch.Tensor, roughness: torch.Tensor, metallic: torch.Tensor,
                 normal: Optional[torch.Tensor] = None, optimize_part: str = "arm", spp: int = 64, lr: float = 3e-4,
                 scale_delta: float = 0.1, saver: Optional[DeviceSaveBest] = None):
        self.scene, self.gt, self.spp, self.scale_delta = scene, gt_image, spp, scale_delta
        self.part = optimize_part
        self.maps = {"albedo": albedo, "roughness": roughness, "metallic": metallic, "normal": normal}
        self.originals = {k: v.detach().clone() for k, v in self.maps.items() if v is not None}
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
        return m

    def step(self) -> torch.Tensor:
        m = self.current_maps()
        normal = None if self.scene.use_mesh_normal else m["normal"]
        pred = _render.render_w_brdf(self.scene, m["albedo"], m["roughness"], m["metallic"], normal, self.spp)   # :384-386
        parts = {k: m[k] for k in self.opt_keys}
        loss, loss_mse, pred_srgb, _ = _loss.brdf_loss(pred, self.gt, parts, self.originals, self.scale_delta, self.gt_srgb)
        loss.backward()                                                                                          # :420
        self.saver.update(loss_mse, albedo=m["albedo"], roughness=m["roughness"], metallic=m["metallic"], rendered_img=pred_srgb)
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


class FusedBrdfPhase:
    """Hot loop B in `model_name == 'none'` mode (inverse_img_w_mi.py:347-468) with every step between the parameter
    maps and their Adam update executed by libmatpbr.so: render (clamp folded in), loss statistics, fused loss backward
    with regularisers / clamp gating / SaveBest snapshot, Adam.  Same arithmetic as `BrdfPhase` (which composes the same
    step from torch ops and serves as its parity reference); nothing returns to the host inside a step.

    Early stopping needs the MSE on the host.  `history()` returns the per-iteration MSE recorded on the device, so
    the caller can replay EarlyStopping every `k` iterations instead of synchronising on each one."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, albedo: torch.Tensor, roughness: torch.Tensor, metallic: torch.Tensor,
                 spp: int = 64, lr: float = 3e-4, scale_delta: float = 0.1, stats: Optional[torch.Tensor] = None, history_len: int = 5000):
        from . import ops

        if not scene.use_mesh_normal:
            raise NotImplementedError("FusedBrdfPhase optimises a/r/m under the geometric normal; use BrdfPhase for 'n'")
        self.ops, self.scene, self.spp, self.scale_delta = ops, scene, int(spp), float(scale_delta)
        self.gt = gt_image.contiguous()
        self.gt_srgb = _loss.linear_to_srgb(self.gt).contiguous()
        c = lambda t: t.detach().clone().contiguous()
        self.p = {"albedo": c(albedo), "roughness": c(roughness), "metallic": c(metallic)}
        self.orig = {k: c(v) for k, v in self.p.items()}
        self.g = {k: torch.empty_like(v) for k, v in self.p.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.best = {k: c(v) for k, v in self.p.items()}
        self.best_img = torch.zeros_like(self.gt)
        self.pred = torch.empty_like(self.gt)
        B = self.gt.shape[0] if self.gt.ndim == 4 else 1
        self.stats = stats if stats is not None else ops.new_loss_stats(B, self.gt.device)
        self.ws = torch.empty(int(_lib_ws(B)) // 4, dtype=torch.float32, device=self.gt.device)
        self.hist = torch.zeros((history_len, B), dtype=torch.float32, device=self.gt.device)
        self.lr, self.t = float(lr), 0

    def step(self) -> None:
        ops, sc, p = self.ops, self.scene, self.p
        n, light = sc.shading_normal(), sc.light
        B = self.stats.shape[0]
        if B > 1 and light.ndim == 2:
            light = light.unsqueeze(0).expand(B, -1, -1).contiguous()
        ops.shade_fwd(p["albedo"], p["roughness"], p["metallic"], n, light, self.spp, sc.fov, clamp_params=True, out=self.pred)
        ops.brdf_loss_stats(self.pred, self.gt, self.gt_srgb, p["albedo"], p["roughness"], p["metallic"], self.orig["albedo"],
                            self.orig["roughness"], self.orig["metallic"], self.scale_delta, self.stats, self.ws)
        ops.shade_bwd_brdf_loss(p["albedo"], p["roughness"], p["metallic"], n, light, self.pred, self.gt_srgb, self.stats,
                                self.orig["albedo"], self.orig["roughness"], self.orig["metallic"], self.scale_delta, self.spp,
                                self.g["albedo"], self.g["roughness"], self.g["metallic"], self.best["albedo"], self.best["roughness"],
                                self.best["metallic"], self.best_img, sc.fov)
        if self.t < self.hist.shape[0]:
            self.hist[self.t].copy_(self.stats[:, ops.STAT_MSE])
        self.t += 1
        for k in p:
            ops.adam_step(p[k], self.g[k], self.m[k], self.v[k], self.lr, self.t)
        if self.lr > 1.5e-4 and self.t % 100 == 0:      # StepLR(100, 0.8) stepped while lr > 1.5e-4 (:363-365,431-432)
            self.lr *= 0.8

    def history(self) -> torch.Tensor:
        """[iterations so far, B] loss_mse of every iteration (device tensor)."""
        return self.hist[: self.t]

    def current_maps(self) -> Dict[str, torch.Tensor]:
        return {"albedo": self.p["albedo"].clamp(0, 1), "roughness": self.p["roughness"].clamp(0.07, 1), "metallic": self.p["metallic"].clamp(0, 1)}


def _lib_ws(batch: int) -> int:
    from . import _lib

    return _lib.load().matpbr_brdf_loss_workspace_bytes(int(batch))
