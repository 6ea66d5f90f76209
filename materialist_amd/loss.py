"""Losses of the optimisation loop, restated from inverse_img_w_mi.py (a11 of SURVEY.md section 8a).

env phase  (:241-245):  loss = MSE + L1 on x^(1/2.2)
BRDF phase (:388-418, 516-542): pred *= mean(gt)/mean(pred).detach();
                                loss = 3*(L1/MSE).detach()*MSE + L1 + scale_delta * sum L1(part - original)
`linear_to_srgb` is the pure gamma 2.2 of myutils/misc.py:167-170.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

_EPS = 1e-8  # the deterministic render can return exact zeros (back-facing pixels); x^(1/2.2) has no gradient there


def linear_to_srgb(image: torch.Tensor) -> torch.Tensor:
    return image ** (1.0 / 2.2)


def srgb_to_linear(image: torch.Tensor) -> torch.Tensor:
    return image ** 2.2


def _mse(x, y):
    """Per-image mean: scalar for [H,W,C], [B] for a batch [B,H,W,C] (images are independent problems)."""
    d = (x - y) ** 2
    return d.mean() if d.ndim < 4 else d.mean(dim=(1, 2, 3))


def _l1(x, y):
    d = (x - y).abs()
    return d.mean() if d.ndim < 4 else d.mean(dim=(1, 2, 3))


def _img_mean(x):
    return x.mean() if x.ndim < 4 else x.mean(dim=(1, 2, 3), keepdim=True)


def env_loss(pred_image: torch.Tensor, gt_image: torch.Tensor):
    """Returns (loss, loss_mse, loss_l1): inverse_img_w_mi.py:241-245.  For a batch the per-image losses are
    summed into `loss` (independent images, independent gradients) and returned per image in loss_mse/l1."""
    pred_srgb = linear_to_srgb(pred_image.clamp_min(_EPS))
    gt_srgb = linear_to_srgb(gt_image)
    loss_mse = _mse(pred_srgb, gt_srgb)
    loss_l1 = _l1(pred_srgb, gt_srgb)
    return (loss_mse + loss_l1).sum(), loss_mse, loss_l1


FUSED = True      # brdf_loss of CUDA tensors as ONE autograd node on the C ABI (two statistics launches forward, one launch backward); False: torch ops
_fused_cache: Dict[tuple, tuple] = {}


class _BrdfLossFn(torch.autograd.Function):
    """inverse_img_w_mi.py:388-418 as one node: forward = matpbr_brdf_loss_stats (exposure ratio, MSE and L1 on the gamma-2.2 images, the
    regularisers' L1, the loss), backward = matpbr_brdf_loss_dpred (d loss / d pred from those statistics) and sign(map - anchor) for the
    regularised maps.  The torch composition runs six reductions over the image (two means for the ratio, MSE, L1, one L1 per map)."""

    @staticmethod
    def forward(ctx, pred, gt, gt_srgb, scale_delta, part, stats, ws, pa, pr, pm, a0, r0, m0):
        from . import ops

        pred = pred.contiguous()
        ops.brdf_loss_stats(pred, gt, gt_srgb, pa, pr, pm, a0, r0, m0, scale_delta, stats, ws, optimize_part=part)
        ctx.save_for_backward(pred, gt_srgb, stats.clone(), pa, pr, pm, a0, r0, m0)
        ctx.scale_delta, ctx.part = float(scale_delta), part
        single = pred.ndim == 3
        mse = stats[0, ops.STAT_MSE].clone() if single else stats[:, ops.STAT_MSE].clone()
        ratio = stats[0, ops.STAT_RATIO].clone() if single else stats[:, ops.STAT_RATIO].clone().reshape(-1, 1, 1, 1)
        ctx.mark_non_differentiable(mse, ratio)
        return stats[:, ops.STAT_LOSS].sum(), mse, ratio

    @staticmethod
    def backward(ctx, g_loss, _g_mse, _g_ratio):
        from . import ops

        pred, gt_srgb, stats, pa, pr, pm, a0, r0, m0 = ctx.saved_tensors
        d_pred = ops.brdf_loss_dpred(pred, gt_srgb, stats, torch.empty_like(pred)) if ctx.needs_input_grad[0] else None
        if d_pred is not None:
            d_pred.mul_(g_loss)
        grads = []
        for i, (key, p_, p0) in enumerate((("a", pa, a0), ("r", pr, r0), ("m", pm, m0))):
            if key in ctx.part and ctx.needs_input_grad[7 + i]:
                n = p_.numel() // (1 if pred.ndim == 3 else pred.shape[0])                     # elements per image (the L1 is a per-image mean)
                grads.append(torch.sign(p_ - p0).mul_(g_loss * (ctx.scale_delta / n)))        # d (scale_delta L1(map, anchor)) / d map, per image
            else:
                grads.append(None)
        return (d_pred, None, None, None, None, None, None, *grads, None, None, None)


def _brdf_loss_fused(pred_image, gt_image, parts, originals, scale_delta, gt_srgb):
    from . import ops

    dev, single = pred_image.device, pred_image.ndim == 3
    B = 1 if single else pred_image.shape[0]
    key = (dev, B, tuple(pred_image.shape))
    if key not in _fused_cache:
        shp1 = tuple(pred_image.shape[:-1]) + (1,)
        _fused_cache[key] = (ops.new_loss_stats(B, dev), torch.empty(int(ops._lib.load().matpbr_brdf_loss_workspace_bytes(B)) // 4, dtype=torch.float32, device=dev),
                             torch.zeros(shp1, dtype=torch.float32, device=dev))
    stats, ws, zero1 = _fused_cache[key]
    part = "".join(c for c, k in (("a", "albedo"), ("r", "roughness"), ("m", "metallic")) if k in parts)
    if gt_srgb is None:
        gt_srgb = linear_to_srgb(gt_image)
    pick = lambda k, dummy: (parts[k].contiguous(), originals[k].contiguous()) if k in parts else (dummy, dummy)
    (pa, a0), (pr, r0), (pm, m0) = pick("albedo", gt_image), pick("roughness", zero1), pick("metallic", zero1)
    loss, mse, ratio = _BrdfLossFn.apply(pred_image, gt_image.contiguous(), gt_srgb.contiguous(), float(scale_delta), part, stats, ws, pa, pr, pm, a0, r0, m0)
    with torch.no_grad():
        pred_srgb = linear_to_srgb((pred_image.detach() * ratio).clamp_min_(_EPS))
    return loss, mse, pred_srgb, ratio


def brdf_loss(pred_image: torch.Tensor, gt_image: torch.Tensor, parts: Dict[str, torch.Tensor], originals: Dict[str, torch.Tensor],
              scale_delta: float = 0.1, gt_srgb: Optional[torch.Tensor] = None):
    """Returns (loss, loss_mse, pred_srgb, ratio): inverse_img_w_mi.py:388-418 / 516-542.
    `parts` / `originals` hold the maps being optimised in this phase ('albedo', 'roughness', 'metallic', 'normal').
    CUDA tensors and material maps only: one autograd node on the C ABI (`_BrdfLossFn`; `FUSED = False` restores the torch composition)."""
    if FUSED and pred_image.is_cuda and parts and all(k in ("albedo", "roughness", "metallic") for k in parts):
        return _brdf_loss_fused(pred_image, gt_image, parts, originals, scale_delta, gt_srgb)
    ratio = _img_mean(gt_image) / _img_mean(pred_image.detach())
    pred_image = pred_image * ratio
    pred_srgb = linear_to_srgb(pred_image.clamp_min(_EPS))
    if gt_srgb is None:
        gt_srgb = linear_to_srgb(gt_image)
    loss_mse = _mse(pred_srgb, gt_srgb)
    loss_l1 = _l1(pred_srgb, gt_srgb)
    aux = 0
    for key, val in parts.items():
        aux = aux + _l1(val, originals[key])
    scale_ratio = loss_l1.detach() / loss_mse.detach()
    render_loss = 3 * scale_ratio * loss_mse + loss_l1
    return (render_loss + aux * scale_delta).sum(), loss_mse, pred_srgb, ratio


def psnr(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """PSNR (dB, peak 1) of linear-RGB images clipped to [0,1] after gamma 2.2."""
    p = linear_to_srgb(pred.clamp(0, 1))
    g = linear_to_srgb(gt.clamp(0, 1))
    return -10.0 * torch.log10(_mse(p, g).clamp_min(1e-20))


def tensors_digest(*tensors) -> str:
    """SHA-256 (16 hex digits) of the bytes of the given tensors, in order: the stage digests of optimize.optimize_envmap_ARMN."""
    import hashlib

    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]
