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


def brdf_loss(pred_image: torch.Tensor, gt_image: torch.Tensor, parts: Dict[str, torch.Tensor], originals: Dict[str, torch.Tensor],
              scale_delta: float = 0.1, gt_srgb: Optional[torch.Tensor] = None):
    """Returns (loss, loss_mse, pred_srgb, ratio): inverse_img_w_mi.py:388-418 / 516-542.
    `parts` / `originals` hold the maps being optimised in this phase ('albedo', 'roughness', 'metallic', 'normal')."""
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
