"""PosMLP: the coordinate MLPs of the reference (mymodels/mlps.py:129-251) on PyTorch-ROCm.

Structure (App. B of SURVEY.md), restated rather than copied:
  * points = every pixel of the [h, w] grid implied by the row count (N > 512 rows: square sqrt(N); else h = sqrt(N/2), w = 2h);
  * positional code of the integer (row, col) with `multires` octaves: [p, sin(p), cos(p), sin(2p), cos(2p), ...] (:8-54);
  * x0 = cat(code, img); hidden layers are Linear followed by sin (the reference's SineLayer applies no omega, :102-103);
    the layers listed in `skip` take cat(x, x0) and the layer before each of them is narrowed by len(x0) (:159-164,223-224);
  * last layer: zero-initialised Linear (:174-176);
  * heads: 'envmap' softplus; 'arm' 1.3 tanh(y) + img with a straight-through clamp to [0,1]; 'armn' the same on the first five
    channels and tanh(y + img) on the last three; 'normal' normalize(tanh(y + img)) (:230-251).
Parameter names match the reference (`lin{l}.linear.{weight,bias}`, last `lin{L}.{weight,bias}`) so its state_dict loads
unchanged.  Unlike the reference there is no `torch.isnan(...).any()` host synchronisation per layer (:218-229); call
`check_finite()` when a check is wanted.  The GEMMs (M = 262 144, K = N = 256 at 512x512) run on hipBLASLt / MFMA.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Sine(nn.Module):
    def __init__(self, n_in: int, n_out: int):
        super().__init__()
        self.linear = nn.Linear(n_in, n_out)      # default PyTorch init; the SIREN init is commented out in the reference (:86)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.sin(self.linear(x))


def grid_shape(n_rows: int) -> Tuple[int, int]:
    """mymodels/mlps.py:190-198."""
    if n_rows > 512:
        h = math.isqrt(n_rows)
        if h * h != n_rows:
            raise ValueError("the point set must be a square image")
        return h, h
    h = math.isqrt(n_rows // 2)
    if 2 * h * h != n_rows:
        raise ValueError("width should be double of height")
    return h, 2 * h


def positional_code(h: int, w: int, multires: int, device, dtype) -> torch.Tensor:
    """[h*w, 2 + 4*multires] (or [h*w, 2] when multires == 0): integer pixel coordinates, not normalised (:199-205)."""
    rows, cols = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    p = torch.stack([rows.flatten(), cols.flatten()], dim=1).to(dtype)
    feats = [p]
    for k in range(multires):
        f = 2.0 ** k                              # log-sampled bands 2^0 .. 2^(multires-1) (:25-28,42-50)
        feats += [torch.sin(p * f), torch.cos(p * f)]
    return torch.cat(feats, dim=1)


class PosMLP(nn.Module):
    def __init__(self, color_ch: int, out_dims: int, hidden: Sequence[int] = (256, 256, 256, 256), skip: Sequence[int] = (1, 3),
                 multires_view: int = 2, output_type: str = "envmap"):
        super().__init__()
        if output_type not in ("envmap", "arm", "armn", "normal"):
            raise ValueError("output_type should be envmap or arm or armn")
        self.output_type, self.multires, self.skip, self.color_ch = output_type, int(multires_view), tuple(skip), int(color_ch)
        d0 = 2 + 4 * self.multires + color_ch if self.multires > 0 else 2 + color_ch
        dims = [d0] + list(hidden) + [out_dims]
        self.n_layers = len(dims) - 1
        for l in range(self.n_layers):
            n_out = dims[l + 1] - d0 if (l + 1) in self.skip else dims[l + 1]
            if l < self.n_layers - 1:
                layer = _Sine(dims[l], n_out)
            else:
                layer = nn.Linear(dims[l], n_out)
                nn.init.zeros_(layer.weight)
                nn.init.zeros_(layer.bias)
            setattr(self, f"lin{l}", layer)
        self._code: Dict[tuple, torch.Tensor] = {}

    def _points(self, img: torch.Tensor) -> torch.Tensor:
        h, w = grid_shape(img.shape[0])
        key = (h, w, img.device, img.dtype)
        if key not in self._code:
            self._code[key] = positional_code(h, w, self.multires, img.device, img.dtype)
        return torch.cat([self._code[key], img], dim=1)

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        x0 = self._points(img)
        x = x0
        for l in range(self.n_layers):
            if l in self.skip:
                x = torch.cat([x, x0], dim=-1)
            x = getattr(self, f"lin{l}")(x)
        if self.output_type == "envmap":
            return F.softplus(x)
        if self.output_type == "arm":
            y = 1.3 * torch.tanh(x) + img
            return y.clamp(0, 1).detach() + y - y.detach()
        if self.output_type == "armn":
            y = 1.3 * torch.tanh(x[..., 0:5]) + img[..., 0:5]
            y = y.clamp(0, 1).detach() + y - y.detach()
            return torch.cat([y, torch.tanh(x[..., 5:8] + img[..., 5:8])], dim=-1)
        return F.normalize(torch.tanh(x + img), p=2, dim=-1)

    def check_finite(self) -> None:
        for name, p in self.named_parameters():
            if not torch.isfinite(p).all():
                raise ValueError(f"nan value in {name}")


def envmap_net(**kw) -> PosMLP:
    """PosMLP(in_dims=5, out_dims=3, ..., multires_view=2, output_type='envmap', color_ch=3) (inverse_img_w_mi.py:117-124)."""
    return PosMLP(color_ch=3, out_dims=3, multires_view=2, output_type="envmap", **kw)


def brdf_net(output_type: str = "arm", **kw) -> PosMLP:
    """'arm': in 7 -> 15 inputs, 5 outputs, 2 octaves; 'armn': 10 inputs (raw coordinates), 8 outputs (inverse_img_w_mi.py:159-172)."""
    if output_type == "arm":
        return PosMLP(color_ch=5, out_dims=5, multires_view=2, output_type="arm", **kw)
    return PosMLP(color_ch=8, out_dims=8, multires_view=0, output_type="armn", **kw)
