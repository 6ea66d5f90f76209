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


def _split_k_tn(g: torch.Tensor, x: torch.Tensor, chunks: int = 64) -> torch.Tensor:
    """g.T @ x for g [M, n], x [M, k] with M >> n, k: the reduction dimension is split into `chunks` batched GEMMs whose
    partial products are then added -- a plain [n, M] x [M, k] call lands on a 32x64 macro-tile kernel at 40 TFLOP/s."""
    M = g.shape[0]
    if M % chunks or M < 64 * chunks:
        return g.t() @ x
    return torch.bmm(g.view(chunks, M // chunks, -1).transpose(1, 2), x.view(chunks, M // chunks, -1)).sum(0)


def _column_sum(g: torch.Tensor) -> torch.Tensor:
    if g.is_cuda and g.dtype == torch.float32:
        from . import ops

        return ops.column_sum(g)
    return g.sum(0)


class _LinearSin(torch.autograd.Function):
    """y = sin(x @ W[:, :k].T + x0 @ W[:, k:].T + b) with a backward pass made only of GEMMs / GEMVs.

    Two things PyTorch's stock composition does badly at M = 262 144 rows: the bias gradient is a column reduction of a
    [M, 256] tensor (`reduce_kernel`, 2-4 ms each on MI355X, 45 % of an iteration) -- here it is a GEMV with a ones vector;
    and the skip connection materialises cat(x, x0) (a 268 MB copy per skip layer, forward and backward) -- here the weight
    is split by columns instead, so x and x0 are multiplied separately into the same output."""

    @staticmethod
    def forward(ctx, x, x0, weight, bias, apply_sin):
        k = x.shape[1]
        pre = torch.addmm(bias, x, weight[:, :k].t())
        if x0 is not None:
            pre.addmm_(x0, weight[:, k:].t())
        ctx.save_for_backward(x, x0, weight, pre if apply_sin else None)
        ctx.apply_sin = apply_sin
        return torch.sin(pre) if apply_sin else pre

    @staticmethod
    def backward(ctx, grad):
        x, x0, weight, pre = ctx.saved_tensors
        k = x.shape[1]
        g = grad * torch.cos(pre) if ctx.apply_sin else grad
        g = g.contiguous()
        d_w, d_x0 = _split_k_tn(g, x), None
        if x0 is not None:
            d_w = torch.cat([d_w, _split_k_tn(g, x0)], dim=1)        # [out, k + len(x0)]: small
            d_x0 = g @ weight[:, k:] if ctx.needs_input_grad[1] else None
        d_b = _column_sum(g)
        d_x = g @ weight[:, :k] if ctx.needs_input_grad[0] else None
        return d_x, d_x0, d_w, d_b, None


class _Sine(nn.Module):
    def __init__(self, n_in: int, n_out: int):
        super().__init__()
        self.linear = nn.Linear(n_in, n_out)      # default PyTorch init; the SIREN init is commented out in the reference (:86)

    def forward(self, x: torch.Tensor, x0=None) -> torch.Tensor:
        return _LinearSin.apply(x, x0, self.linear.weight, self.linear.bias, True)


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
            layer = getattr(self, f"lin{l}")
            skip_in = x0 if l in self.skip else None          # cat(x, x0) without the copy: the weight is split by columns
            if l < self.n_layers - 1:
                x = layer(x, skip_in)
            else:
                x = _LinearSin.apply(x, skip_in, layer.weight, layer.bias, False)
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
