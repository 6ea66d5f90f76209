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


def _sin_bwd(d_y: torch.Tensor, pre: torch.Tensor) -> torch.Tensor:
    if d_y.is_cuda and d_y.dtype == torch.float32:
        from . import ops

        return ops.sin_bwd(d_y, pre)
    return d_y * torch.cos(pre)


class _PosMlpFn(torch.autograd.Function):
    """The whole coordinate MLP as one autograd node, laid out for the hardware:

    * a layer that feeds a skip layer writes sin(pre) straight into the first columns of the next layer's [M, 256] input
      buffer whose last columns already hold x0 -- `cat(x, x0)` (:223-224) costs no copy and no split GEMM (the reference
      narrows those layers to 256 - len(x0) for exactly this reason);
    * backward = GEMMs (weight gradients 64-way split-K), one fused `d_y * cos(pre)` pass per layer on the strided views,
      HIP column sums for the bias gradients.  PyTorch's stock composition spends 45 % of an iteration in `reduce_kernel`."""

    @staticmethod
    def forward(ctx, x0, skip, n_hidden, *wb):
        L = len(wb) // 2
        weights, biases = wb[0::2], wb[1::2]
        d0 = x0.shape[1]
        inp, inps, pres = x0, [], []
        for l in range(L):
            W, b = weights[l], biases[l]
            inps.append(inp)
            pre = torch.addmm(b, inp, W.t())
            if l == L - 1:
                out = pre
                break
            pres.append(pre)
            n_out = W.shape[0]
            if (l + 1) in skip:
                buf = torch.empty((x0.shape[0], n_out + d0), dtype=x0.dtype, device=x0.device)
                torch.sin(pre, out=buf[:, :n_out])
                buf[:, n_out:] = x0
                inp = buf
            else:
                inp = torch.sin(pre)
        ctx.save_for_backward(x0, *weights, *inps[1:], *pres)
        ctx.skip, ctx.L = tuple(skip), L
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = ctx.L
        saved = ctx.saved_tensors
        x0, weights = saved[0], saved[1:1 + L]
        inps = (x0,) + tuple(saved[1 + L:1 + L + (L - 1)])
        pres = saved[1 + L + (L - 1):]
        d0 = x0.shape[1]
        need_x0 = ctx.needs_input_grad[0]
        d_x0 = torch.zeros_like(x0) if need_x0 else None
        grads = [None] * (2 * L)
        g = grad_out.contiguous()
        for l in range(L - 1, -1, -1):
            grads[2 * l] = _split_k_tn(g, inps[l])
            grads[2 * l + 1] = _column_sum(g)
            if l == 0:
                if need_x0:
                    d_x0 += g @ weights[0]
                break
            d_inp = g @ weights[l]                                   # [M, in_l]
            n_prev = weights[l - 1].shape[0]
            if need_x0 and l in ctx.skip:
                d_x0 += d_inp[:, n_prev:]
            g = _sin_bwd(d_inp[:, :n_prev], pres[l - 1])             # through sin of layer l-1, contiguous [M, n_prev]
        return (d_x0, None, None, *grads)


def _ceil4(n: int) -> int:
    return (n + 3) // 4 * 4


def _pad_cols(t: torch.Tensor, cols: int) -> torch.Tensor:
    """[rows, n] -> contiguous [rows, cols] with zero padding (row stride a multiple of 4 floats for the MFMA kernels)."""
    if t.shape[1] == cols and t.is_contiguous():
        return t
    out = t.new_zeros((t.shape[0], cols))
    out[:, :t.shape[1]] = t
    return out


class _PosMlpHipFn(torch.autograd.Function):
    """The coordinate MLP on the hand-written exact-f32 MFMA kernels of libmatpbr.so (csrc/posmlp_kernels.hip): one launch per sine
    layer forward (`sin`/`cos` in the GEMM epilogue; the pre-activation is never stored), one per layer for dL/d input (the `* cos`
    and the bias-gradient column sums in the epilogue) and one slab-split launch per weight gradient.  The skinny ends (the
    zero-initialised output layer [M,256] x [256,5], its weight gradient and the first layer's) run on the streaming kernels of the
    same library when the hidden layers are 256 wide; other widths fall back to BLAS calls for those three.
    Same buffer trick as `_PosMlpFn` for the skip layers: the producer writes the first columns of a 256-wide buffer whose tail
    holds x0.  The input x0 gets no gradient here (it is the constant `start_arm` of the optimisation loop)."""

    # Partial products per f32 product of the 256-wide sine layers (large point sets): 0 = the exact-f32 MFMA kernels; 6 / 9 = the
    # split-operand kernels on the bf16 matrix pipe (three bf16 pieces per operand, f32 accumulation; posmlp_kernels.hip "bx").
    PRODUCTS = 6
    MIN_ROWS = 8192         # persistent 128-row tiles from here up ...
    SMALL_ROWS = 1024       # ... one 32x32 tile per wave straight from L2 up to here (the 16x32 envmap MLP); BLAS in between

    @staticmethod
    def supported(x0: torch.Tensor, skip, weights) -> bool:
        L = len(weights)
        if not (x0.is_cuda and x0.dtype == torch.float32 and not x0.requires_grad and L >= 2):
            return False
        if _PosMlpHipFn.SMALL_ROWS < x0.shape[0] < _PosMlpHipFn.MIN_ROWS:
            return False
        d0, k = x0.shape[1], x0.shape[1]
        for l in range(L - 1):
            n = weights[l].shape[0]
            if weights[l].shape[1] != k or n > 256 or k > 256:
                return False
            k = n + d0 if (l + 1) in skip else n
            if k % 4:
                return False
        return weights[L - 1].shape[1] == k and k <= 256

    @staticmethod
    def forward(ctx, x0, skip, *wb):
        from . import ops

        L = len(wb) // 2
        weights, biases = wb[0::2], wb[1::2]
        M, d0 = x0.shape
        inp, K = _pad_cols(x0, _ceil4(d0)), d0
        inps, coss = [], []
        for l in range(L - 1):
            W, b = weights[l], biases[l]
            n = W.shape[0]
            width = n + d0 if (l + 1) in skip else n
            buf = torch.empty((M, width), dtype=torch.float32, device=x0.device)
            cbuf = torch.empty((M, width), dtype=torch.float32, device=x0.device)
            inps.append(inp)
            if _PosMlpHipFn.PRODUCTS and M % 128 == 0 and M >= _PosMlpHipFn.MIN_ROWS and width == 256 and inp.stride(0) >= (K + 31) // 32 * 32:
                ops.mlp_layer_fwd_bx(inp, ops.mlp_split_weights(W, n, K), b, buf, cbuf, n, K, _PosMlpHipFn.PRODUCTS)
            else:
                ops.mlp_layer_fwd(inp, _pad_cols(W, _ceil4(K)), b, buf, cbuf, K)
            if width != n:
                buf[:, n:] = x0
            coss.append(cbuf)
            inp, K = buf, width
        inps.append(inp)
        small = M <= _PosMlpHipFn.SMALL_ROWS
        if small:                                      # the output layer on the small-tile kernel too: no BLAS call in the iteration
            n_last = weights[L - 1].shape[0]
            out = torch.empty((M, _ceil4(n_last)), dtype=torch.float32, device=x0.device)   # row stride: a multiple of 4 floats
            ops.mlp_layer_fwd(inp, _pad_cols(weights[L - 1], _ceil4(K)), biases[L - 1], out, None, K)
            out = out[:, :n_last]
        elif weights[L - 1].shape[0] in (3, 5, 8) and K % 4 == 0 and inp.stride(0) % 4 == 0:
            n_last = weights[L - 1].shape[0]                  # one streaming pass of libmatpbr.so (no BLAS call in the iteration)
            out = torch.empty((M, 8), dtype=torch.float32, device=x0.device)
            ops.mlp_skinny_fwd(inp, _pad_cols(weights[L - 1], _ceil4(K)), biases[L - 1].contiguous(), out, K)
            out = out[:, :n_last]
        else:
            out = torch.addmm(biases[L - 1], inp, weights[L - 1].t())
        ctx.save_for_backward(x0, *weights, *inps, *coss)
        ctx.L, ctx.small = L, small
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import ops

        L = ctx.L
        saved = ctx.saved_tensors
        x0, weights = saved[0], saved[1:1 + L]
        inps = saved[1 + L:1 + 2 * L]
        coss = saved[1 + 2 * L:]
        grads = [None] * (2 * L)
        g = grad_out.contiguous()
        n_out = g.shape[1]
        k_last = weights[L - 1].shape[1]
        skinny = (not ctx.small) and n_out <= 8 and inps[L - 1].stride(0) >= 256 and k_last <= 256       # the 256-wide operand of the skinny kernels
        if skinny:
            g = _pad_cols(g, 8)
            d_w = torch.empty((n_out, k_last), dtype=torch.float32, device=g.device)
            d_b = torch.empty(8, dtype=torch.float32, device=g.device)
            ops.mlp_skinny_bwd_weight(g, inps[L - 1], d_w, n_out, k_last, d_bias=d_b)
            grads[2 * (L - 1)], grads[2 * (L - 1) + 1] = d_w, d_b[:n_out]
            n_red = n_out
        else:
            grads[2 * (L - 1) + 1] = _column_sum(g)
            g, n_red = _pad_cols(g, _ceil4(n_out)), n_out
            grads[2 * (L - 1)] = (ops.mlp_layer_bwd_weight(g, inps[L - 1], n_out, k_last) if ctx.small
                                  else _split_k_tn(g[:, :n_out], inps[L - 1]))
        for l in range(L - 1, 0, -1):                  # g = dL/d pre of layer l  ->  dL/d pre of layer l-1, its bias gradient
            n_prev = weights[l - 1].shape[0]
            wt = _pad_cols(weights[l][:, :n_prev].t(), 256 if n_red > 224 else _ceil4(n_red))   # 256-wide rows: the fast kernels' precondition
            g_prev = torch.empty_like(coss[l - 1])
            d_b = torch.empty(n_prev, dtype=torch.float32, device=g.device)
            M = g.shape[0]
            if (_PosMlpHipFn.PRODUCTS and M % 128 == 0 and M >= _PosMlpHipFn.MIN_ROWS and g_prev.stride(0) == 256
                    and g.stride(0) >= (n_red + 31) // 32 * 32 and wt.stride(0) >= n_red):
                ops.mlp_layer_bwd_input_bx(g, ops.mlp_split_weights(wt, n_prev, n_red), coss[l - 1], g_prev, n_prev, n_red, d_b, _PosMlpHipFn.PRODUCTS)
            else:
                ops.mlp_layer_bwd_input(g, wt, coss[l - 1], g_prev, n_prev, n_red, d_b)
            grads[2 * (l - 1) + 1] = d_b
            g, n_red = g_prev, n_prev
            if l - 1 >= 1:
                x_in, k_in = inps[l - 1], weights[l - 1].shape[1]
                if _PosMlpHipFn.PRODUCTS and M % 16 == 0 and M >= _PosMlpHipFn.MIN_ROWS and g.stride(0) >= 256 and x_in.stride(0) >= 256:
                    grads[2 * (l - 1)] = ops.mlp_layer_bwd_weight_bx(g, x_in, n_prev, k_in, _PosMlpHipFn.PRODUCTS)
                else:
                    grads[2 * (l - 1)] = ops.mlp_layer_bwd_weight(g, x_in, n_prev, k_in)
        if ctx.small:
            grads[0] = ops.mlp_layer_bwd_weight(g, inps[0], n_red, x0.shape[1])
        else:
            d0 = x0.shape[1]
            if d0 <= 16 and g.stride(0) >= 256:           # K = 15: the skinny kernel (x0 by scalar loads, one pass over g)
                d_w0 = torch.zeros((n_red, _ceil4(d0)), dtype=torch.float32, device=g.device)
                ops.mlp_skinny_bwd_weight(_pad_cols(x0, 8 if d0 <= 8 else 16), g, d_w0, d0, n_red, transposed_out=True)
                grads[0] = d_w0[:, :d0]
            else:
                grads[0] = _split_k_tn(g[:, :n_red], x0)
        return (None, None, *grads)


class _Sine(nn.Module):
    """Hidden layer: Linear followed by sin (the reference's SineLayer applies no omega_0, mymodels/mlps.py:102-103)."""

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
        self._x0_cache = None

    def _points(self, img: torch.Tensor) -> torch.Tensor:
        h, w = grid_shape(img.shape[0])
        key = (h, w, img.device, img.dtype)
        if key not in self._code:
            self._code[key] = positional_code(h, w, self.multires, img.device, img.dtype)
        # the optimisation loops feed the same constant tensor every iteration: keep its [code | img] matrix
        tag = (img.data_ptr(), img._version, tuple(img.shape))
        if not img.requires_grad and self._x0_cache is not None and self._x0_cache[0] == tag:
            return self._x0_cache[1]
        x0 = torch.cat([self._code[key], img], dim=1)
        self._x0_cache = (tag, x0) if not img.requires_grad else None
        return x0

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        x0 = self._points(img)
        wb = []
        for l in range(self.n_layers):
            layer = getattr(self, f"lin{l}")
            lin = layer.linear if l < self.n_layers - 1 else layer
            wb += [lin.weight, lin.bias]
        if _PosMlpHipFn.supported(x0, self.skip, wb[0::2]):
            x = _PosMlpHipFn.apply(x0, self.skip, *wb)
        else:
            x = _PosMlpFn.apply(x0, self.skip, self.n_layers - 1, *wb)
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
