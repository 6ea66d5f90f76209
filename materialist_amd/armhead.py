"""Hot loop B in the reference's default `pos_mlp` mode, launch by launch on the C ABI (inverse_img_w_mi.py:159-172,470-590):

    arm    = brdf_net(start_arm)                   PosMLP(output_type='arm'): 15 -> 241 -> 256 -> 241 -> 256 -> 5, residual tanh head
    maps   = albedo | roughness * 0.93 + 0.07 | metallic                                   (:493-496)
    image  = render_w_brdf(scene, maps, spp)       light and geometric normals fixed during the part
    loss   = 3 (L1/MSE) MSE + L1 on x^(1/2.2) + scale_delta * L1 anchors; backward; AdamW + StepLR; SaveBest; EarlyStopping

At 512 x 512 the network is eleven products over 262144 points.  Composed from framework ops (autograd node, three BLAS calls for
the skinny ends, the head as elementwise kernels, a foreach AdamW) the iteration carried ~0.4 ms of glue around the sine layers;
here every launch of the iteration is a kernel of libmatpbr.so: one launch that cuts all weights into their bf16 pieces, the first layer on the f32 MFMA kernel, the 256-wide layers on
the split-operand bf16 MFMA kernels (forward with sin/cos epilogue, input gradient with cos and bias-gradient epilogue, weight
gradient), the output layer with the tanh head, the head's backward, the two skinny weight gradients, one AdamW launch over a
flat parameter buffer whose step count and learning rate live in device memory, and one select for the SaveBest snapshot of the
weights.  No autograd graph, no allocation, no BLAS.  PyTorch holds the memory; the parameters stay `torch.nn.Parameter`s of the
PosMLP module (views of the flat buffer), so `state_dict()` / `load_state_dict()` and the reference's checkpoint layout are
untouched.
"""
from __future__ import annotations

import types
from typing import Dict, Optional

import torch

from . import _lib, ops
from . import loss as _loss
from . import render as _render
from .posmlp import _PosMlpHipFn

_al4 = lambda n: (n + 3) // 4 * 4


def flat_state(net: torch.nn.Module, dev) -> dict:
    """The module's parameters as views of one flat fp32 buffer: weights with rows padded to a multiple of 4 floats, biases padded
    likewise (every view 16-byte aligned).  Created once per module; later phases find it on the module."""
    L = net.n_layers
    lins = [(getattr(net, f"lin{l}").linear if l < L - 1 else getattr(net, f"lin{l}")) for l in range(L)]
    st = getattr(net, "_flat_state", None)
    if (st is not None and "spans" in st and st["flat"].device == torch.device(dev)
            and all(lin.weight.data_ptr() == wp.data_ptr() and lin.bias.data_ptr() == bp.data_ptr() for lin, (wp, bp) in zip(lins, st["views"]))):
        return st                                             # still the module's storage (nothing re-pointed the parameters since)
    total = sum(lin.weight.shape[0] * _al4(lin.weight.shape[1]) + _al4(lin.weight.shape[0]) for lin in lins)
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    off, views, spans = 0, [], {}
    for l, lin in enumerate(lins):
        n, k = lin.weight.shape
        wp = flat[off:off + n * _al4(k)].view(n, _al4(k))
        wp[:, :k].copy_(lin.weight.detach())
        lin.weight.data = wp[:, :k]                          # a strided view when k is not a multiple of 4 (the first layer)
        off += n * _al4(k)
        bp = flat[off:off + n]
        bp.copy_(lin.bias.detach())
        lin.bias.data = bp
        off += _al4(n)
        views.append((wp, bp))
    names = {id(p): name for name, p in net.named_parameters()}
    for lin, (wp, bp) in zip(lins, views):
        spans[names[id(lin.weight)]] = (wp, lin.weight.shape[1])
        spans[names[id(lin.bias)]] = (bp, None)
    st = net._flat_state = {"flat": flat, "views": views, "spans": spans}
    return st


class ArmMlpPhase:
    """Same interface as `loop.PosMlpBrdfPhase` (step, step_and_check, stats, best, best_img, best_weights, pred, opt.param_groups)."""

    PACKED = True     # one float per sine activation (class switch: the tests run both)
    FUSED_OUT_BWD = True   # the output layer's backward pass in one launch over the last sine layer's activations (class switch)
    FUSED_FIRST_BWD = True # the first layer's weight / bias gradient from the epilogue of the input-gradient kernel above it (class switch)
    BWD_F16 = True         # the backward products of the 256-wide layers on two f16 pieces under one exponent per 128-row tile
                           # (include/matpbr.h `matpbr_mlp_layer_bwd_input_blk`); needs PACKED, FUSED_OUT_BWD, FUSED_FIRST_BWD.  Class switch
    FWD_CHAIN = True       # the whole forward pass as one launch with the activations in registers between the layers
                           # (include/matpbr.h `matpbr_mlp_chain_fwd`; needs PACKED, FWD_PRODUCTS 3 and the reference's 15-241-256-241-256-5 shape)
    DEFER_REDUCE = True    # the folds of the backward pass's partial sums (weight gradients' slabs, column sums behind the bias gradients) in ONE
                           # launch before the optimiser step instead of one behind every product (include/matpbr.h `matpbr_mlp_reduce_jobs`); needs
                           # BWD_F16.  Class switch: the same bits either way (tests/test_gpu_parity.py)
    FWD_PRODUCTS = 3       # forward sine layers on two f16 pieces per operand, three products (include/matpbr.h `matpbr_mlp_split_weights_fmt`);
                           # 0: as the backward products (`_PosMlpHipFn.PRODUCTS`, three bf16 pieces).  Class switch: the tests run both

    @staticmethod
    def why_not(scene: _render.Scene, gt_image: torch.Tensor, net: torch.nn.Module, optimize_part: str, mask) -> Optional[str]:
        """None when the launch-by-launch phase takes this part; otherwise what sends it to the autograd composition (several times slower:
        bench.py modes.pos_mlp_exact_f32 is that class of loop) -- optimize.py says so in the run's log."""
        if gt_image.ndim != 3:
            return "a batch of images (the network optimises one image per process, as the reference does)"
        if not gt_image.is_cuda:
            return "the image is not on a GPU"
        if not scene.use_mesh_normal or "n" in optimize_part:
            return "predicted normals / a part that moves the normal map (the eight-output 'armn' network and the render under a normal map: loop.PosMlpNormalPhase with armhead.MlpEngine)"
        if getattr(net, "output_type", None) != "arm" or not _PosMlpHipFn.PRODUCTS:
            return "not the five-output 'arm' network on the split-operand kernels"
        M = gt_image.shape[0] * gt_image.shape[1]
        if M % 128 or M < _PosMlpHipFn.MIN_ROWS:
            return f"{M} pixels: the layer kernels take whole 128-row tiles of at least {_PosMlpHipFn.MIN_ROWS} rows"
        L = net.n_layers
        d0 = getattr(net, "lin0").linear.weight.shape[1]
        for l in range(L - 1):
            n = getattr(net, f"lin{l}").linear.weight.shape[0]
            if (n + d0 if (l + 1) in net.skip else n) != 256:
                return "hidden layers that are not 256 wide"
        if getattr(net, f"lin{L - 1}").weight.shape != (5, 256) or d0 > 16:
            return "an output layer that is not 256 -> 5, or more than 16 inputs"
        return None

    @staticmethod
    def supported(scene: _render.Scene, gt_image: torch.Tensor, net: torch.nn.Module, optimize_part: str, mask) -> bool:
        return ArmMlpPhase.why_not(scene, gt_image, net, optimize_part, mask) is None

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, net: torch.nn.Module, start_arm: torch.Tensor, fixed: Dict[str, torch.Tensor],
                 optimize_part: str = "arm", spp: int = 64, lr: float = 3e-4, scale_delta: float = 0.1, patience: int = 0,
                 min_delta: float = 0.0, best_mse: Optional[torch.Tensor] = None, history_len: int = 5000, weight_decay: float = 0.01,
                 mask: Optional[torch.Tensor] = None):
        from .loop import _lib_ws

        if not ArmMlpPhase.supported(scene, gt_image, net, optimize_part, None):
            raise NotImplementedError("ArmMlpPhase: 'arm' network with 256-wide layers on one image of at least 8192 pixels (a multiple of 128)")
        self.ops, self.scene, self.net, self.part = ops, scene, net, optimize_part
        self.spp, self.scale_delta = int(spp), float(scale_delta)
        self.gt = gt_image.contiguous()
        dev = self.dev = self.gt.device
        H, W = self.H, self.W = self.gt.shape[0], self.gt.shape[1]
        M = self.M = H * W
        self.gt_srgb = _loss.linear_to_srgb(self.gt).contiguous()
        self.start_arm = start_arm.detach().to(dev, torch.float32).contiguous()
        self.fixed = {k: v.detach().contiguous() for k, v in fixed.items()}
        self.orig = {"albedo": self.start_arm[:, 0:3].reshape(H, W, 3).contiguous(),             # regulariser anchors (:189-201)
                     "roughness": self.start_arm[:, 3:4].reshape(H, W, 1).contiguous(),
                     "metallic": self.start_arm[:, 4:5].reshape(H, W, 1).contiguous()}
        self.products = int(_PosMlpHipFn.PRODUCTS)
        self.fwd_products = int(self.FWD_PRODUCTS) or self.products
        self.bwd_f16 = bool(self.BWD_F16 and self.PACKED and self.FUSED_OUT_BWD and self.FUSED_FIRST_BWD)
        # ---- network state --------------------------------------------------------------------------------------------------------
        st = flat_state(net, dev)
        self.flat, self.views, self._spans = st["flat"], st["views"], st["spans"]
        self.L = net.n_layers
        self.gflat = torch.zeros_like(self.flat)
        self.adam_m, self.adam_v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)       # a fresh AdamW per part (:470)
        self.hyper = torch.tensor([float(lr), 0.0], dtype=torch.float32, device=dev)
        self.weight_decay = float(weight_decay)
        self._lr, self._sched_epoch, self.base_lr = float(lr), 0, float(lr)
        self.opt = types.SimpleNamespace(param_groups=[{"lr": float(lr)}])                        # what the callers read back
        self.gviews, off = [], 0
        for wp, bp in self.views:
            gw = self.gflat[off:off + wp.numel()].view_as(wp)
            off += wp.numel()
            gb = self.gflat[off:off + _al4(bp.numel())]
            off += _al4(bp.numel())
            self.gviews.append((gw, gb))
        self._best_flat = self.flat.clone()
        # ---- activations ----------------------------------------------------------------------------------------------------------
        E = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        x0 = net._points(self.start_arm)
        self.d0 = d0 = x0.shape[1]
        self.x0p = E(M, 16)
        self.x0p[:, :d0] = x0
        self.ns = [self.views[l][0].shape[0] for l in range(self.L - 1)]
        self.bufs = [E(M, 256) for _ in range(self.L - 1)]
        for l, n in enumerate(self.ns):
            if n != 256:
                self.bufs[l][:, n:] = x0                        # cat(x, x0) of the skip layers: x0 is constant, written once
        # PACKED: the sines carry the sign of their cosine in the last mantissa bit and the backward pass rebuilds cos = +-sqrt(1 - sin^2)
        # (include/matpbr.h `matpbr_mlp_layer_fwd_sgn`): no cosine matrices, a third less traffic per forward layer
        self.packed = bool(self.PACKED)
        self.cbufs = [None if self.packed else torch.empty(M, 256, dtype=torch.float32, device=dev) for _ in range(self.L - 1)]
        self.gbufs = [torch.empty(M, 256, dtype=torch.float32, device=dev) for _ in range(2)]
        self.th, self.d_x = E(M, 8), E(M, 8)
        self.w_out_t = E(256, 8)                                 # the output weight transposed (operand of the input-gradient kernel)
        # split-operand images of the 256-wide layers' weights, forward and (transposed) backward operand: written by ONE launch
        # at the start of every iteration (the weights change once per optimiser step)
        nbytes = int(_lib.load().matpbr_mlp_wsplit_bytes(256))
        self.wsplit_f = {l: torch.empty(nbytes, dtype=torch.uint8, device=dev) for l in range(1, self.L - 1)}
        self.wsplit_b = {l: torch.empty(nbytes, dtype=torch.uint8, device=dev) for l in range(1, self.L - 1)}
        jobs = []
        for l in range(1, self.L - 1):
            wp, _ = self.views[l]
            jobs.append((wp.data_ptr(), wp.stride(0), self.ns[l], 256, ops.WSPLIT_F16X2 if self.fwd_products == 3 else 0,
                         self.wsplit_f[l].data_ptr()))                                                             # W_l [n_l, 256]
            jobs.append((wp.data_ptr(), wp.stride(0), self.ns[l - 1], self.ns[l], 1 | (ops.WSPLIT_F16X2 if self.bwd_f16 else 0),
                         self.wsplit_b[l].data_ptr()))                                                             # (W_l[:, :n_{l-1}])^T
        if len(jobs) > 8:
            raise NotImplementedError("ArmMlpPhase: at most four 256-wide layers after the first")
        import ctypes as _ct

        self.chain = bool(self.FWD_CHAIN and self.packed and self.fwd_products == 3 and self.L == 5 and all(n in (241, 256) for n in self.ns)
                          and self.views[-1][0].shape[0] <= 8 and self.d0 <= 15 and all((n == 241) == ((l + 1) in net.skip) for l, n in enumerate(self.ns)))
        # one preparation launch per iteration (matpbr_mlp_chain_prep): the chain's forward images, the backward products' f16 images, the reset of
        # the gradient tiles' maxima
        self.prep_all = bool(self.chain and self.bwd_f16 and self.L == 5)
        if self.chain:
            jobs = [] if self.prep_all else [jb for jb in jobs if jb[4] & 1]   # the forward images are the chain's own
            lib = _lib.load()
            self._chain_images = torch.empty(int(lib.matpbr_mlp_chain_images_bytes()), dtype=torch.uint8, device=dev)
            P_ = _ct.c_void_p
            ws_, bs_ = [v[0] for v in self.views], [v[1] for v in self.views]
            self._chain_prep = [(P_ * 5)(*[w_.data_ptr() for w_ in ws_]), (_ct.c_int * 5)(*[w_.stride(0) for w_ in ws_]), (_ct.c_int * 5)(*[w_.shape[0] for w_ in ws_]),
                                (P_ * 5)(*[b_.data_ptr() for b_ in bs_]), int(self.d0), P_(self._chain_images.data_ptr()), None, None, 0]
            if self.prep_all:
                self._chain_prep[6] = (P_ * 3)(*[self.wsplit_b[l].data_ptr() for l in (1, 2, 3)])
            self._chain_out = (P_ * 4)(*[b_.data_ptr() for b_ in self.bufs])
            self._chain_n = (_ct.c_int * 4)(*self.ns)
        nj = len(jobs)
        self._split_args = ((_ct.c_void_p * nj)(*[j[0] for j in jobs]), (_ct.c_int * nj)(*[j[1] for j in jobs]), (_ct.c_int * nj)(*[j[2] for j in jobs]),
                            (_ct.c_int * nj)(*[j[3] for j in jobs]), (_ct.c_int * nj)(*[j[4] for j in jobs]), (_ct.c_void_p * nj)(*[j[5] for j in jobs]), nj)
        # block exponents of the gradient matrices (one row of tile maxima per sine layer's dL/d pre), zeroed once per iteration
        self.tmax = torch.zeros(max(self.L - 1, 1), M // 128, dtype=torch.int32, device=dev) if self.bwd_f16 else None
        if self.prep_all:
            self._chain_prep[7], self._chain_prep[8] = _ct.c_void_p(self.tmax.data_ptr()), self.tmax.numel()
        self._jobs = (_lib.ReduceJob * 16)()                     # the backward pass's deferred folds (DEFER_REDUCE)
        self.maps = {"albedo": E(H, W, 3), "roughness": E(H, W, 1), "metallic": E(H, W, 1)}
        keys = {"a": "albedo", "r": "roughness", "m": "metallic"}
        self.live = [keys[c] for c in self.part if c in keys]
        # ---- render / loss state ----------------------------------------------------------------------------------------------------
        self.stats = ops.new_loss_stats(1, dev)
        if best_mse is not None:
            self.stats[:, ops.STAT_BEST] = best_mse.to(dev).reshape(-1)
        # EarlyStopping lives in the statistics buffer on the device (matpbr_brdf_loss_stats_es): the host polls it every `sync_every`
        # iterations; iterations enqueued past the stop change nothing (statistics, snapshot and AdamW rest)
        self.patience, self.min_delta, self.sync_every = int(patience), float(min_delta), 8
        self.pred = torch.empty_like(self.gt)
        self.g = {"albedo": E(H, W, 3), "roughness": E(H, W, 1), "metallic": E(H, W, 1)}
        self.best = {k: v.clone() for k, v in self.fixed.items()}
        self.best_img = torch.zeros_like(self.gt)
        self.hist = E(history_len, 1)
        self.ws = torch.empty(int(_lib_ws(1)) // 4, dtype=torch.float32, device=dev)
        self._n = scene.shading_normal().contiguous()
        self._light = scene.light.detach().contiguous()
        self.dcache = ops.diffuse_cache(self._n, self._light, self.spp, scene.fov)
        self.jac = ops.plane9(self.gt)
        # a part that leaves the roughness alone ('a' of --opt_order 'rm a'): the specular sums are constants of the part (:497-504)
        self.s1 = None if "roughness" in self.live else torch.empty((3, 1, H, W), dtype=torch.float32, device=dev)
        self._bg_mask = scene.bg_mask
        if self._bg_mask is not None:
            self._bg_rgb = (scene.bg_basis @ self._light.reshape(25, 3)).contiguous()
            self._bg_idx = ops.background_index(self._bg_mask)          # once: boolean-mask indexing would synchronise with the host every iteration
            self._bg_rows = self._bg_rgb.reshape(-1, 3)[self._bg_idx].contiguous()
            self._bg_rows_t = self._bg_rows.t().contiguous()
        # --use_mask (:509-511): inside the mask roughness and metallic are their masked means (of the clamped maps); the render, the loss and
        # the snapshots see the filled maps, the network receives the mean of the masked gradients through each entry's own clamp
        self._mask_u8 = None if mask is None else mask.to(dev).reshape(H, W).to(torch.uint8).contiguous()
        self._fed = {k: E(H, W, 1) for k in ("roughness", "metallic")} if mask is not None else None
        self.t = 0

    # ------------------------------------------------------------------------------------------------------------------------------
    def set_lr(self, lr: float) -> None:
        self._lr = float(lr)
        self.hyper[0:1].fill_(self._lr)
        self.opt.param_groups[0]["lr"] = self._lr

    def forward(self) -> Dict[str, torch.Tensor]:
        """brdf_net(start_arm) and the maps of :493-504 (maps that the part does not optimise keep their fixed values)."""
        o, P = ops, self.fwd_products
        if self._split_args[-1] > 0:
            with torch.cuda.device(self.dev):
                _lib.check(_lib.load().matpbr_mlp_split_weights_multi(*self._split_args, o._stream(self.flat)), "matpbr_mlp_split_weights_multi")
        if self.chain:
            live = self.live
            lib, st = _lib.load(), o._stream(self.flat)
            mp = [o._ptr(self.maps[k]) if k in live else None for k in ("albedo", "roughness", "metallic")]
            with torch.cuda.device(self.dev):
                _lib.check(lib.matpbr_mlp_chain_prep(*self._chain_prep, st), "matpbr_mlp_chain_prep")
                _lib.check(lib.matpbr_mlp_chain_fwd(o._ptr(self.x0p), self.x0p.stride(0), o._ptr(self._chain_images), self._chain_out, 256, self._chain_n,
                                                    o._ptr(self.start_arm), self.start_arm.stride(0), o._ptr(self.th), *mp, self.views[-1][0].shape[0], self.M, st),
                           "matpbr_mlp_chain_fwd")
            return {k: (self.maps[k] if k in live else self.fixed[k]) for k in self.maps}
        wp, bp = self.views[0]
        # skip layers on the split-operand kernel: whole 16-byte stores over the x0 tail, rewritten by a small launch (-20 us per layer
        # against guarding the straddling word of every row in the epilogue); the first layer's kernel keeps its guard (no gain there)
        tail = lambda l: self.x0p if self.ns[l] != 256 else None
        if self.packed:
            o.mlp_layer_fwd(self.x0p, wp, bp, self.bufs[0], None, self.d0, packed=True)
        else:
            o.mlp_layer_fwd(self.x0p, wp, bp, self.bufs[0], self.cbufs[0], self.d0)
        live = self.live
        maps = [self.maps[k] if k in live else None for k in ("albedo", "roughness", "metallic")]
        wo, bo = self.views[-1]
        for l in range(1, self.L - 1):
            wp, bp = self.views[l]
            if l == self.L - 2 and self.ns[l] == 256:           # the last sine layer finishes the network in its epilogue
                o.mlp_layer_fwd_bx_head(self.bufs[l - 1], self.wsplit_f[l], bp, self.bufs[l], self.cbufs[l], 256, P, wo, bo, self.start_arm, self.th,
                                        *maps)
            else:
                o.mlp_layer_fwd_bx(self.bufs[l - 1], self.wsplit_f[l], bp, self.bufs[l], self.cbufs[l], self.ns[l], 256, P, tail=tail(l))
        if self.ns[-1] != 256:
            o.mlp_arm_head_fwd(self.bufs[-1], wo, bo, self.start_arm, self.th, *maps, 256)
        return {k: (self.maps[k] if k in live else self.fixed[k]) for k in self.maps}

    def backward(self) -> None:
        """d loss / d maps (self.g) -> the gradient of every parameter, into the flat gradient buffer."""
        o, P, live = ops, self.products, self.live
        o.mlp_arm_head_bwd(self.g["albedo"] if "albedo" in live else None, self.g["roughness"] if "roughness" in live else None,
                           self.g["metallic"] if "metallic" in live else None, self.th, self.d_x)
        wp, _ = self.views[-1]
        gw, gb = self.gviews[-1]
        g_prev = self.gbufs[0]
        _, gb_prev = self.gviews[self.L - 2]
        # the output layer in ONE pass over the last sine layer's activations: its weight / bias gradient, dL/d pre of that sine layer and its
        # bias gradient (separately: a second 268 MB read of the sines, a transposing copy of the output weight, two reduce launches)
        f16 = self.bwd_f16
        import ctypes as _ct

        jobs, nj = (self._jobs, 0) if (f16 and self.DEFER_REDUCE) else (None, 0)
        slot = (lambda k: (_ct.cast(_ct.byref(jobs, k * _ct.sizeof(_lib.ReduceJob)), _ct.c_void_p), f"_d{k}")) if jobs is not None else (lambda k: None)
        if f16:
            if not self.prep_all:                                # (otherwise zeroed by the iteration's preparation launch)
                self.tmax.zero_()
            o.mlp_out_layer_bwd_tmax(self.d_x, self.bufs[-1], wp, g_prev, self.tmax[self.L - 2], gw, gb, gb_prev, 5, self.ns[-1], defer=slot(nj))
            nj += 1
        elif self.FUSED_OUT_BWD:
            o.mlp_out_layer_bwd(self.d_x, self.bufs[-1], None if self.packed else self.cbufs[-1], wp, g_prev, gw, gb, gb_prev, 5, self.ns[-1])
        else:
            o.mlp_skinny_bwd_weight(self.d_x, self.bufs[-1], gw, 5, 256, d_bias=gb)
            self.w_out_t[:, :5].copy_(wp.t())
            o.mlp_layer_bwd_input(self.d_x, self.w_out_t, self.bufs[-1] if self.packed else self.cbufs[-1], g_prev, self.ns[-1], 5, gb_prev,
                                  packed=self.packed)
        g, n_red = g_prev, self.ns[-1]
        for l in range(self.L - 2, 0, -1):                       # g = dL/d pre of layer l: its weight gradient, then dL/d pre of layer l-1
            wp, _ = self.views[l]
            gw, _ = self.gviews[l]
            if f16:
                o.mlp_layer_bwd_weight_blk(g, self.tmax[l], self.bufs[l - 1], n_red, 256, out=gw, defer=slot(nj))
                nj += 1
            else:
                o.mlp_layer_bwd_weight_bx(g, self.bufs[l - 1], n_red, 256, P, out=gw)
            n_prev = self.ns[l - 1]
            _, gb = self.gviews[l - 1]
            c_prev = self.bufs[l - 1] if self.packed else self.cbufs[l - 1]
            if f16:
                if l == 1:
                    o.mlp_first_layer_bwd_blk(g, self.tmax[l], self.wsplit_b[l], c_prev, self.x0p, self.gviews[0][0], self.d0, n_prev, n_red, gb, defer=slot(nj))
                    nj += 2
                    if jobs is not None:
                        o.mlp_reduce_jobs(jobs, nj, self.flat)
                    return
                g_prev = self.gbufs[0] if g is self.gbufs[1] else self.gbufs[1]
                o.mlp_layer_bwd_input_blk(g, self.tmax[l], self.wsplit_b[l], c_prev, g_prev, n_prev, n_red, gb, self.tmax[l - 1], defer=slot(nj))
                nj += 1
                g, n_red = g_prev, n_prev
                continue
            if l == 1 and self.FUSED_FIRST_BWD:
                # into the first layer: its pre-activation gradient is not stored, the same launch forms its weight and bias gradient
                o.mlp_first_layer_bwd_bx(g, self.wsplit_b[l], c_prev, self.x0p, self.gviews[0][0], self.d0, n_prev, n_red, gb, P, packed=self.packed)
                return
            g_prev = self.gbufs[0] if g is self.gbufs[1] else self.gbufs[1]
            o.mlp_layer_bwd_input_bx(g, self.wsplit_b[l], c_prev, g_prev, n_prev, n_red, gb, P, packed=self.packed)
            g, n_red = g_prev, n_prev
        gw, _ = self.gviews[0]
        o.mlp_skinny_bwd_weight(self.x0p, g, gw, self.d0, n_red, transposed_out=True)

    def step(self) -> None:
        o, sc = ops, self.scene
        d = self.forward()
        if self._mask_u8 is not None:
            d = dict(d)
            for k in ("roughness", "metallic"):                 # a map the part does not optimise is taken as it is (no clamp)
                lo, hi = (0.0, 1.0) if k in self.live else (-3.0e38, 3.0e38)
                d[k] = o.masked_mean_fill(d[k], self._mask_u8, out=self._fed[k], lo=lo, hi=hi)
        if self.s1 is not None and self.t > 0:                  # bit-identical to walking the samples again
            o.shade_fwd_cached(d["albedo"], d["metallic"], self.jac, self.s1, clamp_params=True, out=self.pred)
        else:
            o.shade_fwd(d["albedo"], d["roughness"], d["metallic"], self._n, self._light, self.spp, sc.fov, clamp_params=True, out=self.pred,
                        dcache=self.dcache, jac=self.jac, s1=self.s1)
        if self._bg_mask is not None and not (self.s1 is not None and self.t > 0):
            # pixels without geometry: the environment along the camera ray, no material gradient (after the kernels have written these buffers)
            self.pred.view(-1, 3).index_copy_(0, self._bg_idx, self._bg_rows)
            o.background_into_jac(self.jac, self.s1, self._bg_mask, self._bg_rgb, self._bg_idx, self._bg_rows_t)
        o.brdf_loss_stats(self.pred, self.gt, self.gt_srgb, d["albedo"], d["roughness"], d["metallic"], self.orig["albedo"],
                          self.orig["roughness"], self.orig["metallic"], self.scale_delta, self.stats, self.ws, optimize_part=self.part,
                          es_patience=self.patience, es_min_delta=self.min_delta, history=self.hist)
        o.brdf_loss_bwd_jac(d["albedo"], d["roughness"], d["metallic"], self.jac, self.pred, self.gt_srgb, self.stats,
                            self.orig["albedo"], self.orig["roughness"], self.orig["metallic"], self.scale_delta,
                            self.g["albedo"], self.g["roughness"], self.g["metallic"], self.best["albedo"], self.best["roughness"],
                            self.best["metallic"], self.best_img, optimize_part=self.part)
        if self._mask_u8 is not None:
            for k in ("roughness", "metallic"):
                if k in self.live:
                    o.masked_mean_fill(self.g[k], self._mask_u8, out=self.g[k], gate=self.maps[k])
        self.backward()
        lib = _lib.load()
        with torch.cuda.device(self.dev):       # AdamW; SaveBest keeps the weights that produced the best render (:546-547) in the same pass
            _lib.check(lib.matpbr_adamw_step_snapshot_dev(o._ptr(self.flat), o._ptr(self.gflat), o._ptr(self.adam_m), o._ptr(self.adam_v),
                                                          self.flat.numel(), o._ptr(self.hyper), 0.9, 0.999, 1e-8, self.weight_decay,
                                                          o._ptr(self._best_flat), o._ptr(self.stats), o._stream(self.flat)),
                       "matpbr_adamw_step_snapshot_dev")
        if self._lr > 1.5e-4:                                     # StepLR(100, 0.8) stepped only while lr > 1.5e-4 (:471,553-554)
            self._sched_epoch += 1
            if self._sched_epoch % 100 == 0:
                self.set_lr(self._lr * 0.8)
        self.t += 1

    @property
    def best_weights(self) -> Dict[str, torch.Tensor]:
        out, off = {}, 0
        order = []
        for wp, bp in self.views:
            order += [(off, wp), (off + wp.numel(), bp)]
            off += wp.numel() + _al4(bp.numel())
        by_ptr = {t.data_ptr(): o_ for o_, t in order}
        for name, (view, k) in self._spans.items():
            o_ = by_ptr[view.data_ptr()]
            best = self._best_flat[o_:o_ + view.numel()].view_as(view)
            out[name] = (best[:, :k] if k is not None else best).clone()
        return out

    def current_maps(self) -> Dict[str, torch.Tensor]:
        """The maps the network produces now, clamped as the render sees them (fresh tensors)."""
        d = self.forward()
        if self._mask_u8 is not None:
            d = dict(d)
            for k in ("roughness", "metallic"):
                lo, hi = (0.0, 1.0) if k in self.live else (-3.0e38, 3.0e38)
                d[k] = self.ops.masked_mean_fill(d[k], self._mask_u8, lo=lo, hi=hi)
        return {"albedo": d["albedo"].clamp(0, 1), "roughness": d["roughness"].clamp(0.07, 1), "metallic": d["metallic"].clamp(0, 1)}

    def step_and_check(self) -> bool:
        """One iteration; every `sync_every` iterations the host reads the device's EarlyStopping flag (:550) and returns True when the
        part has stopped.  The iterations enqueued between the stop and the poll were no-ops (`iterations_run` tells how many really ran)."""
        self.step()
        if self.patience <= 0 or self.t % self.sync_every:
            return False
        return bool(self.stats[0, ops.STAT_STOPPED] > 0.5)

    @property
    def iterations_run(self) -> int:
        return int(self.stats[0, ops.STAT_ITERS])

    def lr_at(self, t0: int) -> float:
        """Learning rate of the iteration with 0-based index t0: StepLR(100, 0.8) stepped only while lr > 1.5e-4 (:471,553-554)."""
        lr, k = self.base_lr, 0
        while lr > 1.5e-4 and (k + 1) * 100 <= t0:
            lr *= 0.8
            k += 1
        return lr

    def history(self) -> torch.Tensor:
        return self.hist[: self.t]


class MlpEngine:
    """The coordinate MLP alone, launch by launch on the C ABI, for networks `ArmMlpPhase` does not take whole: any first-layer width up to 16
    inputs and up to 8 raw outputs (the eight-output 'armn' network of `'n'` parts: 10 inputs -- raw pixel coordinates + 8 channels --, hidden layers 246 / 256 / 246 / 256,
    mymodels/mlps.py:236-244, inverse_img_w_mi.py:167-172).  `forward_raw()` -> the last linear layer's output [M, 8]; `backward_raw(d_x)` ->
    every parameter's gradient into the flat gradient buffer; `adamw_step(stats)` -> AdamW on the flat buffer with SaveBest's weight snapshot.
    The head (tanh, residual, clamps, `normalize`) and its backward are the caller's: a few element-wise passes over [M, 8].

    First layer on the thin-K f32 MFMA kernel (packed sines), 256-wide layers forward on two f16 pieces, backward products on two f16 pieces
    under one exponent per 128-row tile, folds deferred to one launch (the kernels of `ArmMlpPhase`'s layer-by-layer path)."""

    @staticmethod
    def why_not(net: torch.nn.Module, M: int, device) -> Optional[str]:
        if torch.device(device).type != "cuda":
            return "not on a GPU"
        if M % 128 or M < _PosMlpHipFn.MIN_ROWS:
            return f"{M} points: the layer kernels take whole 128-row tiles of at least {_PosMlpHipFn.MIN_ROWS} rows"
        L = net.n_layers
        d0 = getattr(net, "lin0").linear.weight.shape[1]
        if d0 > 16 or L < 3 or L > 6:
            return "more than 16 inputs, or fewer than two / more than five sine layers"
        for l in range(L - 1):
            n = getattr(net, f"lin{l}").linear.weight.shape[0]
            if (n + d0 if (l + 1) in net.skip else n) != 256:
                return "hidden layers that are not 256 wide"
        wo = getattr(net, f"lin{L - 1}").weight
        if wo.shape[1] != 256 or wo.shape[0] not in (3, 5, 8):
            return "an output layer that is not 256 -> 3 / 5 / 8"
        return None

    def __init__(self, net: torch.nn.Module, points_in: torch.Tensor, lr: float = 3e-4, weight_decay: float = 0.01):
        import ctypes as _ct

        dev = points_in.device
        why = MlpEngine.why_not(net, points_in.shape[0], dev)
        if why is not None:
            raise NotImplementedError("MlpEngine: " + why)
        self.net, self.dev = net, dev
        M = self.M = points_in.shape[0]
        st = flat_state(net, dev)
        self.flat, self.views, self._spans = st["flat"], st["views"], st["spans"]
        self.L = L = net.n_layers
        self.gflat = torch.zeros_like(self.flat)
        self.adam_m, self.adam_v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.hyper = torch.tensor([float(lr), 0.0], dtype=torch.float32, device=dev)
        self.weight_decay, self._lr = float(weight_decay), float(lr)
        self.gviews, off = [], 0
        for wp, bp in self.views:
            gw = self.gflat[off:off + wp.numel()].view_as(wp)
            off += wp.numel()
            gb = self.gflat[off:off + _al4(bp.numel())]
            off += _al4(bp.numel())
            self.gviews.append((gw, gb))
        self._best_flat = self.flat.clone()
        E = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        x0 = net._points(points_in.detach().to(dev, torch.float32))
        self.d0 = d0 = x0.shape[1]
        self.x0p = E(M, 8 if d0 <= 8 else 16)
        self.x0p[:, :d0] = x0
        self.ns = [self.views[l][0].shape[0] for l in range(L - 1)]
        self.n_out = self.views[-1][0].shape[0]
        self.bufs = [E(M, 256) for _ in range(L - 1)]
        for l, n in enumerate(self.ns):
            if n != 256:
                self.bufs[l][:, n:] = x0                        # cat(x, x0) of the skip layers: written once
        self.gbufs = [torch.empty(M, 256, dtype=torch.float32, device=dev) for _ in range(2)]
        self.xout = E(M, 8)
        nbytes = int(_lib.load().matpbr_mlp_wsplit_bytes(256))
        self.wsplit_f = {l: torch.empty(nbytes, dtype=torch.uint8, device=dev) for l in range(1, L - 1)}
        self.wsplit_b = {l: torch.empty(nbytes, dtype=torch.uint8, device=dev) for l in range(1, L - 1)}
        jobs = []
        for l in range(1, L - 1):
            wp, _ = self.views[l]
            jobs.append((wp.data_ptr(), wp.stride(0), self.ns[l], 256, ops.WSPLIT_F16X2, self.wsplit_f[l].data_ptr()))                 # W_l [n_l, 256]
            jobs.append((wp.data_ptr(), wp.stride(0), self.ns[l - 1], self.ns[l], 1 | ops.WSPLIT_F16X2, self.wsplit_b[l].data_ptr()))   # (W_l[:, :n_{l-1}])^T
        nj = len(jobs)
        self._split_args = ((_ct.c_void_p * nj)(*[j[0] for j in jobs]), (_ct.c_int * nj)(*[j[1] for j in jobs]), (_ct.c_int * nj)(*[j[2] for j in jobs]),
                            (_ct.c_int * nj)(*[j[3] for j in jobs]), (_ct.c_int * nj)(*[j[4] for j in jobs]), (_ct.c_void_p * nj)(*[j[5] for j in jobs]), nj)
        self.tmax = torch.zeros(L - 1, M // 128, dtype=torch.int32, device=dev)
        self._jobs = (_lib.ReduceJob * 16)()

    def set_lr(self, lr: float) -> None:
        self._lr = float(lr)
        self.hyper[0:1].fill_(self._lr)

    def forward_raw(self) -> torch.Tensor:
        o = ops
        with torch.cuda.device(self.dev):
            _lib.check(_lib.load().matpbr_mlp_split_weights_multi(*self._split_args, o._stream(self.flat)), "matpbr_mlp_split_weights_multi")
        wp, bp = self.views[0]
        o.mlp_layer_fwd(self.x0p, wp, bp, self.bufs[0], None, self.d0, packed=True)
        for l in range(1, self.L - 1):
            _, bp = self.views[l]
            o.mlp_layer_fwd_bx(self.bufs[l - 1], self.wsplit_f[l], bp, self.bufs[l], None, self.ns[l], 256, 3, tail=self.x0p if self.ns[l] != 256 else None)
        wo, bo = self.views[-1]
        o.mlp_skinny_fwd(self.bufs[-1], wo, bo, self.xout, 256)
        return self.xout

    def backward_raw(self, d_x: torch.Tensor) -> None:
        """d loss / d (the raw outputs) [M, 8] (columns beyond the network's outputs zero) -> the flat gradient buffer."""
        import ctypes as _ct

        o, jobs = ops, self._jobs
        slot = lambda k: (_ct.cast(_ct.byref(jobs, k * _ct.sizeof(_lib.ReduceJob)), _ct.c_void_p), f"_e{k}")
        nj = 0
        self.tmax.zero_()
        wo, _ = self.views[-1]
        gw, gb = self.gviews[-1]
        _, gb_prev = self.gviews[self.L - 2]
        g = self.gbufs[0]
        o.mlp_out_layer_bwd_tmax(d_x, self.bufs[-1], wo, g, self.tmax[self.L - 2], gw, gb, gb_prev, self.n_out, self.ns[-1], defer=slot(nj))
        nj += 1
        n_red = self.ns[-1]
        for l in range(self.L - 2, 0, -1):                       # g = dL/d pre of layer l: its weight gradient, then dL/d pre of layer l - 1
            gw, _ = self.gviews[l]
            o.mlp_layer_bwd_weight_blk(g, self.tmax[l], self.bufs[l - 1], n_red, 256, out=gw, defer=slot(nj))
            nj += 1
            n_prev = self.ns[l - 1]
            _, gb = self.gviews[l - 1]
            if l == 1:      # into the first layer: its pre-activation gradient is not stored, the same launch forms its weight and bias gradient
                o.mlp_first_layer_bwd_blk(g, self.tmax[l], self.wsplit_b[l], self.bufs[0], self.x0p, self.gviews[0][0], self.d0, n_prev, n_red, gb, defer=slot(nj))
                nj += 2
                break
            g_prev = self.gbufs[1] if g is self.gbufs[0] else self.gbufs[0]
            o.mlp_layer_bwd_input_blk(g, self.tmax[l], self.wsplit_b[l], self.bufs[l - 1], g_prev, n_prev, n_red, gb, self.tmax[l - 1], defer=slot(nj))
            nj += 1
            g, n_red = g_prev, n_prev
        o.mlp_reduce_jobs(jobs, nj, self.flat)

    def adamw_step(self, stats: torch.Tensor) -> None:
        """AdamW on the flat buffer; `stats` (one statistics row): SaveBest's weight snapshot in the same pass when the row says "improved", no
        update for an image whose EarlyStopping has fired, Adam's step count = the row's iteration counter (include/matpbr_mlp.h)."""
        o = ops
        with torch.cuda.device(self.dev):
            _lib.check(_lib.load().matpbr_adamw_step_snapshot_dev(o._ptr(self.flat), o._ptr(self.gflat), o._ptr(self.adam_m), o._ptr(self.adam_v),
                                                                  self.flat.numel(), o._ptr(self.hyper), 0.9, 0.999, 1e-8, self.weight_decay,
                                                                  o._ptr(self._best_flat), o._ptr(stats), o._stream(self.flat)),
                       "matpbr_adamw_step_snapshot_dev")

    @property
    def best_weights(self) -> Dict[str, torch.Tensor]:
        out, off, order = {}, 0, []
        for wp, bp in self.views:
            order += [(off, wp), (off + wp.numel(), bp)]
            off += wp.numel() + _al4(bp.numel())
        by_ptr = {t.data_ptr(): o_ for o_, t in order}
        for name, (view, k) in self._spans.items():
            o_ = by_ptr[view.data_ptr()]
            best = self._best_flat[o_:o_ + view.numel()].view_as(view)
            out[name] = (best[:, :k] if k is not None else best).clone()
        return out
