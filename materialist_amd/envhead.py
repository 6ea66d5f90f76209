"""Hot loop A with the reference's light parameterisation, launch by launch on the C ABI (inverse_img_w_mi.py:117-124,225-254):

    envmap = envmap_net(start_envmap)              PosMLP(output_type='envmap'): 13 -> 243 -> 256 -> 243 -> 256 -> 3, softplus
    image  = render_envmap(scene, envmap, spp)     materials and normals fixed during the phase (:216-220)
    loss   = MSE + L1 on x^(1/2.2); backward; Adam step; SaveBest; EarlyStopping

The envmap MLP works on 512 points: composed from framework ops (autograd node, foreach Adam, SH projection matmul, selects)
the iteration is ~60 launches of which the render is two.  Here the whole iteration is 27 launches of libmatpbr.so -- five
sine / output layers on the small-tile MFMA kernels, softplus + SH projection, the pass over the radiance transfer, the
snapshot of the best envmap, the backward chain (bias gradients in the GEMM epilogues, weights read as stored) and one Adam
launch over a flat parameter buffer whose step count and learning rate live in device memory -- captured once into a hipGraph
and replayed.  PyTorch holds the memory and the capture; the parameters stay `torch.nn.Parameter`s of the PosMLP module (views
of the flat buffer), so `state_dict()` / `load_state_dict()` and the reference's checkpoint layout are untouched.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _lib, ops
from . import loss as _loss
from . import render as _render

_al4 = lambda n: (n + 3) // 4 * 4


class EnvMlpPhase:
    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, net: torch.nn.Module, start_envmap: torch.Tensor, spp: int = 64,
                 lr: float = 1e-3, patience: int = 0, min_delta: float = 0.0, best_mse: Optional[torch.Tensor] = None,
                 history_len: int = 5000, use_graph: bool = True, env_size=(16, 32)):
        if gt_image.ndim != 3:
            raise NotImplementedError("the envmap MLP optimises one image per process (as the reference does)")
        if getattr(net, "output_type", None) != "envmap":
            raise ValueError("EnvMlpPhase needs a PosMLP with output_type='envmap'")
        self.scene, self.net, self.spp = scene, net, int(spp)
        self.gt = gt_image.contiguous()
        dev = self.dev = self.gt.device
        self.H, self.W = self.gt.shape[0], self.gt.shape[1]
        self.env_size = tuple(env_size)
        self.gt_srgb = _loss.linear_to_srgb(self.gt).contiguous()
        self.lib = _lib.load()
        self.use_graph, self._graph, self._warm, self.t = bool(use_graph), None, 0, 0
        self.patience, self.min_delta = int(patience), float(min_delta)
        # ---- network state: parameters as views of one flat buffer (weights with rows padded to 4 floats) ----------------------
        x0 = net._points(start_envmap.detach().to(dev, torch.float32))
        M, d0 = x0.shape
        if M > 1024 or M != self.env_size[0] * self.env_size[1]:
            raise ValueError("the envmap MLP runs on the small-tile kernels: at most 1024 texels")
        self.M, self.d0, self.L = M, d0, net.n_layers
        lins = [(getattr(net, f"lin{l}").linear if l < self.L - 1 else getattr(net, f"lin{l}")) for l in range(self.L)]
        from .armhead import flat_state

        st = flat_state(net, dev)   # validates that the module's parameters still alias the flat buffer on this device, rebuilds otherwise
        self.flat, self.views = st["flat"], st["views"]
        self.gflat = torch.zeros_like(self.flat)
        self.adam_m, self.adam_v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)    # a fresh Adam per phase (:225-229)
        self.hyper = torch.tensor([float(lr), 0.0], dtype=torch.float32, device=dev)
        gviews, off = [], 0
        for wp, bp in self.views:
            gw = self.gflat[off:off + wp.numel()].view_as(wp)
            off += wp.numel()
            gb = self.gflat[off:off + _al4(bp.numel())]
            off += _al4(bp.numel())
            gviews.append((gw, gb))
        # ---- activations ---------------------------------------------------------------------------------------------------------
        E = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.x0p = E(M, _al4(d0))
        self.x0p[:, :d0] = x0
        widths, Ks, ns = [], [], []
        K = d0
        for l in range(self.L - 1):
            n = lins[l].weight.shape[0]
            width = n + d0 if (l + 1) in net.skip else n
            if width % 4:
                raise ValueError("layer widths must be multiples of 4")
            widths.append(width); Ks.append(K); ns.append(n)
            K = width
        self.n_last = lins[-1].weight.shape[0]
        self.bufs = [E(M, w) for w in widths]
        for l, w in enumerate(widths):
            if w != ns[l]:
                self.bufs[l][:, ns[l]:] = x0                     # cat(x, x0) of the skip layers: x0 is constant, written once
        self.cbufs = [E(M, w) for w in widths]
        self.gbufs = [E(M, w) for w in widths]
        self.y = E(M, 4)
        self.g_out = E(M, 4)
        self.env = E(M, 3)
        self.best_env_flat = E(M, 3)
        self.light = E(25, 3)
        self.d_light = E(25, 3)
        self.proj = scene.projection(*self.env_size).contiguous()
        self.stats = ops.new_loss_stats(1, dev)
        if best_mse is not None:
            self.stats[:, ops.STAT_BEST] = best_mse.to(dev).reshape(-1)
        self.hist = E(history_len, 1)
        self.ws_env = E(int(self.lib.matpbr_env_phase_workspace_bytes(self.H, self.W, 1)) // 4 + 1)
        self.ws_in = E(int(self.lib.matpbr_mlp_bwd_input_workspace_bytes(M)) // 4 + 1)
        shp = (self.H, self.W)
        self.T = ops.shade_transfer(scene.a.contiguous(), scene.r.reshape(shp + (1,)).contiguous(), scene.m.reshape(shp + (1,)).contiguous(),
                                    scene.shading_normal().contiguous(), self.spp, scene.fov)
        if scene.bg_mask is not None:    # pixels without geometry see the environment along their camera ray: their transfer is the SH basis there
            ops.background_into_transfer(self.T, self.H, self.W, scene.bg_basis)
        # ---- the launch list of one iteration --------------------------------------------------------------------------------------
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        lib, calls = self.lib, []
        inps = [self.x0p] + self.bufs
        for l in range(self.L - 1):                              # sine layers: sin and cos from the GEMM epilogue
            wp, bp = self.views[l]
            calls.append((lib.matpbr_mlp_layer_fwd, (P(inps[l]), inps[l].stride(0), P(wp), wp.stride(0), P(bp), P(self.bufs[l]), P(self.cbufs[l]),
                                                     self.bufs[l].stride(0), M, ns[l], Ks[l])))
        wp, bp = self.views[-1]
        calls.append((lib.matpbr_mlp_layer_fwd, (P(inps[-1]), inps[-1].stride(0), P(wp), wp.stride(0), P(bp), P(self.y), None, 4, M, self.n_last, K)))
        calls.append((lib.matpbr_env_project, (P(self.y), 4, P(self.proj), P(self.env), P(self.light), M)))
        if self.FUSED_TAIL:
            # the pass over the transfer, then ONE workgroup for the fold, the SaveBest / EarlyStopping commit, the snapshot of the best envmap and
            # the projection's backward (matpbr_env_mlp_phase_step: the same bits as the four kernels below, two launches fewer)
            self._select_at = len(calls)
            self._first_arg = 21
            calls.append((lib.matpbr_env_mlp_phase_step, (P(self.T), P(self.light), P(self.gt_srgb), None, P(self.d_light), P(self.stats), P(self.hist),
                                                          history_len, self.patience, self.min_delta, P(self.ws_env), self.ws_env.numel() * 4, self.H,
                                                          self.W, P(self.y), 4, P(self.proj), P(self.env), P(self.best_env_flat), P(self.g_out), M, 0)))
        else:
            calls.append((lib.matpbr_env_phase_step, (P(self.T), P(self.light), P(self.gt_srgb), None, P(self.d_light), P(self.stats), P(self.hist),
                                                      history_len, self.patience, self.min_delta, P(self.ws_env), self.ws_env.numel() * 4, self.H,
                                                      self.W, 1)))
            self._select_at = len(calls)
            self._first_arg = 3
            calls.append((lib.matpbr_select_improved, (P(self.best_env_flat), P(self.env), P(self.stats), 0, M * 3)))
            calls.append((lib.matpbr_env_project_bwd, (P(self.y), 4, P(self.proj), P(self.d_light), P(self.g_out), 4, M)))
        # backward chain, one launch per layer (matpbr_mlp_small_bwd_step): with g = dL/d pre of a layer in hand, its weight gradient, the input
        # gradient into the layer below (+ that layer's per-tile column sums) and its own bias gradient (the fold of the column sums the launch
        # before left; for the output layer the column sums of g itself) are independent pieces of work
        self.ws_in2 = torch.empty_like(self.ws_in)
        ws2 = [self.ws_in, self.ws_in2]
        groups = (M + 31) // 32
        top = self.L - 1
        for l in range(top, -1, -1):
            gw, gb = gviews[l]
            if l == top:
                g, ldg, n_red, k_in = self.g_out, 4, self.n_last, K
                bias_src, bias_stride, bias_groups = self.g_out, 4, M
            else:
                g, ldg, n_red, k_in = self.gbufs[l], self.gbufs[l].stride(0), ns[l], Ks[l]
                bias_src, bias_stride, bias_groups = ws2[(top - l - 1) & 1], 256, groups
            if l > 0:
                wp, _ = self.views[l]
                din = (P(wp), wp.stride(0), P(self.cbufs[l - 1]), P(self.gbufs[l - 1]), self.gbufs[l - 1].stride(0), P(ws2[(top - l) & 1]), ns[l - 1])
            else:
                din = (None, 0, None, None, 0, None, 0)
            calls.append((lib.matpbr_mlp_small_bwd_step, (P(g), ldg) + din + (P(inps[l]), inps[l].stride(0), P(gw), gw.stride(0), k_in, P(bias_src), bias_stride,
                                                          bias_groups, P(gb), M, n_red)))
        # the update and the step count stop with the image (stats[13] >= 2): the reference breaks right after the stopping iteration (:250-254)
        calls.append((lib.matpbr_adamw_step_snapshot_dev, (P(self.flat), P(self.gflat), P(self.adam_m), P(self.adam_v), self.flat.numel(), P(self.hyper),
                                                           0.9, 0.999, 1e-8, 0.0, None, P(self.stats))))
        self._calls = calls
        self._first = True

    FUSED_TAIL = True       # False: matpbr_env_phase_step + matpbr_select_improved + matpbr_env_project_bwd (the same bits: tests/test_gpu_parity.py)

    # ------------------------------------------------------------------------------------------------------------------------------
    def set_lr(self, lr: float) -> None:
        self.hyper[0:1].fill_(float(lr))

    def _body(self) -> None:
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        for i, (fn, args) in enumerate(self._calls):
            if i == self._select_at and self._first:
                k = getattr(self, "_first_arg", 3)
                args = args[:k] + (1,) + args[k + 1:]           # the first iteration always snapshots (best_env starts undefined)
            code = fn(*args, stream)
            if code != 0:
                _lib.check(code, fn.__name__)
        self._first = False

    def step(self) -> None:
        with torch.cuda.device(self.dev):
            if not self.use_graph:
                self._body()
            elif self._graph is not None:
                self._graph.replay()
            elif self._warm < 3:                                 # eager iterations first (lazy initialisations, the forced first snapshot)
                self._body()
                self._warm += 1
            else:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._body()
                self._graph = graph                              # the capture itself does not execute: replay it for this iteration
                graph.replay()
        self.t += 1

    UNROLL_MAX = 32

    def step_many(self, n: int) -> None:
        """`n` iterations between two polls of the device's EarlyStopping flag, replayed as ONE hipGraph of n unrolled iterations (a replay
        per iteration leaves ~9 us of host time between two graphs -- a quarter of the texel loop's iteration; measured with
        tools/env_trace.sh).  The iterations behind a stop are the no-ops they are with `step()`: the stop lives in device memory."""
        n = int(n)
        if not self.use_graph or self._graph is None or n < 2 or n > self.UNROLL_MAX:
            for _ in range(n):
                self.step()
            return
        graphs = self.__dict__.setdefault("_unrolled", {})
        with torch.cuda.device(self.dev):
            if n not in graphs:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(n):
                        self._body()
                graphs[n] = g
            graphs[n].replay()
        self.t += n

    def head(self) -> torch.Tensor:
        """The envmap of the last executed iteration, [He, We, 3] (what `envmap_net(start_envmap)` returned in it)."""
        return self.env.view(self.env_size + (3,))

    @property
    def best_env(self) -> torch.Tensor:
        return self.best_env_flat.view(self.env_size + (3,))

    @property
    def best_img(self) -> torch.Tensor:
        """Linear render under the best-so-far envmap (SaveBest.rendered_img of the env phase, :247)."""
        light = self.scene.light_from_emitter(self.best_env).detach().reshape(1, 25, 3).contiguous()
        return ops.relight(self.T, light, self.H, self.W)[0]

    @property
    def pred(self) -> torch.Tensor:
        """Render under the current envmap (frames); not kept per iteration."""
        return ops.relight(self.T, self.light.reshape(1, 25, 3), self.H, self.W)[0]

    def poll(self) -> Dict[str, torch.Tensor]:
        st, o = self.stats.cpu(), ops
        return {"stopped": st[:, o.STAT_STOPPED] > 0.5, "iters": st[:, o.STAT_ITERS].to(torch.int64), "best_mse": st[:, o.STAT_BEST],
                "mse": st[:, o.STAT_MSE], "loss": st[:, o.STAT_LOSS]}

    def history(self) -> torch.Tensor:
        return self.hist[: self.t]


class EnvTexelPhase:
    """Hot loop A with the `--model_name none` parameterisation of the light -- the 16 x 32 texels themselves through a softplus
    (optimize.py: `env_raw`; inverse_img_w_mi.py:225-254 with the MLP replaced by its output activation) -- launch by launch on the C
    ABI: softplus + SH projection, the pass over the radiance transfer (render, loss, SaveBest / EarlyStopping, d loss / d light), the
    snapshot of the best envmap, the projection's backward and Adam with its step count and learning rate in device memory: THREE launches
    per iteration since round 4 (`matpbr_env_texel_phase_step`: the pass over the transfer, one workgroup for everything behind it, the next
    envmap's projection; seven kernels before, `FUSED_TAIL = False`), captured into a hipGraph.  (`loop.FusedEnvPhase` runs the same iteration with the head, its
    backward and the optimiser as framework ops: a dozen small launches more.)  `raw` ([He, We, 3], a leaf tensor) is updated in place
    when the phase ends (`sync_params()`), as an optimiser over it would have left it."""

    def __init__(self, scene: _render.Scene, gt_image: torch.Tensor, raw: torch.Tensor, spp: int = 64, lr: float = 1e-3, patience: int = 0,
                 min_delta: float = 0.0, best_mse: Optional[torch.Tensor] = None, history_len: int = 5000, use_graph: bool = True):
        if gt_image.ndim != 3 or raw.ndim != 3 or raw.shape[-1] != 3:
            raise NotImplementedError("EnvTexelPhase: one image, one [He, We, 3] envmap")
        self.scene, self.spp, self.raw = scene, int(spp), raw
        self.gt = gt_image.contiguous()
        dev = self.dev = self.gt.device
        self.H, self.W = self.gt.shape[0], self.gt.shape[1]
        self.env_size = tuple(raw.shape[:2])
        M = self.M = self.env_size[0] * self.env_size[1]
        if M > 1024:
            raise ValueError("at most 1024 texels")
        self.gt_srgb = _loss.linear_to_srgb(self.gt).contiguous()
        self.lib = _lib.load()
        self.use_graph, self._graph, self._warm, self.t = bool(use_graph), None, 0, 0
        self.patience, self.min_delta = int(patience), float(min_delta)
        E = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.y = E(M, 4)                                          # the parameters, rows padded to 16 bytes (the padding column stays zero)
        self.y[:, :3] = raw.detach().reshape(M, 3).to(dev, torch.float32)
        self.g = E(M, 4)
        self.adam_m, self.adam_v = E(M, 4), E(M, 4)               # a fresh Adam per phase (:225-229)
        self.hyper = torch.tensor([float(lr), 0.0], dtype=torch.float32, device=dev)
        self.env, self.best_env_flat = E(M, 3), E(M, 3)
        self.light, self.d_light = E(25, 3), E(25, 3)
        self.proj = scene.projection(*self.env_size).contiguous()
        self.stats = ops.new_loss_stats(1, dev)
        if best_mse is not None:
            self.stats[:, ops.STAT_BEST] = best_mse.to(dev).reshape(-1)
        self.hist = E(history_len, 1)
        self.ws_env = E(int(self.lib.matpbr_env_phase_workspace_bytes(self.H, self.W, 1)) // 4 + 1)
        shp = (self.H, self.W)
        self.T = ops.shade_transfer(scene.a.contiguous(), scene.r.reshape(shp + (1,)).contiguous(), scene.m.reshape(shp + (1,)).contiguous(),
                                    scene.shading_normal().contiguous(), self.spp, scene.fov)
        if scene.bg_mask is not None:    # pixels without geometry see the environment along their camera ray: their transfer is the SH basis there
            ops.background_into_transfer(self.T, self.H, self.W, scene.bg_basis)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        lib = self.lib
        self.fused_tail = bool(self.FUSED_TAIL)
        project = (lib.matpbr_env_project, (P(self.y), 4, P(self.proj), P(self.env), P(self.light), M))
        if self.fused_tail:
            # three launches per iteration (matpbr_env_texel_phase_step): the pass over the transfer, one workgroup that does everything behind it,
            # the NEXT iteration's softplus + SH projection; the first envmap is projected here, once
            with torch.cuda.device(dev):
                _lib.check(project[0](*project[1], ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "matpbr_env_project")
            self._calls = [(lib.matpbr_env_texel_phase_step,
                            (P(self.T), P(self.gt_srgb), None, P(self.d_light), P(self.stats), P(self.hist), history_len, self.patience, self.min_delta,
                             P(self.ws_env), self.ws_env.numel() * 4, self.H, self.W, P(self.y), 4, P(self.proj), P(self.env), P(self.best_env_flat),
                             P(self.light), P(self.g), P(self.adam_m), P(self.adam_v), P(self.hyper), 0.9, 0.999, 1e-8, M, 0))]
            self._select_at, self._first = -1, True
            return
        self._calls = [
            project,
            (lib.matpbr_env_phase_step, (P(self.T), P(self.light), P(self.gt_srgb), None, P(self.d_light), P(self.stats), P(self.hist), history_len,
                                         self.patience, self.min_delta, P(self.ws_env), self.ws_env.numel() * 4, self.H, self.W, 1)),
            (lib.matpbr_select_improved, (P(self.best_env_flat), P(self.env), P(self.stats), 0, M * 3)),
            (lib.matpbr_env_project_bwd, (P(self.y), 4, P(self.proj), P(self.d_light), P(self.g), 4, M)),
            # the update and the step count stop with the image (stats[13] >= 2): the reference breaks right after the stopping iteration (:250-254)
            (lib.matpbr_adamw_step_snapshot_dev, (P(self.y), P(self.g), P(self.adam_m), P(self.adam_v), self.y.numel(), P(self.hyper), 0.9, 0.999, 1e-8,
                                                  0.0, None, P(self.stats))),
        ]
        self._select_at, self._first = 2, True

    FUSED_TAIL = True       # False: the seven launches of round 3 (the same bits: tests/test_gpu_parity.py)

    def _body(self) -> None:
        if not self.fused_tail:
            return EnvMlpPhase._body(self)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        fn, args = self._calls[0]
        if self._first:
            args = args[:-1] + (1,)                             # the first iteration always snapshots (best_env starts undefined)
        _lib.check(fn(*args, stream), "matpbr_env_texel_phase_step")
        self._first = False

    set_lr = EnvMlpPhase.set_lr
    step = EnvMlpPhase.step
    step_many, UNROLL_MAX = EnvMlpPhase.step_many, EnvMlpPhase.UNROLL_MAX
    poll = EnvMlpPhase.poll
    history = EnvMlpPhase.history
    best_env = EnvMlpPhase.best_env
    best_img = EnvMlpPhase.best_img
    pred = EnvMlpPhase.pred

    def head(self) -> torch.Tensor:
        """The envmap of the CURRENT parameters, [He, We, 3]: with the fused tail what the next iteration will render under (the tail's last
        act is the next softplus + projection); with the seven launches the envmap of the last executed iteration."""
        return self.env.view(self.env_size + (3,))

    def sync_params(self) -> None:
        """Write the parameters back into the caller's tensor (call when the phase is over)."""
        with torch.no_grad():
            self.raw.copy_(self.y[:, :3].reshape(self.raw.shape))
