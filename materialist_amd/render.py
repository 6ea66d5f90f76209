"""Operator face of the hot path: `load_estimated_mesh` / `render_envmap` / `render_w_brdf` with the
reference's signatures (inverse_img_w_mi.py:30-80) on top of libmatpbr.so.

The reference keeps geometry, camera, current light and current materials inside a Mitsuba scene and
mutates it through `mi.traverse(scene)` (keys `shape.bsdf.a|r|m|n`, `shape.bsdf.use_mesh_normal`,
`emitter.data`; inverse_img_w_mi.py:63,73-77,217-219,334-339).  `Scene` below is the lightweight
stand-in: same keys, same statefulness (the arguments of one render persist for the next).

Lighting: the reference pushes a [16,32,3] texel map into the envmap emitter.  Here the light the
kernels integrate is its order-4 SH projection (25 coefficients per channel, the convention of
myutils/computeSH.py); `emitter.data` may be set either to texels [He,We,3] (projected with a fixed
25 x He*We matrix, differentiable) or directly to SH coefficients [25,3].
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from . import sh as _sh

DEFAULT_FOV = 35.0  # inverse_img_w_mi.py:32, myutils/default_cam.json


# Gradient convention of the operator face w.r.t. the roughness.  False (default): the sampled directions and their pdfs are
# constants of the backward pass (stop-gradient), which is what the fused loops use.  True: d_r is the exact derivative of the
# rendered value through the GGX quadrature nodes -- the convention of the live reference, whose sampler is differentiable in the
# roughness and whose weight divides by an attached pdf (myutils/mi_plugin.py:227-230,1335-1341).  MATPBR_FLAG_ATTACHED_SAMPLING.
ATTACHED_SAMPLING = False


class _ShadeFn(torch.autograd.Function):
    """Differentiable R(a, r, m, n, light) -> rgb; stands in for dr.wrap_ad + mi.render (inverse_img_w_mi.py:59-80)."""

    @staticmethod
    def forward(ctx, a, r, m, n, light, spp, fov, workspace_holder):
        a, r, m, n, light = (t.contiguous() for t in (a, r, m, n, light))
        # the material gradients are closed forms of sums the forward pass forms anyway: keep them (9 floats per pixel) and the
        # backward pass is one streaming kernel
        jac = ops.plane9(a) if any(ctx.needs_input_grad[:3]) else None
        out = ops.shade_fwd(a, r, m, n, light, spp, fov, jac=jac)
        ctx.save_for_backward(a, r, m, n, light)
        ctx.spp, ctx.fov, ctx.ws, ctx.jac = spp, fov, workspace_holder, jac
        return out

    @staticmethod
    def backward(ctx, d_out):
        a, r, m, n, light = ctx.saved_tensors
        need = ctx.needs_input_grad
        want_mat = need[0] or need[1] or need[2]
        want_n, want_light = need[3], need[4]
        d_out = d_out.contiguous()
        d_a = d_r = d_m = d_n = d_l = None
        if want_mat and ATTACHED_SAMPLING:       # d_r through the GGX quadrature nodes (the live reference's convention)
            d_a, d_r, d_m, _, _ = ops.shade_bwd(a, r, m, n, light, d_out, ctx.spp, ctx.fov, want_mat=True, attached=True)
        elif want_mat:
            d_a, d_r, d_m = ops.shade_bwd_jac(a, r, m, ctx.jac, d_out)
        if want_n or want_light:
            ws = None
            if want_light and ctx.ws is not None:
                ws = ctx.ws["ws"] = ops.workspace_for(a, ctx.ws.get("ws"))
            _, _, _, d_n, d_l = ops.shade_bwd(a, r, m, n, light, d_out, ctx.spp, ctx.fov, want_mat=False, want_n=want_n,
                                              want_light=want_light, workspace=ws)
        return (d_a if need[0] else None, d_r.reshape(r.shape) if need[1] else None, d_m.reshape(m.shape) if need[2] else None,
                d_n if need[3] else None, d_l if need[4] else None, None, None, None)


# What the operator face keeps between calls (the scene is stateful -- "the arguments of one call persist for the next").  The
# reference's loop calls render_w_brdf thousands of times under ONE light and ONE set of shading normals (inverse_img_w_mi.py:317-342,
# 384-386, 513-515), so everything that depends on (normals, view, light) alone is computed once per (light, normals) VERSION -- tensor
# identity + autograd version counter, so an in-place optimiser step on either invalidates it -- and reused:
#   "none"     every call walks both lobes of every pixel (rounds 1-3)
#   "diffuse"  the nine diffuse-lobe coefficients per pixel (matpbr_diffuse_cache): calls walk the GGX samples only
#   "lazy"     (default) additionally the per-pixel local models in the roughness of matpbr_shade_fwd_lazy, for calls that need material
#              gradients only: the forward pass renders from the models (pixels whose roughness left its model's interval are re-sampled:
#              |render - exact| <= 1e-3 max(|exact|, mean|exact|), tests/test_gpu_lazy.py), the backward pass streams the jac planes it leaves
# Calls that need d/d normal or d/d light, and scenes with ATTACHED_SAMPLING, take the round-3 path.
OPERATOR_CACHE = "lazy"


class _ShadeLazyFn(torch.autograd.Function):
    """R(a, r, m | n, light fixed) from the scene's cached models; gradients w.r.t. the materials only."""

    @staticmethod
    def forward(ctx, a, r, m, n, light, spp, fov, cache):
        a, r, m = (t.contiguous() for t in (a, r, m))
        out, jac = ops.shade_fwd_lazy(a, r, m, n, light, spp, cache["dcache"], cache["state"], force=cache["fresh"], stats=cache["stats"],
                                      fov_x_deg=fov, jac32=True)
        if cache["fresh"]:
            # the parity floor of later calls: half the mean radiance of this (exact) render, per image, kept on the device
            B = cache["stats"].shape[0]
            cache["stats"][:, ops.STAT_GT_SUM] = out.reshape(B, -1).sum(dim=1)
            cache["fresh"] = False
        ctx.save_for_backward(a, r, m)
        ctx.jac = jac
        return out

    @staticmethod
    def backward(ctx, d_out):
        a, r, m = ctx.saved_tensors
        need = ctx.needs_input_grad
        d_a, d_r, d_m = ops.shade_bwd_jac(a, r, m, ctx.jac, d_out.contiguous())
        return (d_a if need[0] else None, d_r.reshape(r.shape) if need[1] else None, d_m.reshape(m.shape) if need[2] else None,
                None, None, None, None, None)


class SceneParameters(dict):
    """`mi.traverse(scene)` stand-in: assignment stores, `update()` is accepted for source compatibility."""

    def __init__(self, scene: "Scene"):
        super().__init__()
        self._scene = scene
        for k in Scene.KEYS:
            dict.__setitem__(self, k, scene._get(k))

    def __setitem__(self, key, value):
        if key not in Scene.KEYS:
            raise KeyError(f"unknown scene parameter {key!r}; valid keys: {Scene.KEYS}")
        self._scene._set(key, value)
        dict.__setitem__(self, key, value)

    def update(self, *a, **k):  # params.update() after assignments (inverse_img_w_mi.py:64,78)
        for key, val in dict(*a, **k).items():
            self[key] = val
        return []


class Scene:
    """Geometry (per-pixel geometric normal from depth), pinhole camera, current light and materials."""

    KEYS = ("shape.bsdf.a", "shape.bsdf.r", "shape.bsdf.m", "shape.bsdf.n", "shape.bsdf.use_mesh_normal", "emitter.data")

    def __init__(self, height: int, width: int, device, geo_normal: Optional[torch.Tensor] = None, use_mesh_normal: bool = True,
                 fov_x_deg: float = DEFAULT_FOV, batch: int = 1, env_size=(16, 32)):
        self.H, self.W, self.B = int(height), int(width), int(batch)
        self.device = torch.device(device)
        self.fov = float(fov_x_deg)
        self.use_mesh_normal = bool(use_mesh_normal)
        shp = (self.B, self.H, self.W) if self.B > 1 else (self.H, self.W)
        full = lambda c, v: torch.full(shp + (c,), v, dtype=torch.float32, device=self.device)
        # MatDiffBSDF defaults: every map 0.5 (myutils/mi_plugin.py:1238-1241)
        self.a, self.r, self.m, self.n = full(3, 0.5), full(1, 0.5), full(1, 0.5), full(3, 0.5)
        if geo_normal is None:
            geo_normal = torch.zeros(shp + (3,), dtype=torch.float32, device=self.device)
            geo_normal[..., 2] = 1.0  # fronto-parallel plane facing the camera
        self.geo_normal = geo_normal.to(self.device, torch.float32).contiguous()
        self.env_size = tuple(env_size)
        self._proj: Dict[tuple, torch.Tensor] = {}
        lshape = (self.B, _sh.NSH, 3) if self.B > 1 else (_sh.NSH, 3)
        self.light = torch.zeros(lshape, dtype=torch.float32, device=self.device)
        self.light[..., 0, :] = float(np.sqrt(4 * np.pi))  # unit white radiance until the caller sets emitter.data
        self.emitter_data = self.light
        self._ws = {"ws": None}
        self._cache: Optional[dict] = None               # OPERATOR_CACHE: what was computed for the (light, normals) objects of `_cache["src"]`
        self._seen: Optional[dict] = None                # the pair seen by the last call that found no cache (built when it comes back)
        self.cache_builds = 0                            # how many times it was (re)built (tests, profiling)
        self.bg_mask: Optional[torch.Tensor] = None      # [H,W] bool: pixels without geometry (mesh_mask.png)
        self.bg_basis: Optional[torch.Tensor] = None     # [H*W,25]: Y_k(camera ray) on those pixels, 0 elsewhere

    # -- pixels without geometry ---------------------------------------------------------------------
    def set_mesh_mask(self, mask: Optional[torch.Tensor]) -> None:
        """`mesh_mask.png` (inverse_img_w_mi.py:713-724: `depth[mesh_mask] = 0`, no triangles there): the camera ray of such a
        pixel leaves the scene and sees the environment emitter.  With SH lighting that radiance is sum_k light[k] Y_k(ray), linear
        in the light: the render composes it over the shaded image with `torch.where`, so the light receives its gradient from
        these pixels and the materials none.  One [H,W] mask, or [B,H,W] for a batch; the fused loops take the same pixels as
        constant models (loop.FusedBrdfPhase) and as SH-basis rows of the radiance transfer (loop.FusedEnvPhase)."""
        if mask is None or not bool(mask.any()):
            self.bg_mask = self.bg_basis = None
            return
        shp = (self.B, self.H, self.W) if self.B > 1 else (self.H, self.W)
        mask = mask.to(self.device).bool().reshape(shp)
        f = (self.W / 2.0) / np.tan(np.radians(self.fov) / 2.0)       # pinhole of the kernels
        cx, cy = (self.W - 1) / 2.0, (self.H - 1) / 2.0
        i, j = np.meshgrid(np.arange(self.H, dtype=np.float64), np.arange(self.W, dtype=np.float64), indexing="ij")
        ray = np.stack([(j - cx) / f, -(i - cy) / f, -np.ones_like(i)], -1)
        ray /= np.linalg.norm(ray, axis=-1, keepdims=True)
        Y = torch.from_numpy(_sh.sh_basis(ray).reshape(self.H * self.W, _sh.NSH)).to(self.device, torch.float32)
        self.bg_mask = mask
        self.bg_basis = (Y * mask.reshape(shp[:-2] + (self.H * self.W, 1))).contiguous()        # [H*W,25] or [B,H*W,25]

    def background_radiance(self, light: torch.Tensor) -> torch.Tensor:
        """Radiance the masked pixels see along their camera rays, [(B,)H,W,3]; linear in the [25,3] light(s)."""
        if self.B > 1 and light.ndim == 2:
            light = light.unsqueeze(0).expand(self.B, -1, -1)
        return (self.bg_basis @ light).reshape(self.bg_mask.shape + (3,))

    # -- mi.traverse face ----------------------------------------------------------------------------
    def _get(self, key):
        return {"shape.bsdf.a": self.a, "shape.bsdf.r": self.r, "shape.bsdf.m": self.m, "shape.bsdf.n": self.n,
                "shape.bsdf.use_mesh_normal": self.use_mesh_normal, "emitter.data": self.emitter_data}[key]

    def _set(self, key, value):
        if key == "shape.bsdf.use_mesh_normal":
            self.use_mesh_normal = bool(value)
        elif key == "emitter.data":
            light = self.light_from_emitter(value)           # texels: a fresh projection every time (they may have been stepped in place)
            if light is not self.light:                      # another light: what was kept for the old one goes (also when no render came in between)
                self.invalidate_cache()
            self.emitter_data, self.light = value, light
        else:
            if key == "shape.bsdf.n" and value is not self.n and not self.use_mesh_normal:
                self.invalidate_cache()
            setattr(self, key.rsplit(".", 1)[1], value)

    # -- lighting ------------------------------------------------------------------------------------
    def projection(self, He: int, We: int) -> torch.Tensor:
        key = (He, We)
        if key not in self._proj:
            self._proj[key] = torch.from_numpy(_sh.envmap_to_sh_matrix(He, We)).to(self.device, torch.float32)
        return self._proj[key]

    def light_from_emitter(self, data: torch.Tensor) -> torch.Tensor:
        """[He,We,3] texels -> SH25 by the fixed projection matrix (differentiable); [25,3] passes through."""
        if data.shape[-2:] == (_sh.NSH, 3):
            return data
        if data.ndim < 3 or data.shape[-1] != 3:
            raise ValueError(f"emitter.data must be [He,We,3] texels or [25,3] SH coefficients, got {tuple(data.shape)}")
        He, We = data.shape[-3], data.shape[-2]
        flat = data.reshape(data.shape[:-3] + (He * We, 3))
        return self.projection(He, We) @ flat

    # -- render --------------------------------------------------------------------------------------
    def shading_normal(self) -> torch.Tensor:
        # use_mesh_normal=True shades with the geometric normal, not the n map (F10; mi_plugin.py:1386-1389)
        return self.geo_normal if self.use_mesh_normal else self.n

    def invalidate_cache(self) -> None:
        """Forget what the operator face keeps per (light, normals).  For callers that write the light or the normal map through raw pointers
        (this library's own kernels, `.data`), which no version counter sees."""
        self._cache = None
        self._seen = None

    def _cached(self, light_src: torch.Tensor, light: torch.Tensor, nrm: torch.Tensor, spp: int) -> Optional[dict]:
        """The per-(light, normals) cache, rebuilt when either changed or `spp` did.  "Changed" is decided on the tensor OBJECTS the scene holds
        (`light_src`: `self.light` before any broadcast; `nrm`) -- the cache keeps strong references to them, so their storage cannot be freed
        and handed to another tensor at the same address -- plus their autograd version counters (an in-place optimiser step).  None when the
        light or the normals take part in autograd (they are not constants of the call) or caching is off."""
        if OPERATOR_CACHE == "none" or ATTACHED_SAMPLING or light.requires_grad or nrm.requires_grad or not light.is_cuda:
            return None
        tag = (light_src._version, tuple(light.shape), nrm._version, int(spp), self.fov)

        def same(entry):
            return entry is not None and entry["src"][0] is light_src and entry["src"][1] is nrm and entry["tag"] == tag

        if same(self._cache):
            return self._cache
        if not same(self._seen):           # a light used once (an env-phase iteration, a relit frame) is not worth nine planes per pixel:
            self._seen = {"src": (light_src, nrm), "tag": tag}      # the cache is built when the same (light, normals) come back
            self._cache = None
            return None
        lc, nc = light.detach().contiguous(), nrm.detach().contiguous()
        stats = ops.new_loss_stats(self.B, self.device)
        stats[:, ops.STAT_RATIO] = 1.0
        stats[:, ops.STAT_GT_SUM] = 1e-6          # first (forced) build: every interval against the render itself (a conservative floor)
        self._cache = {"src": (light_src, nrm), "tag": tag, "light": lc, "n": nc, "dcache": ops.diffuse_cache(nc, lc, int(spp), self.fov),
                       "state": None, "stats": stats, "fresh": True}
        self.cache_builds += 1
        return self._cache

    def render(self, spp: int) -> torch.Tensor:
        shp = (self.B, self.H, self.W) if self.B > 1 else (self.H, self.W)
        light = self.light
        if self.B > 1 and light.ndim == 2:
            light = light.unsqueeze(0).expand(self.B, -1, -1)
        r = self.r.reshape(shp + (1,))
        m = self.m.reshape(shp + (1,))
        nrm = self.shading_normal()
        cache = self._cached(self.light, light, nrm, spp)
        want_mat = torch.is_grad_enabled() and any(t.requires_grad for t in (self.a, r, m))
        if cache is not None and want_mat and OPERATOR_CACHE == "lazy":
            if cache["state"] is None:
                cache["state"] = ops.lazy_state(cache["n"])
            img = _ShadeLazyFn.apply(self.a, r, m, cache["n"], cache["light"], int(spp), self.fov, cache)
        elif cache is not None and not want_mat:
            # no gradient asked for: the exact render, GGX samples only (the diffuse lobe from its cached coefficients)
            img = ops.shade_fwd(self.a.detach().contiguous(), r.detach().contiguous(), m.detach().contiguous(), cache["n"], cache["light"],
                                int(spp), self.fov, dcache=cache["dcache"])
        else:
            img = _ShadeFn.apply(self.a, r, m, nrm, light, int(spp), self.fov, self._ws)
        if self.bg_mask is not None:
            img = torch.where(self.bg_mask.unsqueeze(-1), self.background_radiance(light), img)
        return img


def traverse(scene: Scene) -> SceneParameters:
    return SceneParameters(scene)


def load_estimated_mesh(depth: Optional[torch.Tensor], use_mesh_normal: bool, max_path: int = 4, height: int = 512, width: int = 512,
                        device="cuda", fov_x_deg: float = DEFAULT_FOV, batch: int = 1, mesh_mask: Optional[torch.Tensor] = None,
                        geometry: str = "depth") -> Scene:
    """Counterpart of load_estimated_mesh(mesh_path, use_mesh_normal, max_path) (inverse_img_w_mi.py:30-56).
    The reference loads a .ply triangulated from depth; the per-pixel build needs only the geometric normal
    of the heightfield, computed on the GPU from `depth` [H,W] (or [B,H,W]).  `max_path` is accepted for
    signature compatibility: the deterministic render is direct lighting only."""
    del max_path
    geo = None
    if depth is not None:
        depth = depth.to(device, torch.float32)
        height, width = depth.shape[-2], depth.shape[-1]
        batch = depth.shape[0] if depth.ndim == 3 else 1
        if geometry == "mesh":
            # the reference's own mesh of this depth map (gap closing at depth edges included): per-pixel normal = area-weighted normal of
            # the pixel's grid vertex; a pixel whose vertex carries no triangle has no geometry (its camera ray sees the environment)
            from . import mesh as _mesh

            d_host = depth.detach().cpu().numpy().astype(np.float32).reshape(batch, height, width).copy()
            if mesh_mask is not None:
                d_host[mesh_mask.cpu().numpy().astype(bool).reshape(d_host.shape)] = 0.0       # inverse_img_w_mi.py:723
            rms = [_mesh.reference_mesh(d, fov_x_deg) for d in d_host]                          # host pass, milliseconds per image
            geo = torch.from_numpy(np.stack([rm["normals"] for rm in rms])).to(device)
            holes = torch.from_numpy(np.stack([~rm["has_faces"] for rm in rms]))
            mesh_mask = holes if mesh_mask is None else (mesh_mask.cpu().bool().reshape(holes.shape) | holes)
            geo[holes.to(device)] = torch.tensor([0.0, 0.0, 1.0], device=device)               # any unit vector: these pixels are never shaded
            if depth.ndim == 2:
                geo, mesh_mask = geo[0], mesh_mask[0]
        elif geometry == "depth":
            geo = ops.normals_from_depth(depth.contiguous(), fov_x_deg)
        else:
            raise ValueError("geometry: 'depth' (central differences of the depth map, on the GPU) or 'mesh' (the reference's mesher, on the host)")
    scene = Scene(height, width, device, geo, use_mesh_normal, fov_x_deg, batch)
    if mesh_mask is not None:
        scene.set_mesh_mask(mesh_mask)
    return scene


def render_envmap(scene: Scene, envmap: torch.Tensor, spp: int = 64) -> torch.Tensor:
    """inverse_img_w_mi.py:59-67: set `emitter.data`, render with the materials already in the scene."""
    params = traverse(scene)
    params["emitter.data"] = envmap
    params.update()
    return scene.render(spp)


def render_w_brdf(scene: Scene, albedo: torch.Tensor, roughness: torch.Tensor, metallic: torch.Tensor,
                  normal: Optional[torch.Tensor] = None, spp: int = 64) -> torch.Tensor:
    """inverse_img_w_mi.py:69-80: set `shape.bsdf.a|r|m[|n]`, render with the light already in the scene."""
    params = traverse(scene)
    params["shape.bsdf.a"] = albedo
    params["shape.bsdf.r"] = roughness
    params["shape.bsdf.m"] = metallic
    if normal is not None:
        params["shape.bsdf.n"] = normal
    params.update()
    return scene.render(spp)
