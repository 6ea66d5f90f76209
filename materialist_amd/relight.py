"""Forward-only re-rendering of an optimised scene under new lighting: the `render_final.py` side of the reference
(`render_w_mi` :148-203, `render_real` :241-260, `rotate_envmap` :290-298, `render_rolling_envmap` :300-418).  SURVEY.md 8(f4).

The reference re-traces the scene for every light (10 x spp 64 + OptiX denoise, or spp 32 per rolling frame).  The
deterministic render is linear in the light, so the per-pixel transfer is computed once (`matpbr_shade_transfer`) and each
frame is a 75-term dot product per pixel (`matpbr_relight`, HBM-bound).  Rolling the envmap by whole texel columns is the
SH rotation about +y by the same angle (`sh.rotate_y_matrix`), so no envmap is ever re-projected or written to disk.
Material editing inside `best_results/mask.png` (`edit=` of `render_w_mi`, :143-181) is applied to the maps before the
transfer is computed.  Object insertion (`--mode oi`, :100-141,207-237) needs extra meshes and is not part of this build.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import loss as _loss
from . import ops
from . import sh as _sh
from .imageio_exr import read_exr, write_exr
from .imageio_hdr import read_hdr
from .pipeline import OUT_DIR, load_image, write_png


def load_estimated_brdf(root_dir: str, device="cuda") -> Dict[str, torch.Tensor]:
    """myutils/mi_plugin.py:701-739: best_results/{albedo,roughness,metallic,normal}.exr (+ envmap.hdr); roughness is rescaled
    `r * 0.95 + 0.05` on reload (:716)."""
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
    mat = {"albedo": t(read_exr(os.path.join(root_dir, "albedo.exr"))),
           "roughness": t(read_exr(os.path.join(root_dir, "roughness.exr"))[..., :1]) * 0.95 + 0.05,
           "metallic": t(read_exr(os.path.join(root_dir, "metallic.exr"))[..., :1]),
           "normal": t(read_exr(os.path.join(root_dir, "normal.exr")))}
    p = os.path.join(root_dir, "envmap.hdr")
    if os.path.exists(p):
        mat["envmap"] = t(read_hdr(p))
    p = os.path.join(root_dir, "mask.png")                            # "load mask for Material editing" (:729-733)
    if os.path.exists(p):
        from PIL import Image

        m = np.asarray(Image.open(p))
        mat["mask"] = torch.from_numpy(np.ascontiguousarray((m[..., 0] if m.ndim == 3 else m) > 0)).to(device)
    return mat


def rgb_to_hsv(rgb: np.ndarray) -> np.ndarray:
    """[...,3] in [0,1] -> h, s, v in [0,1] (the convention of skimage.color.rgb2hsv the reference edits with, :143-146)."""
    rgb = np.asarray(rgb, dtype=np.float64)
    v = rgb.max(-1)
    delta = v - rgb.min(-1)
    s = np.where(v > 0, delta / np.where(v > 0, v, 1), 0.0)
    d = np.where(delta > 0, delta, 1.0)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    h = np.where(v == r, (g - b) / d, np.where(v == g, 2.0 + (b - r) / d, 4.0 + (r - g) / d))
    h = np.where(delta > 0, (h / 6.0) % 1.0, 0.0)
    return np.stack([h, s, v], -1)


def hsv_to_rgb(hsv: np.ndarray) -> np.ndarray:
    hsv = np.asarray(hsv, dtype=np.float64)
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    i = np.floor(h * 6.0)
    f = h * 6.0 - i
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    i = i.astype(np.int64) % 6
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.stack([r, g, b], -1)


def apply_edit(mat: Dict[str, torch.Tensor], edit: Optional[Dict[str, object]]) -> str:
    """render_final.py:143-181: inside the mask, shift the albedo in HSV (`edit['albedo']` = [dh, ds, dv], clipped to [0,1]) and/or
    overwrite roughness / metallic with a constant.  Returns the reference's file-name flag (`_r_0.2`, ...); the albedo flag carries
    the first shift component (the reference's own expression for it, `edit[key].tolist()[0,0]`, raises a TypeError)."""
    flag = ""
    for key in ("albedo", "roughness", "metallic"):
        val = (edit or {}).get(key)
        if val is None:
            continue
        if "mask" not in mat:
            raise FileNotFoundError("Unable to edit img, no mask found")
        mask = mat["mask"]
        if key == "albedo":
            shift = np.asarray(val, dtype=np.float64).reshape(-1)[:3]
            sel = mat[key][mask].cpu().numpy()
            out = hsv_to_rgb(np.clip(rgb_to_hsv(sel) + shift, 0, 1))
            mat[key][mask] = torch.from_numpy(out.astype(np.float32)).to(mat[key].device)
            flag += f"_a_{shift[0]}"
        else:
            mat[key][mask] = float(val)
            flag += f"_{key[:1]}_{val}"
    return flag


def find_envmap(save_name: str, env_path: Optional[str], input_path: Optional[str]) -> str:
    """render_final.py:243-259 / :304-321: explicit path, else <input_path>/<name>/best_results/envmap.hdr, else the default tree."""
    if env_path is not None:
        return env_path
    cands = []
    if input_path is not None:
        cands.append(os.path.join(input_path, save_name, "best_results", "envmap.hdr"))
    cands.append(os.path.join(OUT_DIR, save_name, "best_results", "envmap.hdr"))
    for c in cands:
        if os.path.exists(c):
            return c
    raise ValueError("No envmap found")


def envmap_to_light(env: np.ndarray) -> np.ndarray:
    He, We = env.shape[:2]
    return _sh.envmap_to_sh_matrix(He, We) @ env.reshape(He * We, 3).astype(np.float64)


class Relighter:
    """Transfer of one optimised scene; `frames(lights)` renders any number of lights."""

    def __init__(self, mat: Dict[str, torch.Tensor], shading_normal: torch.Tensor, spp: int = 64, fov_x_deg: float = 35.0,
                 mesh_mask: Optional[torch.Tensor] = None):
        self.H, self.W = mat["albedo"].shape[0], mat["albedo"].shape[1]
        self.T = ops.shade_transfer(mat["albedo"].contiguous(), mat["roughness"].contiguous(), mat["metallic"].contiguous(),
                                    shading_normal.contiguous(), spp, fov_x_deg)
        self.bg = None
        if mesh_mask is not None and bool(mesh_mask.any()):        # pixels without geometry show the environment (mesh_mask.png)
            from .render import Scene

            sc = Scene(self.H, self.W, self.T.device, fov_x_deg=fov_x_deg)
            sc.set_mesh_mask(mesh_mask)
            self.bg = (sc.bg_mask, sc.bg_basis)

    def frames(self, lights) -> torch.Tensor:
        L = torch.as_tensor(np.asarray(lights, dtype=np.float32)).reshape(-1, 25, 3).to(self.T.device)
        out = ops.relight(self.T, L, self.H, self.W)
        if self.bg is not None:
            mask, basis = self.bg
            bg = torch.einsum("pk,fkc->fpc", basis, L).reshape(-1, self.H, self.W, 3)
            out = torch.where(mask[None, :, :, None], bg, out)
        return out


def _mesh_mask(scene_dir: str) -> Optional[torch.Tensor]:
    p = os.path.join(scene_dir, "mesh_mask.png")
    if not os.path.exists(p):
        return None
    from PIL import Image

    mk = np.asarray(Image.open(p))
    return torch.from_numpy(np.ascontiguousarray((mk[..., 0] if mk.ndim == 3 else mk) > 0))


def _scene_normal(scene_dir: str, mat: Dict[str, torch.Tensor], save_name: str, device) -> torch.Tensor:
    if "mn" in save_name:                                            # 'mn' in the name = optimised normals (:155-160)
        return mat["normal"]
    depth = read_exr(os.path.join(scene_dir, "depthPred.exr"))[..., 0]
    depth = 2 * depth.max() - depth                                  # inverse_img_w_mi.py:722
    return ops.normals_from_depth(torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32)).to(device))


def render_real(save_name: str, env_path: Optional[str] = None, input_path: Optional[str] = None, save_path: Optional[str] = None,
                spp: int = 64, device="cuda", edit: Optional[Dict[str, object]] = None) -> str:
    """render_final.py:148-203,241-260: one re-render under `env_path` -> mi_<name>_<env>_<edit flag>.exr / .png."""
    scene_dir = os.path.join(input_path if input_path is not None else OUT_DIR, save_name)
    env_path = find_envmap(save_name, env_path, input_path)
    mat = load_estimated_brdf(os.path.join(scene_dir, "best_results"), device)
    edit_flag = apply_edit(mat, edit)
    rl = Relighter(mat, _scene_normal(scene_dir, mat, save_name, device), spp, mesh_mask=_mesh_mask(scene_dir))
    img = rl.frames(envmap_to_light(load_image(env_path))[None])[0]
    env_id = os.path.basename(env_path)[:-4]
    out_dir = os.path.join(save_path if save_path else OUT_DIR, save_name)
    os.makedirs(out_dir, exist_ok=True)
    base = os.path.join(out_dir, f"mi_{save_name}_{env_id}_{edit_flag}")   # (:199-202)
    write_exr(base + ".exr", img.cpu().numpy())
    write_png(base + ".png", _loss.linear_to_srgb(img.clamp_min(0)).cpu().numpy())
    return base + ".png"


def render_rolling_envmap(save_name: str, env_path: Optional[str], frames: int = 36, rotation_step: float = 10.0,
                          input_path: Optional[str] = None, save_path: Optional[str] = None, spp: int = 64, device="cuda",
                          write_frames: bool = True, edit: Optional[Dict[str, object]] = None) -> Dict[str, object]:
    """render_final.py:300-418: `frames` renders, the envmap rolled by int(angle/360*W) columns per frame."""
    scene_dir = os.path.join(input_path if input_path is not None else OUT_DIR, save_name)
    env_path = find_envmap(save_name, env_path, input_path)
    env = load_image(env_path)
    We = env.shape[1]
    light0 = envmap_to_light(env)
    lights = []
    for f in range(frames):
        shift = int((f * rotation_step / 360.0) * We)                # rotate_envmap (:290-298)
        lights.append(_sh.rotate_y_matrix(2 * np.pi * shift / We) @ light0)
    mat = load_estimated_brdf(os.path.join(scene_dir, "best_results"), device)
    apply_edit(mat, edit)
    rl = Relighter(mat, _scene_normal(scene_dir, mat, save_name, device), spp, mesh_mask=_mesh_mask(scene_dir))
    out_dir = os.path.join(save_path if save_path else OUT_DIR, save_name)
    anim_dir = os.path.join(out_dir, "rolling_envmap_animation")
    os.makedirs(anim_dir, exist_ok=True)
    env_id = os.path.basename(env_path)[:-4]
    paths: List[str] = []
    imgs = []
    for f0 in range(0, frames, 24):                                  # matpbr_relight's chunk: the transfer is read once per 24 frames
        batch = rl.frames(np.stack(lights[f0:f0 + 24]))
        if write_frames:
            srgb = _loss.linear_to_srgb(batch.clamp_min(0)).clamp(0, 1).cpu().numpy()
            for k in range(srgb.shape[0]):
                path = os.path.join(anim_dir, f"frame_{f0 + k:04d}.png")
                write_png(path, srgb[k])
                paths.append(path)
                imgs.append((srgb[k] * 255 + 0.5).astype(np.uint8))
    gif_path = None
    if imgs:
        from PIL import Image

        from .video_mp4 import write_mp4

        gif_path = os.path.join(out_dir, f"rolling_envmap_{save_name}_{env_id}.gif")     # render_final.py:405-414 writes both
        pil = [Image.fromarray(i) for i in imgs]
        pil[0].save(gif_path, save_all=True, append_images=pil[1:], duration=100, loop=0)
        mp4_path = write_mp4(os.path.join(out_dir, f"rolling_envmap_{save_name}_{env_id}.mp4"), imgs, fps=10)
    return {"animation_dir": anim_dir, "frames": paths, "gif": gif_path, "mp4": mp4_path if imgs else None}
