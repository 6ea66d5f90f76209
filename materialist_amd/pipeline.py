"""Pipeline head and output layout of `inverse_img_w_mi.py` (inverse_image :623-770, get_output_dir :82-104, the
writers of optimize_envmap_ARMN :257-303,588-599) on top of `optimize.optimize_envmap_ARMN`.  SURVEY.md section 8(f1).

The initial maps come from MaterialNet (`matnet_weights` = the reference's `matnet_weights.pth`, which the reference
downloads from the HF hub, :648-654; there is no network here), or from `pred_dir` (files in the layout the reference
writes: albedoPred.exr, normalPred.exr, roughnessPred.png, metallicPred.png, depthPred.exr), or, failing both, from a
flat prior (albedo = the linearised image, roughness 0.5, metallic 0, fronto-parallel plane).  The reference turns the
collected frames into env_optimization.mp4 / mat_optimization.mp4 (`create_video_from_frames`, :593-612) through imageio's ffmpeg
plugin; no video encoder exists in this image, so the same frames are JPEG-coded and muxed by `video_mp4.write_mp4` (Motion-JPEG
in MP4) under the same file names; the PNGs are kept.
"""
from __future__ import annotations

import json
import os
import time
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import loss as _loss
from .imageio_exr import read_exr, write_exr
from .imageio_hdr import read_hdr, write_hdr

BASE_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = os.path.join(BASE_DIR, "output_imgs")        # global_config.py:1-4


def get_output_dir(save_name: str, save_path: Optional[str] = None) -> str:
    """inverse_img_w_mi.py:82-104."""
    if save_path:
        if os.path.isabs(save_path):
            return os.path.join(save_path, save_name)
        return os.path.join(OUT_DIR, save_path, save_name)
    return save_name if os.path.isabs(save_name) else os.path.join(OUT_DIR, save_name)


def center_crop_and_resize(arr: np.ndarray, target_size=(512, 512)) -> np.ndarray:
    """myutils/misc.py:10-34: centre crop to a square, drop alpha, uint8 -> [0,1], bilinear resize (align_corners=True)."""
    h, w, _ = arr.shape
    d = min(h, w)
    sh, sw = (h - d) // 2, (w - d) // 2
    crop = arr[sh:sh + d, sw:sw + d, :3]
    if arr.dtype == np.uint8:
        crop = crop.astype(np.float32) / 255.0
    elif arr.dtype in (np.float32, np.float16):
        crop = crop.astype(np.float32)
    else:
        raise ValueError("Unsupported data type, only uint8 and float16/32 are supported.")
    t = torch.from_numpy(np.ascontiguousarray(crop)).permute(2, 0, 1).unsqueeze(0).to(torch.float32)
    t = F.interpolate(t, size=tuple(target_size), mode="bilinear", align_corners=True)
    return t.squeeze(0).permute(1, 2, 0).numpy()


def load_image(path: str) -> np.ndarray:
    if path.endswith(".exr"):
        return read_exr(path)
    if path.endswith(".hdr"):
        return read_hdr(path)
    from PIL import Image

    return np.array(Image.open(path))


def _srgb_encode(x: np.ndarray) -> np.ndarray:
    """True sRGB transfer curve, what a PNG written from linear data carries (mi.util.write_bitmap, :677-678)."""
    x = np.clip(x, 0.0, 1.0)
    return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)


def write_png(path: str, img: np.ndarray, linear: bool = False) -> None:
    from PIL import Image

    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[..., 0]
    if linear:
        img = _srgb_encode(img)
    Image.fromarray((np.clip(img, 0, 1) * 255.0 + 0.5).astype(np.uint8)).save(path)


def _srgb_decode(x: np.ndarray) -> np.ndarray:
    return np.where(x <= 0.04045, x / 12.92, np.power((x + 0.055) / 1.055, 2.4)).astype(np.float32)


def flat_prior(img_linear: np.ndarray) -> Dict[str, np.ndarray]:
    """Stand-in for MaterialNet.infer_image (dpt.py:219-241) until f3: no learned prior."""
    H, W, _ = img_linear.shape
    normal = np.zeros((H, W, 3), np.float32)
    normal[..., 2] = 1.0
    return {"albedo": np.clip(img_linear, 0, 1).astype(np.float32), "normal": normal, "roughness": np.full((H, W), 0.5, np.float32),
            "metallic": np.zeros((H, W), np.float32), "depth": np.full((H, W), 2.0, np.float32)}


def load_predictions(pred_dir: str, size) -> Dict[str, np.ndarray]:
    g = lambda n: os.path.join(pred_dir, n)
    from PIL import Image

    # roughnessPred.png / metallicPred.png are written with mi.util.write_bitmap (inverse_img_w_mi.py:675-676), which sRGB-encodes
    # 8-bit output: decode, so that a --pred_dir produced by the reference (or by this pipeline) reads back as linear values
    gray = lambda p: _srgb_decode(np.asarray(Image.open(p).convert("L"), dtype=np.float32) / 255.0)
    out = {"albedo": read_exr(g("albedoPred.exr")), "normal": read_exr(g("normalPred.exr")), "roughness": gray(g("roughnessPred.png")),
           "metallic": gray(g("metallicPred.png")), "depth": read_exr(g("depthPred.exr"))[..., 0]}
    for k, v in out.items():
        if v.shape[:2] != tuple(size):
            vv = v if v.ndim == 3 else v[..., None]
            vv = center_crop_and_resize(vv.astype(np.float32), size)
            out[k] = vv if v.ndim == 3 else vv[..., 0]
    return out


class FrameWriter:
    """env.png / env_frames / mat_frames / opt_env_img.png (inverse_img_w_mi.py:257-284,438-446,559-566).

    The reference encodes a PNG on the optimisation thread every 10 epochs; at this build's iteration rates that would cost more
    than the optimisation itself (a 1536x1024 mat_frame takes ~150 ms to deflate, 30 iterations of pos_mlp or 900 of the fused
    loop).  Here the panel is composed and quantised to 8 bits on the GPU, copied to pinned host memory without blocking, and
    encoded by a small thread pool; `close()` (called by `inverse_image`) drains it.  File names and contents are unchanged; the
    cadence is "every poll, but at most one frame per `min_interval` seconds" instead of "every 10 epochs" (at 6 k it/s the latter
    is 600 PNGs per second); the last frame of a phase (`final=True`) is always written."""

    def __init__(self, output_dir: str, workers: int = 4, max_pending: int = 64, min_interval: float = 0.2):
        from concurrent.futures import ThreadPoolExecutor
        import threading

        self.dir = output_dir
        self.env_dir = os.path.join(output_dir, "env_frames")
        self.mat_dir = os.path.join(output_dir, "mat_frames")
        os.makedirs(self.env_dir, exist_ok=True)
        os.makedirs(self.mat_dir, exist_ok=True)
        self.env_frames: List[str] = []
        self.mat_frames: List[str] = []
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="frame-png")
        self._pending: List = []
        self._max_pending = max_pending
        self._lock = threading.Lock()
        self._seq = 0
        self._latest: Dict[str, int] = {}
        self._min_interval = float(min_interval)
        self._last_frame = {"env": -1e30, "mat": -1e30}

    def due(self, kind: str, force: bool = False) -> bool:
        """Rate limiter; callers ask before they compose a frame (`kind` = "env" | "mat")."""
        now = time.perf_counter()
        if not force and now - self._last_frame[kind] < self._min_interval:
            return False
        self._last_frame[kind] = now
        return True

    @staticmethod
    def _to_host_u8(img: torch.Tensor):
        """[0,1] float panel on the device -> (uint8 host tensor, event that marks the copy done)."""
        q = (img.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
        if not q.is_cuda:
            return q, None
        host = torch.empty(q.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(q, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    def _submit(self, host: torch.Tensor, ev, paths: Sequence[str], overwrite_paths: Sequence[str] = ()) -> None:
        from PIL import Image

        self._seq += 1
        seq = self._seq

        def job():
            if ev is not None:
                ev.synchronize()
            arr = host.numpy()
            if arr.ndim == 3 and arr.shape[2] == 1:
                arr = arr[..., 0]
            im = Image.fromarray(arr)
            for path in paths:
                im.save(path)
            for path in overwrite_paths:                      # files rewritten by every frame: the newest frame wins
                with self._lock:
                    if self._latest.get(path, 0) > seq:
                        continue
                    self._latest[path] = seq
                    im.save(path)

        self._pending.append(self._pool.submit(job))
        while len(self._pending) > self._max_pending:
            self._pending.pop(0).result()

    def close(self) -> None:
        for f in self._pending:
            f.result()
        self._pending.clear()
        self._pool.shutdown(wait=True)

    @staticmethod
    def _env_panel(gt_srgb: torch.Tensor, pred_srgb: torch.Tensor, envmap: torch.Tensor) -> torch.Tensor:
        H, W, _ = gt_srgb.shape
        canvas = torch.zeros_like(gt_srgb)
        h, w = envmap.shape[:2]
        dh = min(h * 3, H // 2)
        dw = min(int(dh * (w / h)), W)
        e = F.interpolate(envmap.permute(2, 0, 1).unsqueeze(0), size=(dh, dw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
        sh, sw = (H - dh) // 2, (W - dw) // 2
        canvas[sh:sh + dh, sw:sw + dw] = e
        return torch.cat([gt_srgb, pred_srgb, canvas], dim=1)

    def env_frame(self, loop_num: int, epoch: int, gt: torch.Tensor, pred: torch.Tensor, envmap: torch.Tensor, final: bool = False) -> None:
        g = _loss.linear_to_srgb(gt.clamp_min(0))
        p = _loss.linear_to_srgb(pred.clamp_min(0))
        panel, ev = self._to_host_u8(self._env_panel(g, p, envmap))
        env8, ev2 = self._to_host_u8(envmap)
        path = os.path.join(self.env_dir, f"opt_env_frame_{loop_num}_{epoch:04d}.png")
        self._submit(env8, ev2, (), (os.path.join(self.dir, "env.png"),))
        self._submit(panel, ev, (path,), (os.path.join(self.dir, "opt_env_img.png"),) if final else ())
        self.env_frames.append(path)

    def mat_frame(self, loop_num: int, part: str, epoch: int, gt: torch.Tensor, pred_srgb: torch.Tensor, maps: Dict[str, torch.Tensor],
                  normal: torch.Tensor) -> None:
        tiles = [_loss.linear_to_srgb(gt.clamp_min(0)), pred_srgb, maps["albedo"], maps["roughness"].expand(-1, -1, 3),
                 maps["metallic"].expand(-1, -1, 3), normal]
        rows = [torch.cat(tiles[:3], dim=1), torch.cat(tiles[3:], dim=1)]           # make_grid(nrow=3) without padding
        path = os.path.join(self.mat_dir, f"mat_frame_{loop_num}_{part}_{epoch:04d}.png")
        host, ev = self._to_host_u8(torch.cat(rows, dim=0))
        self._submit(host, ev, (path,))
        self.mat_frames.append(path)


def create_video_from_frames(frame_paths: Sequence[str], out_path: str, fps: int = 10) -> Optional[str]:
    """create_video_from_frames (inverse_img_w_mi.py:602-612): every collected frame, in order, at `fps`, as an mp4.  No ffmpeg in
    this image: the frames are JPEG-coded and muxed by `video_mp4.write_mp4` (Motion-JPEG in MP4)."""
    from PIL import Image

    from .video_mp4 import write_mp4

    paths = [p for p in frame_paths if os.path.exists(p)]
    if not paths:
        return None
    ims = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
    h, w = ims[0].shape[:2]
    ims = [im if im.shape[:2] == (h, w) else np.asarray(Image.fromarray(im).resize((w, h), Image.BILINEAR)) for im in ims]
    return write_mp4(out_path, ims, fps=fps)


def create_animation_from_frames(frame_paths: Sequence[str], out_path: str, fps: int = 10, max_frames: int = 40, max_side: int = 480) -> Optional[str]:
    """Stand-in for create_video_from_frames (inverse_img_w_mi.py:593-599; mp4 through OpenCV there): an animated GIF of at most
    `max_frames` evenly spaced frames, the long side reduced to `max_side` pixels (GIF encoding is slow: 80 frames at 768 pixels cost
    7 s, a third of a whole pos_mlp inversion; these defaults cost about 1.5 s)."""
    from PIL import Image

    paths = [p for p in frame_paths if os.path.exists(p)]
    if not paths:
        return None
    if len(paths) > max_frames:
        idx = np.linspace(0, len(paths) - 1, max_frames).round().astype(int)
        paths = [paths[i] for i in idx]
    ims = []
    for p in paths:
        im = Image.open(p).convert("RGB")
        k = max_side / max(im.size)
        if k < 1:
            im = im.resize((max(1, int(im.size[0] * k)), max(1, int(im.size[1] * k))), Image.BILINEAR)
        ims.append(im)
    ims[0].save(out_path, save_all=True, append_images=ims[1:], duration=int(1000 / fps), loop=0)
    return out_path


def save_results(path: str, best: Dict[str, torch.Tensor], normal: torch.Tensor) -> None:
    """SaveBest.save_results (myutils/misc.py:99-111): best_results/{envmap.hdr, albedo/roughness/metallic/rendered_img/normal.exr}."""
    os.makedirs(path, exist_ok=True)
    np_ = lambda t: t.detach().cpu().numpy()
    if "envmap" in best:
        write_hdr(os.path.join(path, "envmap.hdr"), np_(best["envmap"]))
    write_exr(os.path.join(path, "albedo.exr"), np_(best["albedo"]))
    write_exr(os.path.join(path, "roughness.exr"), np_(best["roughness"]))
    write_exr(os.path.join(path, "metallic.exr"), np_(best["metallic"]))
    if "rendered_img" in best:
        write_exr(os.path.join(path, "rendered_img.exr"), np_(best["rendered_img"]))
    write_exr(os.path.join(path, "normal.exr"), np_(normal))


def initial_prediction(img: np.ndarray, size: int, device, matnet=None, matnet_weights: Optional[str] = None, pred_dir: Optional[str] = None):
    """MaterialNet's initial guess of the maps (inverse_img_w_mi.py:648-661): from a loaded network, a weights file, a directory of
    predictions in the reference's file layout, or -- with a warning -- the flat prior."""
    if matnet is None and matnet_weights:
        from .materialnet import MaterialNet

        matnet = MaterialNet()
        matnet.load_state_dict(torch.load(matnet_weights, map_location="cpu", weights_only=True))
    if matnet is not None:
        return matnet.to(device).eval().infer_image(img)
    if pred_dir:
        return load_predictions(pred_dir, (size, size))
    warnings.warn("neither --matnet_weights nor --pred_dir given: starting from a FLAT prior (albedo = image, roughness 0.5, "
                  "metallic 0, planar depth) instead of MaterialNet's prediction (inverse_img_w_mi.py:648-661)", UserWarning)
    return flat_prior(img)


def inverse_images_batched(img_paths: Sequence[str], save_names: Sequence[str], opt_src: str = "arm", opt_order: Sequence[str] = ("arm",),
                           opt_env_from: int = 0, save_path: Optional[str] = None, size: int = 512, spp: int = 64, num_epochs: int = 5000,
                           pred_dirs: Optional[Sequence[Optional[str]]] = None, device: str = "cuda", matnet=None, log=print,
                           use_mask: bool = False) -> Dict[str, object]:
    """`--model_name none` on several photographs at once: the images of a rank's shard as ONE batch in the kernels' batch dimension
    (per-image light, SaveBest and EarlyStopping on the device), each with the reference's output directory (inverse_img_w_mi.py:623-770
    per image; the reference runs them one after another, run_inverse_pipeline.sh:16-28).  `mesh_mask.png` and pixels the mesher
    leaves without a triangle are per-image background masks of the batch; `--use_mask` reads every image's `best_results/mask.png`
    (:702-711; an image without one runs unmasked) and its masked means are per image (loop.MaskedBatchPhase); parts with 'n' stay with
    `inverse_image`."""
    from . import optimize, render
    from . import mesh as _mesh

    if "n" in str(list(opt_order)):
        raise ValueError("inverse_images_batched optimises a / r / m under the geometric normal")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
    mats, depths, normals, out_dirs, holes, masks = [], [], [], [], [], []
    for k, (path, name) in enumerate(zip(img_paths, save_names)):
        output_dir = get_output_dir(name, save_path)
        os.makedirs(os.path.join(output_dir, "best_results"), exist_ok=True)
        if use_mask:                                         # :702-711
            from PIL import Image

            mp = os.path.join(output_dir, "best_results", "mask.png")
            if os.path.exists(mp):
                mk = np.asarray(Image.open(mp))
                mk = np.ascontiguousarray((mk[..., 0] if mk.ndim == 3 else mk) > 0)
                if mk.shape != (size, size):
                    raise ValueError(f"{mp}: mask is {mk.shape}, the run is {size}x{size}")
                log(f"Applied mask from {mp}")
            else:
                warnings.warn(f"No mask found for {name}, continuing without mask", UserWarning)
                mk = np.zeros((size, size), dtype=bool)
            masks.append(torch.from_numpy(mk))
        mm_path, mesh_mask = os.path.join(output_dir, "mesh_mask.png"), np.zeros((size, size), dtype=bool)
        if os.path.exists(mm_path):                          # :713-724: pixels without geometry
            from PIL import Image

            mk = np.asarray(Image.open(mm_path))
            mesh_mask = np.ascontiguousarray((mk[..., 0] if mk.ndim == 3 else mk) > 0)
            if mesh_mask.shape != (size, size):
                raise ValueError(f"{mm_path}: mask is {mesh_mask.shape}, the run is {size}x{size}")
        img = center_crop_and_resize(load_image(path), (size, size))
        if not path.endswith(".exr"):
            img = np.asarray(_loss.srgb_to_linear(torch.from_numpy(img)).numpy(), dtype=np.float32)
        pred = initial_prediction(img, size, device, matnet=matnet, pred_dir=pred_dirs[k] if pred_dirs else None)
        mats.append({"gt_image": t(img), "albedo": t(pred["albedo"]).clamp(0, 1), "roughness": t(pred["roughness"]).unsqueeze(-1).clamp(0.07, 1),
                     "metallic": t(pred["metallic"]).unsqueeze(-1).clamp(0, 1)})
        write_exr(os.path.join(output_dir, "albedoPred.exr"), pred["albedo"])
        write_exr(os.path.join(output_dir, "normalPred.exr"), pred["normal"])
        write_png(os.path.join(output_dir, "roughnessPred.png"), pred["roughness"], linear=True)
        write_png(os.path.join(output_dir, "metallicPred.png"), pred["metallic"], linear=True)
        write_exr(os.path.join(output_dir, "depthPred.exr"), pred["depth"])
        write_exr(os.path.join(output_dir, "gt_image.exr"), img)
        write_png(os.path.join(output_dir, "gt_image.png"), img, linear=True)
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            json.dump({"img_path": path, "save_name": name, "opt_src": opt_src, "opt_order": list(opt_order), "use_mask": bool(use_mask),
                       "opt_env_from": opt_env_from, "model_name": "none", "timestamp": time.strftime("%Y-%m-%d %H:%M:%S"),
                       "image_size": list(img.shape[:2]), "spp": spp, "output_type": "arm", "use_mesh_normal": True}, f, indent=4)
        depth = np.array(2 * pred["depth"].max() - pred["depth"], dtype=np.float32)
        depth[mesh_mask] = 0.0                                                                   # :723
        rm = _mesh.reference_mesh(depth, render.DEFAULT_FOV)
        mesh_path = os.path.join(output_dir, f"{name}.ply")
        if not os.path.exists(mesh_path):
            _mesh.write_ply(mesh_path, rm["vertices"], rm["triangles"])
        holes.append(torch.from_numpy(~rm["has_faces"]))
        nrm_k = t(rm["normals"])
        nrm_k[holes[-1].to(device)] = torch.tensor([0.0, 0.0, 1.0], device=device)                # never shaded: these pixels show the environment
        normals.append(nrm_k)
        depths.append(t(depth))
        out_dirs.append(output_dir)
    B = len(mats)
    mat = {k: torch.stack([m[k] for m in mats]) for k in mats[0]}
    scene = render.load_estimated_mesh(torch.stack(depths), use_mesh_normal=True, device=device)
    scene.geo_normal = torch.stack(normals).contiguous()                   # the reference mesher's per-pixel normals (gap closing included)
    scene.set_mesh_mask(torch.stack(holes))                                # vertices without a triangle: no geometry along that camera ray
    use_mask = bool(use_mask) and any(bool(m.any()) for m in masks)
    if use_mask:
        mat["mask"] = torch.stack(masks).to(device)
    res = optimize.optimize_envmap_ARMN(scene, mat, optimize_order=list(opt_order), spp=spp, opt_env_from=opt_env_from, opt_src=opt_src,
                                        num_epochs=num_epochs, log=log, model_name="none", use_mask=use_mask)
    nrm = scene.geo_normal
    for b, output_dir in enumerate(out_dirs):
        best = {k: res[k][b] for k in ("albedo", "roughness", "metallic", "rendered_img")}
        env = res["envmap"][b] if res["envmap"].ndim == 4 else res["envmap"]
        best["envmap"] = env
        save_results(os.path.join(output_dir, "best_results"), best, nrm[b])
        write_hdr(os.path.join(output_dir, "final_envmap.hdr"), env.detach().cpu().numpy())
    res["output_dirs"] = out_dirs
    return res


def inverse_image(img_inverse_path: str, save_name: str, opt_src: str = "arm", opt_order: Sequence[str] = ("arm",), use_mask: bool = False,
                  opt_env_from: int = 0, save_path: Optional[str] = None, model_name: str = "pos_mlp", size: int = 512, spp: int = 64,
                  num_epochs: int = 5000, pred_dir: Optional[str] = None, device: str = "cuda", sync_every: int = 10,
                  log=print, matnet_weights: Optional[str] = None, frame_interval: float = 0.2, matnet=None, geometry: str = "mesh",
                  digests: Optional[list] = None) -> Dict[str, object]:
    """inverse_img_w_mi.py:623-770 (resolution-generic: `size`; `model_name` is honoured, F4).  `matnet`: an already loaded MaterialNet
    (run_batch.py loads the weights once on rank 0 and broadcasts them)."""
    from . import optimize, render

    if model_name not in ("none", "pos_mlp"):
        raise ValueError("model_name should be 'none' or 'pos_mlp'")
    log(f"Inverse image {img_inverse_path}")
    output_dir = get_output_dir(save_name, save_path)
    os.makedirs(os.path.join(output_dir, "best_results"), exist_ok=True)

    img = center_crop_and_resize(load_image(img_inverse_path), (size, size))                      # :641-642
    if not img_inverse_path.endswith(".exr"):
        warnings.warn("The input image is in PNG/JPG format, assume it is sRGB, will convert to linear", UserWarning)
        img = np.asarray(_loss.srgb_to_linear(torch.from_numpy(img)).numpy(), dtype=np.float32)  # :643-645

    mat: Dict[str, torch.Tensor] = {}
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
    if opt_src != "skip" or list(opt_order) != ["skip"]:
        pred = initial_prediction(img, size, device, matnet=matnet, matnet_weights=matnet_weights, pred_dir=pred_dir)
        mat["gt_image"] = t(img)                                                                 # :663-670
        mat["albedo"] = t(pred["albedo"]).clamp(0, 1)
        mat["normal"] = t(pred["normal"])
        mat["roughness"] = t(pred["roughness"]).unsqueeze(-1).clamp(0.07, 1)
        mat["metallic"] = t(pred["metallic"]).unsqueeze(-1).clamp(0, 1)
        depth = pred["depth"]
        write_exr(os.path.join(output_dir, "albedoPred.exr"), pred["albedo"])                    # :672-678
        write_exr(os.path.join(output_dir, "normalPred.exr"), pred["normal"])
        write_png(os.path.join(output_dir, "roughnessPred.png"), pred["roughness"], linear=True)   # sRGB-encoded, as write_bitmap does
        write_png(os.path.join(output_dir, "metallicPred.png"), pred["metallic"], linear=True)
        write_exr(os.path.join(output_dir, "depthPred.exr"), depth)
        write_exr(os.path.join(output_dir, "gt_image.exr"), img)
        write_png(os.path.join(output_dir, "gt_image.png"), img, linear=True)
        config = {"img_path": img_inverse_path, "save_name": save_name, "opt_src": opt_src, "opt_order": list(opt_order),
                  "use_mask": use_mask, "opt_env_from": opt_env_from, "model_name": model_name,
                  "timestamp": time.strftime("%Y-%m-%d %H:%M:%S"), "image_size": list(img.shape[:2]), "spp": spp,
                  "output_type": "armn" if "n" in str(list(opt_order)) else "arm", "use_mesh_normal": "n" not in str(list(opt_order))}
        with open(os.path.join(output_dir, "config.json"), "w") as f:                            # :679-696
            json.dump(config, f, indent=4)
        depth = 2 * depth.max() - depth                                                          # :722 (MaterialNet predicts inverse depth)
        if use_mask:                                                                             # :702-711
            p = os.path.join(output_dir, "best_results", "mask.png")
            if os.path.exists(p):
                from PIL import Image

                mk = np.asarray(Image.open(p))
                mat["mask"] = torch.from_numpy(np.ascontiguousarray((mk[..., 0] if mk.ndim == 3 else mk) > 0)).to(device)
                if tuple(mat["mask"].shape) != (size, size):
                    raise ValueError(f"{p}: mask is {tuple(mat['mask'].shape)}, the run is {size}x{size}")
            else:
                warnings.warn("No mask found, continuing without mask", UserWarning)
                use_mask = False
        if opt_env_from > 1:                                                                     # :729-735
            p = os.path.join(output_dir, "best_results", "envmap.hdr")
            if os.path.exists(p):
                mat["gt_envmap"] = t(read_hdr(p))
    else:                                                                                        # :737-749 resume from best_results
        br = os.path.join(output_dir, "best_results")
        mat["albedo"] = t(read_exr(os.path.join(br, "albedo.exr"))).clamp(0, 1)
        mat["roughness"] = t(read_exr(os.path.join(br, "roughness.exr"))[..., :1]).clamp(0.07, 1)
        mat["metallic"] = t(read_exr(os.path.join(br, "metallic.exr"))[..., :1]).clamp(0, 1)
        mat["normal"] = t(read_exr(os.path.join(br, "normal.exr")))
        mat["gt_image"] = t(img)
        depth = read_exr(os.path.join(output_dir, "depthPred.exr"))[..., 0]
        depth = 2 * depth.max() - depth

    use_mesh_normal = "n" not in str(list(opt_order))                                            # :751-758
    mesh_mask = None
    mm_path = os.path.join(output_dir, "mesh_mask.png")                                          # :713-724: pixels without geometry
    if os.path.exists(mm_path):
        from PIL import Image

        mk = np.asarray(Image.open(mm_path))
        mesh_mask = torch.from_numpy(np.ascontiguousarray((mk[..., 0] if mk.ndim == 3 else mk) > 0))
        if tuple(mesh_mask.shape) != (size, size):
            raise ValueError(f"{mm_path}: mask is {tuple(mesh_mask.shape)}, the run is {size}x{size}")
        log(f"Applied mask from {mm_path}: {int(mesh_mask.sum())} pixels see the environment directly")
    mesh_path = os.path.join(output_dir, f"{save_name}.ply")                                     # :712,721-727
    if not os.path.exists(mesh_path):
        from . import mesh as _mesh

        d_mesh = np.array(depth, dtype=np.float32)
        if mesh_mask is not None:
            d_mesh[mesh_mask.numpy()] = 0.0                                                      # :723
        rm = _mesh.reference_mesh(d_mesh, render.DEFAULT_FOV)                                    # :726-727, minAngle 6, gap closing included
        _mesh.write_ply(mesh_path, rm["vertices"], rm["triangles"])
    scene = render.load_estimated_mesh(t(depth), use_mesh_normal=use_mesh_normal, device=device, mesh_mask=mesh_mask, geometry=geometry)
    if digests is not None:
        digests.append(("prep", _loss.tensors_digest(t(img), t(depth), scene.geo_normal)))       # the image as read, the depth as flipped, the mesher's normals
    frames = FrameWriter(output_dir, min_interval=frame_interval)
    res = optimize.optimize_envmap_ARMN(scene, mat, optimize_order=list(opt_order), spp=spp, opt_env_from=opt_env_from, opt_src=opt_src,
                                        num_epochs=num_epochs, sync_every=sync_every, log=log, frames=frames,
                                        results_dir=os.path.join(output_dir, "best_results"),
                                        shading_normal=scene.geo_normal if use_mesh_normal else None,
                                        model_name=model_name, use_mask=use_mask and "mask" in mat, digests=digests)
    frames.close()
    create_video_from_frames(frames.env_frames, os.path.join(output_dir, "env_optimization.mp4"))        # :593-599
    create_video_from_frames(frames.mat_frames, os.path.join(output_dir, "mat_optimization.mp4"))
    write_hdr(os.path.join(output_dir, "final_envmap.hdr"), res["envmap"].detach().cpu().numpy())   # :297
    res["output_dir"] = output_dir
    return res
