"""MaterialNet: DINOv2 ViT-B/14 encoder with two DPT heads that predict depth and
albedo / roughness / metallic / normal from one image (Material_net/dpt.py:175-269, dinov2.py, dinov2_layers/, util/blocks.py).

Restated for PyTorch-ROCm, not copied: one compact module tree whose parameter names equal the reference's, so
`matnet_weights.pth` (inverse_img_w_mi.py:648-654) loads with `load_state_dict` unchanged.  Attention goes through
`F.scaled_dot_product_attention` (hipBLASLt / MFMA on ROCm) instead of the optional xformers path (attention.py:65-82).
Inference-only: no drop-path, no mask tokens, no register tokens (the reference instantiates none for 'vitb', dinov2.py:398-415).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

PATCH = 14
EMBED, DEPTH, HEADS = 768, 12, 12            # vit_base (dinov2.py:355-366)
TAPS = (2, 5, 8, 11)                         # intermediate_layer_idx['vitb'] (dpt.py:186-188)
POS_GRID = 37                                # img_size 518 / patch 14
# Encoders.  The reference wires 'vitb' only (dpt.py:186-188: the one entry of intermediate_layer_idx); 'vitl' is the
# configs[3] throughput variant (dinov2.py:367-378 vit_large: 1024 wide, 24 blocks, 16 heads) with the DPT head at its own
# defaults (dpt.py:42-44: features 256, out_channels [256, 512, 1024, 1024]) and evenly spaced taps -- random weights only, no
# checkpoint of that shape exists in the reference.
ENCODERS = {"vitb": {"embed": 768, "depth": 12, "heads": 12, "taps": (2, 5, 8, 11), "features": 128, "out_channels": (96, 192, 384, 768)},
            "vitl": {"embed": 1024, "depth": 24, "heads": 16, "taps": (4, 11, 17, 23), "features": 256, "out_channels": (256, 512, 1024, 1024)}}


class _Attention(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        return self.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Scale(nn.Module):
    def __init__(self, dim: int, init: float = 1.0):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _Block(nn.Module):
    """x + ls1(attn(norm1 x)); x + ls2(mlp(norm2 x))   (dinov2_layers/block.py:82-107, eval path)."""

    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.norm1, self.attn, self.ls1 = nn.LayerNorm(dim, eps=1e-6), _Attention(dim, heads), _Scale(dim)
        self.norm2, self.mlp, self.ls2 = nn.LayerNorm(dim, eps=1e-6), _Mlp(dim, 4 * dim), _Scale(dim)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class _PatchEmbed(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=PATCH, stride=PATCH)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class DinoV2(nn.Module):
    def __init__(self, embed: int = EMBED, depth: int = DEPTH, heads: int = HEADS, taps: Sequence[int] = TAPS):
        super().__init__()
        self.embed_dim, self.default_taps = embed, tuple(taps)
        self.patch_embed = _PatchEmbed(embed)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed))
        self.pos_embed = nn.Parameter(torch.zeros(1, POS_GRID * POS_GRID + 1, embed))
        self.mask_token = nn.Parameter(torch.zeros(1, embed))          # unused at inference; kept so the state_dict matches
        self.blocks = nn.ModuleList(_Block(embed, heads) for _ in range(depth))
        self.norm = nn.LayerNorm(embed, eps=1e-6)

    def _pos(self, n_patches: int, h: int, w: int) -> torch.Tensor:
        """Bicubic resampling of the 37x37 position grid with DINOv2's +0.1 offset (dinov2.py:179-210)."""
        if n_patches == POS_GRID * POS_GRID and h == w:
            return self.pos_embed
        pe = self.pos_embed.float()
        E = self.embed_dim
        grid = pe[:, 1:].reshape(1, POS_GRID, POS_GRID, E).permute(0, 3, 1, 2)
        h0, w0 = h // PATCH + 0.1, w // PATCH + 0.1
        grid = F.interpolate(grid, scale_factor=(h0 / POS_GRID, w0 / POS_GRID), mode="bicubic", antialias=False)
        assert grid.shape[-2:] == (int(h0), int(w0))
        return torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, E)], dim=1).to(self.pos_embed.dtype)

    def taps(self, x: torch.Tensor, which: Sequence[int] = None) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """get_intermediate_layers(x, which, return_class_token=True, norm=True) (dinov2.py:297-321)."""
        which = self.default_taps if which is None else which
        B, _, h, w = x.shape
        t = self.patch_embed(x)
        t = torch.cat([self.cls_token.expand(B, -1, -1), t], dim=1)
        t = t + self._pos(t.shape[1] - 1, h, w)
        out = []
        for i, blk in enumerate(self.blocks):
            t = blk(t)
            if i in which:
                n = self.norm(t)
                out.append((n[:, 1:], n[:, 0]))
        return out


class _RCU(nn.Module):
    def __init__(self, f: int):
        super().__init__()
        self.conv1 = nn.Conv2d(f, f, 3, padding=1)
        self.conv2 = nn.Conv2d(f, f, 3, padding=1)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(F.relu(x)))) + x


class _Fusion(nn.Module):
    """util/blocks.py:83-147: (skip + RCU1(lateral)) -> RCU2 -> bilinear up (align_corners) -> 1x1 conv."""

    def __init__(self, f: int):
        super().__init__()
        self.out_conv = nn.Conv2d(f, f, 1)
        self.resConfUnit1, self.resConfUnit2 = _RCU(f), _RCU(f)

    def forward(self, x, lateral=None, size=None):
        if lateral is not None:
            x = x + self.resConfUnit1(lateral)
        x = self.resConfUnit2(x)
        kw = {"scale_factor": 2} if size is None else {"size": tuple(size)}
        return self.out_conv(F.interpolate(x, mode="bilinear", align_corners=True, **kw))


class DPTHead(nn.Module):
    def __init__(self, in_ch: int, features: int, out_channels: Sequence[int], output_type: str):
        super().__init__()
        self.output_type = output_type
        oc = list(out_channels)
        self.projects = nn.ModuleList(nn.Conv2d(in_ch, c, 1) for c in oc)
        self.resize_layers = nn.ModuleList([nn.ConvTranspose2d(oc[0], oc[0], 4, stride=4), nn.ConvTranspose2d(oc[1], oc[1], 2, stride=2),
                                            nn.Identity(), nn.Conv2d(oc[3], oc[3], 3, stride=2, padding=1)])
        sc = nn.Module()
        for i, c in enumerate(oc):
            setattr(sc, f"layer{i + 1}_rn", nn.Conv2d(c, features, 3, padding=1, bias=False))
        for i in range(1, 5):
            setattr(sc, f"refinenet{i}", _Fusion(features))
        sc.output_conv1 = nn.Conv2d(features, features // 2, 3, padding=1)
        last = 1 if output_type == "depth" else 8
        tail = [nn.Conv2d(features // 2, 32, 3, padding=1), nn.ReLU(True), nn.Conv2d(32, last, 1)]
        if output_type == "depth":
            tail += [nn.ReLU(True), nn.Identity()]
        sc.output_conv2 = nn.Sequential(*tail)
        self.scratch = sc

    def forward(self, taps, ph: int, pw: int):
        lv = []
        for i, (tok, _) in enumerate(taps):
            x = tok.permute(0, 2, 1).reshape(tok.shape[0], tok.shape[-1], ph, pw)
            lv.append(self.resize_layers[i](self.projects[i](x)))
        s = self.scratch
        l1, l2, l3, l4 = s.layer1_rn(lv[0]), s.layer2_rn(lv[1]), s.layer3_rn(lv[2]), s.layer4_rn(lv[3])
        p = s.refinenet4(l4, size=l3.shape[2:])
        p = s.refinenet3(p, l3, size=l2.shape[2:])
        p = s.refinenet2(p, l2, size=l1.shape[2:])
        p = s.refinenet1(p, l1)
        out = F.interpolate(s.output_conv1(p), (ph * PATCH, pw * PATCH), mode="bilinear", align_corners=True)
        out = s.output_conv2(out)
        if self.output_type == "material":                      # relu(arm5), normalize(tanh(n3))  (dpt.py:161-170)
            out = torch.cat([F.relu(out[:, :5]), F.normalize(torch.tanh(out[:, 5:8]), p=2, dim=1, eps=1e-6)], dim=1)
        return out


def network_input_size(width: int, height: int, target: int = 518, multiple: int = PATCH) -> Tuple[int, int]:
    """Resize(lower_bound, keep_aspect_ratio, ensure_multiple_of=14).get_size (util/transform.py:58-100) -> (new_w, new_h)."""
    scale = max(target / width, target / height)

    def fit(x):
        y = int(np.round(x / multiple) * multiple)
        return y if y >= target else int(np.ceil(x / multiple) * multiple)

    return fit(scale * width), fit(scale * height)


DinoV2B = DinoV2   # the reference's encoder (vit_base)


class MaterialNet(nn.Module):
    def __init__(self, features: int = None, out_channels: Sequence[int] = None, encoder: str = "vitb"):
        super().__init__()
        cfg = ENCODERS[encoder]
        features = cfg["features"] if features is None else features
        out_channels = cfg["out_channels"] if out_channels is None else out_channels
        self.encoder = encoder
        self.pretrained = DinoV2(cfg["embed"], cfg["depth"], cfg["heads"], cfg["taps"])
        self.depth_head = DPTHead(cfg["embed"], features, out_channels, "depth")
        self.material_head = DPTHead(cfg["embed"], features, out_channels, "material")

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        ph, pw = x.shape[-2] // PATCH, x.shape[-1] // PATCH
        taps = self.pretrained.taps(x)
        depth = F.relu(self.depth_head(taps, ph, pw))
        armn = self.material_head(taps, ph, pw)
        return {"depth": depth, "albedo": armn[:, :3], "roughness": armn[:, 3:4], "metallic": armn[:, 4:5], "normal": armn[:, 5:8]}

    @torch.no_grad()
    def infer_image(self, raw_image: np.ndarray, input_size: int = 518) -> Dict[str, np.ndarray]:
        """dpt.py:219-269: [H,W,3] (uint8 or float in [0,1], NOT ImageNet-normalised) -> maps at the input resolution.
        The reference resizes with cv2.INTER_CUBIC; here torch bicubic (same a = -0.75 kernel, half-pixel centres)."""
        h, w = raw_image.shape[:2]
        img = raw_image.astype(np.float32) / 255.0 if raw_image.dtype == np.uint8 else raw_image.astype(np.float32)
        dev = self.pretrained.cls_token.device
        t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).unsqueeze(0).to(dev)
        nw, nh = network_input_size(w, h, input_size)
        t = F.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False)
        if t.mean() >= 10:
            t = t / 255.0
        out = self.forward(t)
        up = lambda z: F.interpolate(z, (h, w), mode="bilinear", align_corners=True)[0]
        return {"depth": up(out["depth"])[0].cpu().numpy(), "albedo": up(out["albedo"]).permute(1, 2, 0).cpu().numpy(),
                "roughness": up(out["roughness"])[0].cpu().numpy(), "metallic": up(out["metallic"])[0].cpu().numpy(),
                "normal": up(out["normal"]).permute(1, 2, 0).cpu().numpy()}


def init_from_names(model: nn.Module) -> nn.Module:
    """Deterministic weights that depend only on each tensor's NAME and shape (test fixture support: the reference module and
    this one get identical weights without shipping 433 MB)."""
    with torch.no_grad():
        for name, t in sorted(model.state_dict().items()):
            gen = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            fan_in = max(1, int(np.prod(t.shape[1:]))) if t.ndim > 1 else 1
            v = torch.randn(t.shape, generator=gen, dtype=torch.float64) / math.sqrt(fan_in)
            if t.ndim == 1:
                v = v * 0.1
                if name.endswith(("norm.weight", "norm1.weight", "norm2.weight", "gamma")):
                    v = v + 1.0
            t.copy_(v.to(t.dtype))
    return model
