"""Radiance RGBE (.hdr) reader / writer.

The reference reads and writes its environment maps through `mi.Bitmap` / `mi.util.write_bitmap`
(inverse_img_w_mi.py:297,733; myutils/misc.py:101) as `#?RGBE` files with a `-Y H +X W` resolution
line.  This module is the build's own codec for that format: flat and new-style
run-length-encoded scanlines on read, flat scanlines on write.
"""
from __future__ import annotations

import numpy as np


def _decode_rgbe(rgbe: np.ndarray) -> np.ndarray:
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - (128 + 8)), 0.0).astype(np.float32)
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


def read_hdr(path: str) -> np.ndarray:
    """Return float32 [H, W, 3] linear radiance."""
    with open(path, "rb") as f:
        data = f.read()
    if not (data.startswith(b"#?RGBE") or data.startswith(b"#?RADIANCE")):
        raise ValueError(f"{path}: not a Radiance HDR file")
    pos = 0
    # header ends with an empty line, then the resolution line
    while True:
        end = data.index(b"\n", pos)
        line = data[pos:end]
        pos = end + 1
        if line.strip() == b"":
            break
    end = data.index(b"\n", pos)
    res = data[pos:end].split()
    pos = end + 1
    if len(res) != 4 or res[0] != b"-Y" or res[2] != b"+X":
        raise ValueError(f"{path}: unsupported resolution line {res!r}")
    H, W = int(res[1]), int(res[3])
    buf = np.frombuffer(data, dtype=np.uint8, offset=pos)
    out = np.empty((H, W, 4), dtype=np.uint8)
    p = 0
    for y in range(H):
        if W >= 8 and W < 32768 and buf[p] == 2 and buf[p + 1] == 2 and ((int(buf[p + 2]) << 8) | int(buf[p + 3])) == W:
            p += 4
            for ch in range(4):
                x = 0
                while x < W:
                    n = int(buf[p])
                    p += 1
                    if n > 128:
                        n -= 128
                        out[y, x:x + n, ch] = buf[p]
                        p += 1
                    else:
                        out[y, x:x + n, ch] = buf[p:p + n]
                        p += n
                    x += n
        else:
            out[y] = buf[p:p + 4 * W].reshape(W, 4)
            p += 4 * W
    return _decode_rgbe(out)


def encode_rgbe(img: np.ndarray) -> np.ndarray:
    img = np.maximum(np.asarray(img, dtype=np.float32), 0.0)
    v = img.max(axis=-1)
    m, e = np.frexp(v)
    scale = np.where(v > 1e-32, m * 256.0 / np.maximum(v, 1e-32), 0.0)
    rgbe = np.zeros(img.shape[:-1] + (4,), dtype=np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(v > 1e-32, e + 128, 0).astype(np.uint8)
    return rgbe


def write_hdr(path: str, img: np.ndarray) -> None:
    """Write float [H, W, 3] as a flat (non-RLE) `#?RGBE` file."""
    img = np.asarray(img, dtype=np.float32)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("write_hdr expects [H, W, 3]")
    H, W, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(f"-Y {H} +X {W}\n".encode())
        f.write(encode_rgbe(img).tobytes())
