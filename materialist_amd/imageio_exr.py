"""Minimal OpenEXR (version 2, single-part scanline) writer / reader for the pipeline's float maps.

The reference writes its predictions and results through `mi.util.write_bitmap(..., '.exr')`
(inverse_img_w_mi.py:672-678; myutils/misc.py:99-111): float32 scanline images with channels `B,G,R` or `Y`
(SURVEY.md App. D).  This codec writes exactly that layout, uncompressed or ZIP, and reads back NO / ZIPS / ZIP
compressed float32/float16 files (enough to reload what it wrote: `--opt_src skip`, :737-749).  PIZ, which the committed
sample outputs of the reference use, is not decoded.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_MAGIC = 20000630
_PIXEL_TYPES = {0: np.uint32, 1: np.float16, 2: np.float32}
_COMPRESSION = {"none": 0, "zips": 2, "zip": 3}
_LINES_PER_BLOCK = {0: 1, 2: 1, 3: 16}


def _attr(name: str, typ: str, payload: bytes) -> bytes:
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def _zip_encode(raw: bytes) -> bytes:
    """OpenEXR ZIP: de-interleave bytes (even/odd), delta-predict, deflate."""
    a = np.frombuffer(raw, dtype=np.uint8)
    n = a.size
    t = np.empty(n, dtype=np.uint8)
    half = (n + 1) // 2
    t[:half] = a[0::2]
    t[half:] = a[1::2]
    d = t.astype(np.int16)
    d[1:] = (d[1:] - t[:-1].astype(np.int16) + 128 + 256) % 256
    return zlib.compress(d.astype(np.uint8).tobytes(), 6)


def _zip_decode(comp: bytes, expected: int) -> bytes:
    t = np.frombuffer(zlib.decompress(comp), dtype=np.uint8).astype(np.int32)
    # undo the delta predictor: t[i] = t[i-1] + d[i] - 128  (mod 256)
    t = (np.cumsum(t - np.concatenate([[0], np.full(t.size - 1, 128)])) % 256).astype(np.uint8)
    n = t.size
    half = (n + 1) // 2
    out = np.empty(n, dtype=np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    assert n == expected
    return out.tobytes()


def write_exr(path: str, img: np.ndarray, compression: str = "zip") -> None:
    """img: [H,W] or [H,W,1] -> channel Y;  [H,W,3] RGB -> channels B,G,R (alphabetical, as OpenEXR stores them)."""
    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 2:
        img = img[..., None]
    H, W, C = img.shape
    if C == 1:
        names, planes = ["Y"], [img[..., 0]]
    elif C == 3:
        names, planes = ["B", "G", "R"], [img[..., 2], img[..., 1], img[..., 0]]
    else:
        raise ValueError("write_exr expects 1 or 3 channels")
    comp = _COMPRESSION[compression]
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, W - 1, H - 1)
    header = b"".join([
        _attr("channels", "chlist", chlist),
        _attr("compression", "compression", struct.pack("<B", comp)),
        _attr("dataWindow", "box2i", box),
        _attr("displayWindow", "box2i", box),
        _attr("lineOrder", "lineOrder", struct.pack("<B", 0)),
        _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)),
        _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)),
        _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)),
    ]) + b"\0"
    lpb = _LINES_PER_BLOCK[comp]
    blocks = []
    for y0 in range(0, H, lpb):
        y1 = min(y0 + lpb, H)
        raw = b"".join(planes[c][y].tobytes() for y in range(y0, y1) for c in range(len(names)))
        data = raw
        if comp:
            z = _zip_encode(raw)
            data = z if len(z) < len(raw) else raw
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    head = struct.pack("<ii", _MAGIC, 2) + header
    table_pos = len(head)
    off = table_pos + 8 * len(blocks)
    table = []
    for b in blocks:
        table.append(off)
        off += len(b)
    with open(path, "wb") as f:
        f.write(head)
        f.write(struct.pack(f"<{len(table)}Q", *table))
        for b in blocks:
            f.write(b)


def read_exr(path: str) -> np.ndarray:
    """Returns float32 [H,W,C]: RGB order when channels B,G,R are present, else the file's channels in name order."""
    with open(path, "rb") as f:
        data = f.read()
    magic, version = struct.unpack_from("<ii", data, 0)
    if magic != _MAGIC or (version & 0x200) or (version & 0x1000):
        raise ValueError(f"{path}: not a single-part scanline OpenEXR file")
    pos = 8
    attrs = {}
    while data[pos] != 0:
        end = data.index(b"\0", pos)
        name = data[pos:end].decode()
        pos = end + 1
        end = data.index(b"\0", pos)
        typ = data[pos:end].decode()
        pos = end + 1
        (size,) = struct.unpack_from("<i", data, pos)
        pos += 4
        attrs[name] = (typ, data[pos:pos + size])
        pos += size
    pos += 1
    ch, p = [], 0
    raw = attrs["channels"][1]
    while raw[p] != 0:
        end = raw.index(b"\0", p)
        nm = raw[p:end].decode()
        ptype, = struct.unpack_from("<i", raw, end + 1)
        ch.append((nm, ptype))
        p = end + 1 + 16
    comp = attrs["compression"][1][0]
    if comp not in _LINES_PER_BLOCK:
        raise NotImplementedError(f"{path}: EXR compression {comp} (e.g. PIZ = 4) is not supported")
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    lpb = _LINES_PER_BLOCK[comp]
    nblocks = (H + lpb - 1) // lpb
    offsets = struct.unpack_from(f"<{nblocks}Q", data, pos)
    out = np.zeros((H, W, len(ch)), dtype=np.float32)
    line_bytes = sum(W * np.dtype(_PIXEL_TYPES[t]).itemsize for _, t in ch)
    for off in offsets:
        y, size = struct.unpack_from("<ii", data, off)
        nl = min(lpb, y1 - y + 1)
        blob = data[off + 8: off + 8 + size]
        expect = nl * line_bytes
        if comp and size < expect:
            blob = _zip_decode(blob, expect)
        p = 0
        for ly in range(nl):
            for ci, (_, t) in enumerate(ch):
                dt = np.dtype(_PIXEL_TYPES[t])
                out[y - y0 + ly, :, ci] = np.frombuffer(blob, dtype=dt, count=W, offset=p).astype(np.float32)
                p += W * dt.itemsize
    names = [n for n, _ in ch]
    if set("RGB") <= set(names):
        out = out[..., [names.index("R"), names.index("G"), names.index("B")]]
    return out
