"""A minimal MP4 writer for the optimisation / relighting videos (inverse_img_w_mi.py:593-612 `create_video_from_frames`,
render_final.py:405-409): the reference writes them through imageio's ffmpeg plugin; no encoder exists in this image, so the
frames are JPEG-coded by Pillow and muxed here into an ISO base media file (ISO/IEC 14496-12) as a Motion-JPEG video track:
sample entry `mp4v` whose `esds` DecoderConfigDescriptor carries objectTypeIndication 0x6C ("Visual ISO/IEC 10918-1", JPEG) --
the registered way to carry JPEG frames in MP4, which ffmpeg-based players decode as MJPEG.  One sample per frame, one chunk per
sample, constant frame duration, every sample a sync sample.

`read_mp4_frames` parses exactly what `write_mp4` produces (used by the tests; not a general demuxer).
"""
from __future__ import annotations

import io
import struct
from typing import List, Sequence

import numpy as np


def _box(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I4s", 8 + len(payload), kind) + payload


def _full(kind: bytes, version: int, flags: int, payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def _descr(tag: int, payload: bytes) -> bytes:
    n = len(payload)
    if n >= 1 << 21:
        raise ValueError("descriptor too long")
    return bytes([tag, 0x80 | (n >> 14) & 0x7F, 0x80 | (n >> 7) & 0x7F, n & 0x7F]) + payload      # 3-byte expandable size


def encode_jpeg(frame: np.ndarray, quality: int = 90) -> bytes:
    """frame [H, W, 3] uint8 (or float in [0, 1]) -> baseline JPEG bytes."""
    from PIL import Image

    f = np.asarray(frame)
    if f.dtype != np.uint8:
        f = (np.clip(f, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    if f.ndim != 3 or f.shape[2] != 3:
        raise ValueError("frames must be [H, W, 3]")
    buf = io.BytesIO()
    Image.fromarray(f, "RGB").save(buf, format="JPEG", quality=int(quality), subsampling=2, optimize=False, progressive=False)
    return buf.getvalue()


def write_mp4(path: str, frames: Sequence[np.ndarray], fps: int = 10, quality: int = 90) -> str:
    """Frames (same size, [H, W, 3]) -> `path` (Motion-JPEG in MP4).  Width and height are padded to even by edge replication."""
    if not len(frames):
        raise ValueError("no frames")
    first = np.asarray(frames[0])
    H, W = first.shape[:2]
    Hp, Wp = H + (H & 1), W + (W & 1)
    samples: List[bytes] = []
    for f in frames:
        f = np.asarray(f)
        if f.shape[:2] != (H, W):
            raise ValueError("all frames must have the same size")
        if (Hp, Wp) != (H, W):
            f = np.pad(f, ((0, Hp - H), (0, Wp - W), (0, 0)), mode="edge")
        samples.append(encode_jpeg(f, quality))
    n = len(samples)
    timescale, delta = int(fps) * 1000, 1000
    duration = n * delta
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isom" + b"iso2" + b"mp41")
    mdat_payload = b"".join(samples)
    mdat = _box(b"mdat", mdat_payload)
    first_off = len(ftyp) + 8
    offsets, o = [], first_off
    for s in samples:
        offsets.append(o)
        o += len(s)
    if o >= 1 << 32:
        raise ValueError("video larger than 4 GiB")
    unity = struct.pack(">9i", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration) + struct.pack(">iH", 0x10000, 0x0100) + b"\0" * 10 + unity +
                 b"\0" * 24 + struct.pack(">I", 2))
    tkhd = _full(b"tkhd", 0, 7, struct.pack(">IIIII", 0, 0, 1, 0, duration) + b"\0" * 8 + struct.pack(">hhhH", 0, 0, 0, 0) + unity +
                 struct.pack(">II", Wp << 16, Hp << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration) + struct.pack(">HH", 0x55C4, 0))
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide") + b"\0" * 12 + b"VideoHandler\0")
    vmhd = _full(b"vmhd", 0, 1, struct.pack(">HHHH", 0, 0, 0, 0))
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
    avg_bitrate = int(8 * len(mdat_payload) * fps / max(n, 1))
    dec_cfg = _descr(0x04, struct.pack(">BB", 0x6C, (0x04 << 2) | 1) + struct.pack(">I", max(len(s) for s in samples))[1:] +
                     struct.pack(">II", avg_bitrate, avg_bitrate))                  # objectType JPEG, streamType visual, buffer size, bitrates
    es = _descr(0x03, struct.pack(">HB", 1, 0) + dec_cfg + _descr(0x06, b"\x02"))  # ES_ID 1; SLConfig predefined = 2 (MP4)
    esds = _full(b"esds", 0, 0, es)
    mp4v = _box(b"mp4v", b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HH", Wp, Hp) + struct.pack(">II", 0x480000, 0x480000) +
                struct.pack(">I", 0) + struct.pack(">H", 1) + b"\0" * 32 + struct.pack(">Hh", 24, -1) + esds)
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + mp4v)
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, 1, 1))
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", len(s)) for s in samples))
    stco = _full(b"stco", 0, 0, struct.pack(">I", n) + b"".join(struct.pack(">I", x) for x in offsets))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
    minf = _box(b"minf", vmhd + dinf + stbl)
    mdia = _box(b"mdia", mdhd + hdlr + minf)
    moov = _box(b"moov", mvhd + _box(b"trak", tkhd + mdia))
    with open(path, "wb") as fh:
        fh.write(ftyp)
        fh.write(mdat)
        fh.write(moov)
    return path


def _children(buf: bytes, start: int, end: int):
    o = start
    while o + 8 <= end:
        size, kind = struct.unpack(">I4s", buf[o:o + 8])
        if size < 8 or o + size > end:
            raise ValueError("malformed box")
        yield kind, o + 8, o + size
        o += size


def _find(buf: bytes, start: int, end: int, *path: bytes):
    for kind, a, b in _children(buf, start, end):
        if kind == path[0]:
            return (a, b) if len(path) == 1 else _find(buf, a, b, *path[1:])
    raise KeyError(path[0])


def read_mp4_frames(path: str):
    """(frames [n, H, W, 3] uint8, fps, object_type) of a file written by `write_mp4`: walks ftyp / moov / trak / mdia / minf / stbl,
    reads stsz / stco / stts and the esds object type, decodes every sample with Pillow."""
    from PIL import Image

    buf = open(path, "rb").read()
    top = {k: (a, b) for k, a, b in _children(buf, 0, len(buf))}
    if buf[4:8] != b"ftyp" or b"moov" not in top or b"mdat" not in top:
        raise ValueError("not an MP4 written by write_mp4")
    a, b = _find(buf, *top[b"moov"], b"trak", b"mdia")
    m0, _ = _find(buf, a, b, b"mdhd")
    timescale, _dur = struct.unpack(">II", buf[m0 + 12:m0 + 20])
    s0, s1 = _find(buf, a, b, b"minf", b"stbl")
    z0, _ = _find(buf, s0, s1, b"stsz")
    _, n = struct.unpack(">II", buf[z0 + 4:z0 + 12])
    sizes = struct.unpack(">%dI" % n, buf[z0 + 12:z0 + 12 + 4 * n])
    c0, _ = _find(buf, s0, s1, b"stco")
    offs = struct.unpack(">%dI" % n, buf[c0 + 8:c0 + 8 + 4 * n])
    t0, _ = _find(buf, s0, s1, b"stts")
    _cnt, _n, delta = struct.unpack(">III", buf[t0 + 4:t0 + 16])
    d0, d1 = _find(buf, s0, s1, b"stsd")
    entry = buf[d0 + 8:d1]
    if entry[4:8] != b"mp4v":
        raise ValueError("unexpected sample entry")
    k = entry.index(b"esds")
    cfg = entry.index(b"\x04", k + 12)                      # DecoderConfigDescriptor tag after the ES descriptor header
    object_type = entry[cfg + 4]
    frames = [np.asarray(Image.open(io.BytesIO(buf[o:o + s])).convert("RGB")) for o, s in zip(offs, sizes)]
    return np.stack(frames), timescale / delta, object_type
