"""Minimal OpenEXR (version 2, single-part scanline) writer / reader for the pipeline's float maps.

The reference writes its predictions and results through `mi.util.write_bitmap(..., '.exr')`
(inverse_img_w_mi.py:672-678; myutils/misc.py:99-111): float32 scanline images with channels `B,G,R` or `Y`
(SURVEY.md App. D).  This codec writes exactly that layout, uncompressed or ZIP, and reads back NO / ZIPS / ZIP
compressed float32/float16 files (enough to reload what it wrote: `--opt_src skip`, :737-749), and decodes PIZ -- the
compression of the sample outputs the reference ships (`output_imgs/*/`, written through OpenEXR's default) -- so a run of the
reference can be resumed or re-rendered here.  The PIZ decoder restates the published OpenEXR algorithm (16-bit Haar-like
wavelet on the two half-words of each float, value-range LUT, canonical Huffman code with a run-length symbol); [ext]: OpenEXR
is a third-party format, not part of the reference tree.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_MAGIC = 20000630
_PIXEL_TYPES = {0: np.uint32, 1: np.float16, 2: np.float32}
_COMPRESSION = {"none": 0, "zips": 2, "zip": 3}
_LINES_PER_BLOCK = {0: 1, 2: 1, 3: 16, 4: 32}
_PIZ = 4


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


def _huf_decode(buf: bytes, n_raw: int) -> np.ndarray:
    """OpenEXR Huffman block -> n_raw uint16 symbols.  Layout: u32 first symbol, u32 last symbol (= the run-length symbol),
    u32 table bytes, u32 payload bits, u32 reserved; code lengths packed 6 bits each (59..62: 2..5 zeros, 63: 6 + next 8 bits
    zeros); canonical codes (longer codes numerically smaller, ascending symbol order within a length); payload MSB first."""
    im, i_max, _table_len, n_bits = struct.unpack_from("<IIII", buf, 0)
    lengths = np.zeros(65537, dtype=np.int64)
    pos, c, lc, i = 20, 0, 0, im
    while i <= i_max:
        while lc < 14:
            c = (c << 8) | (buf[pos] if pos < len(buf) else 0)
            pos += 1
            lc += 8
        lc -= 6
        l = (c >> lc) & 63
        if l == 63:
            lc -= 8
            i += ((c >> lc) & 255) + 6
        elif l >= 59:
            i += l - 59 + 2
        else:
            lengths[i] = l
            i += 1
        c &= (1 << lc) - 1
    pos -= lc // 8                                     # whole bytes that were fetched ahead of the table's end
    count = np.bincount(lengths, minlength=59)
    start = np.zeros(59, dtype=np.int64)
    code = 0
    for l in range(58, 0, -1):
        start[l] = code
        code = (code + count[l]) >> 1
    syms = np.nonzero(lengths)[0]
    order = np.argsort(lengths[syms], kind="stable")
    syms = syms[order]
    lens = lengths[syms]
    first = np.searchsorted(lens, lens, side="left")   # rank within the same length (ascending symbol)
    codes = start[lens] + (np.arange(syms.size) - first)
    short_len, short_sym, long_codes = [0] * (1 << 14), [0] * (1 << 14), {}
    for s, l, cd in zip(syms.tolist(), lens.tolist(), codes.tolist()):
        if l <= 14:
            lo = cd << (14 - l)
            n = 1 << (14 - l)
            short_len[lo:lo + n] = [l] * n
            short_sym[lo:lo + n] = [s] * n
        else:
            long_codes[(l, cd)] = s
    payload = bytes(buf[pos:pos + (n_bits + 7) // 8]) + b"\0" * 32
    out = np.empty(n_raw, dtype=np.uint16)
    o, p, c, lc, rlc = 0, 0, 0, 0, i_max
    while o < n_raw:
        if lc < 72:
            c = ((c & ((1 << lc) - 1)) << 64) | int.from_bytes(payload[p:p + 8], "big")
            p += 8
            lc += 64
        idx = (c >> (lc - 14)) & 0x3FFF
        l = short_len[idx]
        if l:
            sym = short_sym[idx]
        else:
            for l in range(15, 59):
                sym = long_codes.get((l, (c >> (lc - l)) & ((1 << l) - 1)))
                if sym is not None:
                    break
            else:
                raise ValueError("corrupt Huffman stream in PIZ block")
        lc -= l
        if sym == rlc:
            lc -= 8
            n = (c >> lc) & 255
            if o == 0 or o + n > n_raw:
                raise ValueError("corrupt run length in PIZ block")
            out[o:o + n] = out[o - 1]
            o += n
        else:
            out[o] = sym
            o += 1
    return out


def _wdec14(l: np.ndarray, h: np.ndarray):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    return (ai & 0xFFFF).astype(np.uint16), ((ai - hs) & 0xFFFF).astype(np.uint16)


def _wdec16(l: np.ndarray, h: np.ndarray):
    m, d = l.astype(np.int32), h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xFFFF
    aa = (d + bb - 0x8000) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a: np.ndarray, max_value: int) -> None:
    """In-place inverse of OpenEXR's 2-D wavelet on a [ny, nx] uint16 view, coarse to fine."""
    dec = _wdec14 if max_value < (1 << 14) else _wdec16
    ny, nx = a.shape
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        cy = (ny - p2) // p2 + 1 if ny >= p2 else 0
        cx = (nx - p2) // p2 + 1 if nx >= p2 else 0
        ys, ys1 = slice(0, cy * p2, p2), slice(p, p + cy * p2, p2)
        xs, xs1 = slice(0, cx * p2, p2), slice(p, p + cx * p2, p2)
        if cy and cx:
            i00, i10 = dec(a[ys, xs], a[ys1, xs])
            i01, i11 = dec(a[ys, xs1], a[ys1, xs1])
            a[ys, xs], a[ys, xs1] = dec(i00, i01)
            a[ys1, xs], a[ys1, xs1] = dec(i10, i11)
        if (nx & p) and cy:                            # one unpaired column at this level
            x = cx * p2
            a[ys, x], a[ys1, x] = dec(a[ys, x], a[ys1, x])
        if (ny & p) and cx:                            # one unpaired row
            y = cy * p2
            a[y, xs], a[y, xs1] = dec(a[y, xs], a[y, xs1])
        p2 = p
        p >>= 1


def _piz_decode(blob: bytes, n_lines: int, width: int, halfs_per_channel) -> bytes:
    """One PIZ chunk -> the scanline-interleaved raw bytes an uncompressed chunk would hold."""
    lo, hi = struct.unpack_from("<HH", blob, 0)
    bitmap = np.zeros(8192, dtype=np.uint8)
    pos = 4
    if lo <= hi:
        bitmap[lo:hi + 1] = np.frombuffer(blob, dtype=np.uint8, count=hi - lo + 1, offset=pos)
        pos += hi - lo + 1
    present = np.unpackbits(bitmap, bitorder="little").astype(bool)
    present[0] = True                                  # zero is always representable
    lut = np.zeros(65536, dtype=np.uint16)
    values = np.nonzero(present)[0]
    lut[:values.size] = values
    max_value = values.size - 1
    (length,) = struct.unpack_from("<i", blob, pos)
    pos += 4
    total = sum(n_lines * width * s for s in halfs_per_channel)
    tmp = _huf_decode(blob[pos:pos + length], total)
    start, planes = 0, []
    for s in halfs_per_channel:
        block = tmp[start:start + n_lines * width * s].reshape(n_lines, width * s)
        for j in range(s):
            _wav2_decode(block[:, j::s], max_value)
        planes.append(block)
        start += block.size
    lines = [lut[pl[y]].astype("<u2").tobytes() for y in range(n_lines) for pl in planes]
    return b"".join(lines)


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
        raise NotImplementedError(f"{path}: EXR compression {comp} is not supported (none, ZIPS, ZIP, PIZ are)")
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
        if comp == _PIZ and size < expect:
            blob = _piz_decode(blob, nl, W, [np.dtype(_PIXEL_TYPES[t]).itemsize // 2 for _, t in ch])
        elif comp and size < expect:
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
