"""The MP4 writer of the optimisation / relighting videos (materialist_amd/video_mp4.py): structure and round trip."""
import struct

import numpy as np


def test_mp4_round_trip(tmp_path):
    from materialist_amd.video_mp4 import read_mp4_frames, write_mp4

    rng = np.random.default_rng(0)
    H, W, n = 37, 52, 7                                  # odd height: padded to even by edge replication
    yy, xx = np.mgrid[0:H, 0:W]
    frames = []
    for k in range(n):                                    # smooth content (JPEG is lossy): moving gradients
        f = np.stack([(xx + 5 * k) % 256, (yy * 3 + 2 * k) % 256, np.full_like(xx, 40 + 20 * k)], -1).astype(np.uint8)
        frames.append(f)
    path = write_mp4(str(tmp_path / "v.mp4"), frames, fps=10, quality=95)
    got, fps, object_type = read_mp4_frames(path)
    assert fps == 10.0 and object_type == 0x6C           # "Visual ISO/IEC 10918-1 (JPEG)" in the esds DecoderConfigDescriptor
    assert got.shape == (n, H + 1, W, 3)
    for k in range(n):
        err = np.abs(got[k, :H].astype(np.int32) - frames[k].astype(np.int32))
        assert err.mean() < 6.0, k                        # chroma-subsampled JPEG of sharp wrap-around edges
    assert (got[:, H] .astype(np.int32) - got[:, H - 1].astype(np.int32)).__abs__().mean() < 12.0   # the replicated row
    buf = open(path, "rb").read()
    size, kind = struct.unpack(">I4s", buf[:8])
    assert kind == b"ftyp" and buf[8:12] == b"isom"
    # top-level boxes tile the file exactly: ftyp, mdat, moov
    o, kinds = 0, []
    while o < len(buf):
        size, kind = struct.unpack(">I4s", buf[o:o + 8])
        kinds.append(kind)
        o += size
    assert o == len(buf) and kinds == [b"ftyp", b"mdat", b"moov"]
    # every sample is a complete JPEG (SOI ... EOI) at the offset stco says
    assert buf.count(b"\xff\xd8\xff") >= n


def test_float_frames_and_errors(tmp_path):
    import pytest

    from materialist_amd.video_mp4 import read_mp4_frames, write_mp4

    f = np.linspace(0, 1, 16 * 16 * 3, dtype=np.float32).reshape(16, 16, 3)
    path = write_mp4(str(tmp_path / "f.mp4"), [f, f[::-1]], fps=5)
    got, fps, _ = read_mp4_frames(path)
    assert fps == 5.0 and got.shape == (2, 16, 16, 3)
    assert np.abs(got[0].astype(np.float32) / 255 - f).mean() < 0.05
    with pytest.raises(ValueError):
        write_mp4(str(tmp_path / "e.mp4"), [])
    with pytest.raises(ValueError):
        write_mp4(str(tmp_path / "e.mp4"), [f, f[:8]])
