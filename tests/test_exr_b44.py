"""B44 / B44A (OpenEXR compression 6 / 7) in path_tracer_amd/csrc/exr_codecs.cpp: lossy, HALF channels only, 4x4 pixel blocks
of 14 bytes (B44A: 3 bytes for a block of equal values).  The `exr` crate behind ExrImage::load_exr_image (exr.rs:27) reads
them.  No file from another writer exists here: PARITY UNPINNED — the checks are a second, independent restatement of the
block layout in Python (below), hand-derived byte vectors, the cases the format stores exactly, and error bounds."""
import struct

import numpy as np
import pytest

import path_tracer_amd as pta


def _attr(name, typ, data):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data


# ---- the block layout once more, in Python (unsigned 16-bit wrap-around like the C++)
def _to_ordered(s):
    if (s & 0x7C00) == 0x7C00:
        return 0x8000
    return (~s) & 0xFFFF if s & 0x8000 else s | 0x8000


def _from_ordered(t):
    return t & 0x7FFF if t & 0x8000 else (~t) & 0xFFFF


_CHAIN = [(0, 4), (4, 8), (8, 12), (0, 1), (4, 5), (8, 9), (12, 13), (1, 2), (5, 6), (9, 10), (13, 14), (2, 3), (6, 7), (10, 11), (14, 15)]


def _unpack(b):
    """-> (16 half bit patterns, bytes consumed)"""
    t = [0] * 16
    t[0] = (b[0] << 8) | b[1]
    if b[2] >= 13 << 2:
        return [_from_ordered(t[0])] * 16, 3
    shift = b[2] >> 2
    bits = int.from_bytes(bytes(b[2:14]), "big") & ((1 << 90) - 1)  # 6 bits of shift, then 15 x 6 bits
    for k, (a, c) in enumerate(_CHAIN):
        r = (bits >> (6 * (14 - k))) & 0x3F
        t[c] = (t[a] + (r << shift) - (0x20 << shift)) & 0xFFFF
    return [_from_ordered(v) for v in t], 14


def _decode_channel(data, pos, w, h):
    out = np.zeros((h, w), dtype=np.uint16)
    for y in range(0, h, 4):
        for x in range(0, w, 4):
            s, n = _unpack(data[pos:pos + 14] + b"\0" * 14)
            pos += n
            blk = np.array(s, dtype=np.uint16).reshape(4, 4)
            out[y:y + 4, x:x + 4] = blk[: min(4, h - y), : min(4, w - x)]
    return out, pos


def _chunks(file_bytes, h, lines_per_block=32):
    """payloads of a single-part scanline file written by ir_exr_save (header ends with the first empty attribute name)"""
    pos = 8
    while file_bytes[pos] != 0:
        pos = file_bytes.index(b"\0", pos) + 1
        pos = file_bytes.index(b"\0", pos) + 1
        (n,) = struct.unpack_from("<i", file_bytes, pos)
        pos += 4 + n
    pos += 1
    n_chunks = (h + lines_per_block - 1) // lines_per_block
    out = []
    for k in range(n_chunks):
        (off,) = struct.unpack_from("<Q", file_bytes, pos + 8 * k)
        y, size = struct.unpack_from("<ii", file_bytes, off)
        out.append((y, file_bytes[off + 8: off + 8 + size]))
    return out


def _b44_file(chs, w, h, payload, comp=6, plinear=None):
    """single-chunk scanline file (h <= 32) with the given compressed payload"""
    assert h <= 32
    chl = b""
    for name, t in chs:
        chl += name.encode() + b"\0" + struct.pack("<iBBBBii", t, 1 if plinear and name in plinear else 0, 0, 0, 0, 1, 1)
    hdr = struct.pack("<ii", 20000630, 2)
    hdr += _attr("channels", "chlist", chl + b"\0")
    hdr += _attr("compression", "compression", bytes([comp]))
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    hdr += _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    return hdr + struct.pack("<Q", len(hdr) + 8) + struct.pack("<ii", 0, len(payload)) + payload


def _smooth(w, h, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([0.5 + 0.4 * np.sin(x / 9.0 + seed), 2.0 + np.cos(y / 7.0) * np.sin(x / 13.0), 0.05 + (x + y) / (w + h)], axis=-1)
    img *= 1.0 + 0.01 * rng.random((h, w, 3), dtype=np.float32)
    return img.reshape(-1, 3).astype(np.float32)


FLAT14 = bytes.fromhex("bc00" "020820" "820820" "820820" "820820")     # 16 x 1.0: shift 0, every difference 0 (+32)
RAMP14 = bytes.fromhex("bc00" "020820" "861861" "861861" "861861")     # rows of 0x3c00 0x3c01 0x3c02 0x3c03


def test_hand_derived_blocks(tmp_path):
    """Byte vectors worked out by hand from the layout (t[0], 6-bit shift, fifteen 6-bit differences biased by 32)."""
    chs = [("B", 1), ("G", 1), ("R", 1)]
    p = tmp_path / "v.exr"
    p.write_bytes(_b44_file(chs, 4, 4, FLAT14 + RAMP14 + bytes.fromhex("c500fc"), comp=7))
    got = pta.ExrImage.load_exr_image(p).pixels_buffer.reshape(4, 4, 3)
    assert (got[..., 2] == 1.0).all()                                                   # B
    ramp = np.array([0x3C00, 0x3C01, 0x3C02, 0x3C03], dtype=np.uint16).view(np.float16).astype(np.float32)
    assert np.array_equal(got[..., 1], np.tile(ramp, (4, 1)))                            # G
    assert (got[..., 0] == np.float32(np.array([0x4500], dtype=np.uint16).view(np.float16)[0])).all()   # R: 3-byte block, 5.0
    # the writer produces the same bytes for those pixels
    src = np.zeros((4, 4, 3), dtype=np.float32)
    src[..., 2] = 1.0
    src[..., 1] = np.tile(ramp, (4, 1))
    src[..., 0] = 5.0
    q = tmp_path / "w.exr"
    pta.ExrImage(src.reshape(-1, 3), 4, 4).save(q, compression=7, pixel_type=1)
    assert _chunks(q.read_bytes(), 4)[0][1] == FLAT14[:2] + b"\xfc" + RAMP14 + bytes.fromhex("c500fc")
    pta.ExrImage(src.reshape(-1, 3), 4, 4).save(q, compression=6, pixel_type=1)     # B44 has no 3-byte form
    assert _chunks(q.read_bytes(), 4)[0][1][:28] == FLAT14 + RAMP14


@pytest.mark.parametrize("comp", [6, 7])
@pytest.mark.parametrize("w,h", [(4, 4), (7, 5), (64, 33), (41, 70), (1, 1), (3, 37)])
def test_reader_equals_python_restatement(tmp_path, comp, w, h):
    """C++ writer -> C++ reader gives exactly what the Python decoder of this file makes of the same chunks; chunk sizes
    are the format's fixed rate."""
    src = _smooth(w, h, comp)
    if comp == 7:
        src.reshape(h, w, 3)[: h // 2, :, 0] = 0.25          # flat regions: 3-byte blocks
    p = tmp_path / "a.exr"
    pta.ExrImage(src, w, h).save(p, compression=comp, pixel_type=1)
    back = pta.ExrImage.load_exr_image(p).pixels_buffer.reshape(h, w, 3)
    want = np.zeros((h, w, 3), dtype=np.float32)
    for y0, payload in _chunks(p.read_bytes(), h):
        bh = min(32, h - y0)
        if len(payload) == bh * w * 6:                       # not smaller than the raw block: stored raw, line by line
            raw = np.frombuffer(payload, dtype=np.float16).reshape(bh, 3, w)
            for c in range(3):
                want[y0:y0 + bh, :, 2 - c] = raw[:, c, :]
            continue
        pos = 0
        for c in range(3):                                   # file order B, G, R
            plane, pos = _decode_channel(payload, pos, w, bh)
            want[y0:y0 + bh, :, 2 - c] = plane.view(np.float16).astype(np.float32)
        assert pos == len(payload)
        if comp == 6:
            assert len(payload) == 3 * 14 * ((w + 3) // 4) * ((bh + 3) // 4)
    assert back.tobytes() == want.tobytes()
    half = src.astype(np.float16).astype(np.float32).reshape(h, w, 3)
    assert np.allclose(back, half, rtol=0.04, atol=1e-3)     # smooth data: a few steps of 2^shift in half's bit pattern


@pytest.mark.parametrize("comp", [6, 7])
def test_slowly_varying_bit_patterns_are_stored_exactly(tmp_path, comp):
    """Differences below 32 along the block's chain fit at shift 0: nothing is rounded."""
    w, h = 37, 45
    rng = np.random.default_rng(3)
    bits = 0x3800 + rng.integers(-15, 16, (1, w, 3)).cumsum(axis=1) + rng.integers(-15, 16, (h, 1, 3)).cumsum(axis=0)
    src = bits.astype(np.uint16).view(np.float16).astype(np.float32).reshape(-1, 3)
    p = tmp_path / "e.exr"
    pta.ExrImage(src, w, h).save(p, compression=comp, pixel_type=1)
    assert pta.ExrImage.load_exr_image(p).pixels_buffer.tobytes() == src.tobytes()
    neg = -src                                                 # negative values order the other way round
    pta.ExrImage(neg, w, h).save(p, compression=comp, pixel_type=1)
    assert pta.ExrImage.load_exr_image(p).pixels_buffer.tobytes() == neg.tobytes()


def test_b44a_flat_image_is_three_bytes_per_block_and_exact(tmp_path):
    w, h = 64, 64
    src = np.tile(np.array([[0.5, 1.5, 1024.0]], dtype=np.float32), (w * h, 1))
    pa, pb = tmp_path / "a.exr", tmp_path / "b.exr"
    pta.ExrImage(src, w, h).save(pa, compression=7, pixel_type=1)
    pta.ExrImage(src, w, h).save(pb, compression=6, pixel_type=1)
    assert sum(len(c) for _, c in _chunks(pa.read_bytes(), h)) == 3 * 3 * 16 * 16
    assert sum(len(c) for _, c in _chunks(pb.read_bytes(), h)) == 3 * 14 * 16 * 16
    for p in (pa, pb):
        assert pta.ExrImage.load_exr_image(p).pixels_buffer.tobytes() == src.tobytes()


def test_wide_range_block_and_non_finite_values(tmp_path):
    """The largest value of a block comes back as exactly as the shift allows; infinities and NaNs are stored as zero."""
    w, h = 8, 4
    src = np.full((h, w, 3), 1.0, dtype=np.float32)
    src[0, 0] = (65504.0, 6.0e-5, 1000.0)
    src[1, 5] = (np.inf, np.nan, -np.inf)
    p = tmp_path / "r.exr"
    pta.ExrImage(src.reshape(-1, 3), w, h).save(p, compression=6, pixel_type=1)
    back = pta.ExrImage.load_exr_image(p).pixels_buffer.reshape(h, w, 3)
    assert np.isfinite(back).all()
    assert back[0, 0, 0] == 65504.0 and back[0, 0, 2] == 1000.0        # the block maxima
    assert abs(back[1, 5]).max() < 0.1                                  # non-finite -> zero (up to the block's step)
    assert np.allclose(back[2:, :, :], 1.0, rtol=0.1)


def test_float_channels_travel_uncompressed(tmp_path):
    """B44 leaves FLOAT samples alone: channel after channel in the block; the writer then stores the block raw (no gain)."""
    w, h = 6, 5
    rng = np.random.default_rng(5)
    planes = {k: rng.random((h, w)).astype(np.float32) for k in "BGR"}
    half_g = (rng.integers(0x3800, 0x3810, (h, w))).astype(np.uint16)
    payload = planes["B"].tobytes()
    for y in range(0, h, 4):                                   # G is HALF: blocks via the 3-byte / 14-byte forms
        for x in range(0, w, 4):
            v = int(half_g[y, x])
            payload += struct.pack(">H", v | 0x8000) + b"\xfc"
            half_g[y:y + 4, x:x + 4] = v
    payload += planes["R"].tobytes()
    p = tmp_path / "f.exr"
    p.write_bytes(_b44_file([("B", 2), ("G", 1), ("R", 2)], w, h, payload, comp=7))
    got = pta.ExrImage.load_exr_image(p).pixels_buffer.reshape(h, w, 3)
    assert np.array_equal(got[..., 0], planes["R"]) and np.array_equal(got[..., 2], planes["B"])
    assert np.array_equal(got[..., 1], half_g.view(np.float16).astype(np.float32))
    src = _smooth(w, h, 1)
    q = tmp_path / "g.exr"
    pta.ExrImage(src, w, h).save(q, compression=6, pixel_type=2)
    assert pta.ExrImage.load_exr_image(q).pixels_buffer.tobytes() == src.tobytes()


def test_plinear_channels_get_eight_log_x(tmp_path):
    """A HALF channel flagged pLinear in the channel list was stored as exp(x / 8): the reader applies 8 log(x)."""
    w, h = 4, 4
    vals = {"B": 1.0, "G": 2.718, "R": 0.0}
    payload = b""
    for k in "BGR":
        hv = int(np.array([vals[k]], dtype=np.float16).view(np.uint16)[0])
        payload += struct.pack(">H", hv | 0x8000) + b"\xfc"
    p = tmp_path / "l.exr"
    p.write_bytes(_b44_file([("B", 1), ("G", 1), ("R", 1)], w, h, payload, comp=7, plinear={"B", "G", "R"}))
    got = pta.ExrImage.load_exr_image(p).pixels_buffer.reshape(h, w, 3)
    assert (got[..., 2] == 0.0).all()                                                    # 8 log 1
    g = np.float32(np.float16(8.0 * np.log(np.float64(np.float16(2.718)))))
    assert (got[..., 1] == g).all()
    assert np.isneginf(got[..., 0]).all()                                                # 8 log 0
    p.write_bytes(_b44_file([("B", 1), ("G", 1), ("R", 1)], w, h, payload, comp=7))
    got = pta.ExrImage.load_exr_image(p).pixels_buffer.reshape(h, w, 3)
    assert (got[..., 2] == 1.0).all() and (got[..., 0] == 0.0).all()


@pytest.mark.parametrize("comp", [6, 7])
def test_corrupted_b44_files_never_crash(tmp_path, comp):
    w, h = 48, 40
    p = tmp_path / "ok.exr"
    pta.ExrImage(_smooth(w, h, 3), w, h).save(p, compression=comp, pixel_type=1)
    data = bytearray(p.read_bytes())
    rng = np.random.default_rng(comp)
    q = tmp_path / "bad.exr"
    errors = 0
    for trial in range(300):
        bad = bytearray(data)
        if trial % 3 == 0:
            bad = bad[: int(rng.integers(1, len(bad)))]
        else:
            for _ in range(int(rng.integers(1, 6))):
                i = int(rng.integers(300, len(bad)))
                bad[i] ^= 1 << int(rng.integers(0, 8))
        q.write_bytes(bytes(bad))
        try:
            img = pta.ExrImage.load_exr_image(q)
            assert (img.width, img.height) == (w, h)
        except pta.InsplorayError:
            errors += 1
    assert errors > 50
