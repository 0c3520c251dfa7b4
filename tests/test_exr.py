"""EXR reader/writer (ir_exr_load / ir_exr_save, path_tracer_amd/csrc/exr_io.cpp) replacing
ExrImage::load_exr_image (exr.rs:26-49).  The `exr` crate and the reference's assets are absent, so loader parity
is UNPINNED by the reference; these are writer->reader round trips plus hand-built files of the public format."""
import os
import struct
import zlib

import numpy as np
import pytest

import path_tracer_amd as pta


def _img(w, h, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.random((h * w, 3), dtype=np.float32) * 4.0
    a[0] = (0.0, 1.0, 65504.0)
    return a


@pytest.mark.parametrize("comp", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (64, 33), (130, 17), (33, 70)])
def test_float_round_trip_is_exact(tmp_path, comp, w, h):
    img = pta.ExrImage(_img(w, h, comp), w, h)
    p = tmp_path / "a.exr"
    img.save(p, compression=comp, pixel_type=2)
    back = pta.ExrImage.load_exr_image(p)
    assert (back.width, back.height) == (w, h)
    assert back.pixels_buffer.tobytes() == img.pixels_buffer.tobytes()


@pytest.mark.parametrize("comp", [0, 1, 3, 4, 5])  # (the lossy 6 and 7: tests/test_exr_b44.py)
def test_half_round_trip(tmp_path, comp):
    w, h = 40, 21
    src = _img(w, h, 5)
    img = pta.ExrImage(src, w, h)
    p = tmp_path / "h.exr"
    img.save(p, compression=comp, pixel_type=1)
    back = pta.ExrImage.load_exr_image(p)
    assert np.array_equal(back.pixels_buffer, src.astype(np.float16).astype(np.float32))


def test_pxr24_rounds_floats_to_24_bits(tmp_path):
    """PXR24 keeps sign, exponent and 15 mantissa bits of FLOAT samples (round to nearest, never into infinity)."""
    w, h = 50, 37
    src = _img(w, h, 9)
    src[1] = (3.4028235e38, -1.0e-30, 1.0000001)
    p = tmp_path / "p.exr"
    pta.ExrImage(src, w, h).save(p, compression=5, pixel_type=2)
    back = pta.ExrImage.load_exr_image(p).pixels_buffer
    bits = src.view(np.uint32).astype(np.uint64)
    rounded = ((bits + ((bits & 0x80) )) >> 8 << 8).astype(np.uint32)
    big = (rounded & 0x7FFFFFFF) >= 0x7F800000                       # rounding would have produced infinity: truncated
    rounded[big] = (bits[big] >> 8 << 8).astype(np.uint32)
    assert np.array_equal(back.view(np.uint32), rounded)
    assert np.allclose(back, src, rtol=2.0 ** -15, atol=0) and np.isfinite(back).all()


@pytest.mark.parametrize("kind", ["zeros", "constant", "two-level", "ramp16", "sky"])
def test_piz_special_images(tmp_path, kind):
    """PIZ corner cases: a single symbol, long runs (the run-length escape), 14-bit and 16-bit wavelet modes."""
    w, h = 96, 70                                              # three 32-line blocks, the last one partial
    if kind == "zeros":
        src = np.zeros((h * w, 3), dtype=np.float32)
    elif kind == "constant":
        src = np.full((h * w, 3), 0.5, dtype=np.float32)
    elif kind == "two-level":
        src = np.where((np.arange(h * w) // 500 % 2 == 0)[:, None], 1.0, 2.0).astype(np.float32).repeat(3, 1)
    elif kind == "ramp16":                                     # > 2^14 distinct 16-bit values in a block: modulo-2^16 wavelet
        src = (np.arange(h * w * 3, dtype=np.float32).reshape(-1, 3) * 0.37 + 1.0)
    else:
        from path_tracer_amd import workloads as W
        src = W.synthetic_sky(w, h).pixels_buffer
    for pt, conv in ((2, lambda a: a), (1, lambda a: a.astype(np.float16).astype(np.float32))):
        p = tmp_path / f"{kind}{pt}.exr"
        pta.ExrImage(src, w, h).save(p, compression=4, pixel_type=pt)
        back = pta.ExrImage.load_exr_image(p)
        assert np.array_equal(back.pixels_buffer.view(np.uint32), conv(src).view(np.uint32)), kind
    if kind in ("zeros", "constant", "two-level"):
        assert os.path.getsize(p) < w * h * 6 // 10              # the run-length escape is doing its job


def test_smooth_image_actually_compresses(tmp_path):
    w, h = 256, 64
    x = np.linspace(0, 1, w, dtype=np.float32)[None, :].repeat(h, 0)
    src = np.stack([x, x * 0.5, np.ones_like(x)], -1).reshape(-1, 3)
    img = pta.ExrImage(src, w, h)
    sizes = {}
    for comp in (0, 1, 3, 4):
        p = tmp_path / f"c{comp}.exr"
        img.save(p, compression=comp)
        sizes[comp] = os.path.getsize(p)
        assert pta.ExrImage.load_exr_image(p).pixels_buffer.tobytes() == src.tobytes()
    assert sizes[3] < sizes[0] // 2 and sizes[1] < sizes[0] and sizes[4] < sizes[0] // 2


def _attr(name, typ, data):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data


def _chlist(chs):
    out = b""
    for name, t in chs:
        out += name.encode() + b"\0" + struct.pack("<iBBBBii", t, 0, 0, 0, 0, 1, 1)
    return out + b"\0"


def _handmade(chs, w, h, xmin, ymin, planes, comp=0, line_order=0):
    """A scanline file written here in Python, independent of the C++ writer: channels in `chs` order (sorted),
    FLOAT/UINT/HALF by type, data window offset (xmin, ymin), lines written in decreasing y if line_order=1."""
    hdr = struct.pack("<ii", 20000630, 2)
    hdr += _attr("channels", "chlist", _chlist(chs))
    hdr += _attr("compression", "compression", bytes([comp]))
    box = struct.pack("<iiii", xmin, ymin, xmin + w - 1, ymin + h - 1)
    hdr += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    hdr += _attr("lineOrder", "lineOrder", bytes([line_order]))
    hdr += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hdr += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    chunks = []
    for y in range(h):
        line = b""
        for name, t in chs:
            v = planes[name][y]
            line += v.astype({0: np.uint32, 1: np.float16, 2: np.float32}[t]).tobytes()
        if comp == 2:  # ZIPS: filter then deflate
            b = np.frombuffer(line, dtype=np.uint8)
            re = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)
            d = re.copy()
            d[1:] = (re[1:] - re[:-1] + 128 + 256) & 0xFF
            z = zlib.compress(d.astype(np.uint8).tobytes())
            if len(z) < len(line):
                line = z
        chunks.append((ymin + y, line))
    order = chunks if line_order == 0 else chunks[::-1]
    table_pos = len(hdr)
    body = b""
    offsets = {}
    for y, data in order:
        offsets[y] = table_pos + 8 * h + len(body)
        body += struct.pack("<ii", y, len(data)) + data
    table = b"".join(struct.pack("<Q", offsets[ymin + y]) for y in range(h))
    return hdr + table + body


def test_handmade_file_with_alpha_offset_window_and_mixed_types(tmp_path):
    w, h = 9, 6
    rng = np.random.default_rng(1)
    planes = {"A": rng.random((h, w)), "B": rng.random((h, w)), "G": rng.integers(0, 1000, (h, w)).astype(np.float64),
              "R": rng.random((h, w)), "Z": rng.random((h, w))}
    chs = [("A", 2), ("B", 1), ("G", 0), ("R", 2), ("Z", 2)]  # alpha + extra channel dropped; HALF B, UINT G
    for comp, lo in [(0, 0), (2, 0), (0, 1)]:
        p = tmp_path / f"m{comp}{lo}.exr"
        p.write_bytes(_handmade(chs, w, h, -3, 10, planes, comp=comp, line_order=lo))
        img = pta.ExrImage.load_exr_image(p)
        assert (img.width, img.height) == (w, h)
        got = img.pixels_buffer.reshape(h, w, 3)
        assert np.array_equal(got[..., 0], planes["R"].astype(np.float32))
        assert np.array_equal(got[..., 1], planes["G"].astype(np.uint32).astype(np.float32))
        assert np.array_equal(got[..., 2], planes["B"].astype(np.float16).astype(np.float32))


def test_layer_prefixed_channels(tmp_path):
    w, h = 4, 3
    rng = np.random.default_rng(2)
    planes = {k: rng.random((h, w)) for k in ("beauty.B", "beauty.G", "beauty.R", "depth.Z")}
    chs = [(k, 2) for k in sorted(planes)]
    p = tmp_path / "layer.exr"
    p.write_bytes(_handmade(chs, w, h, 0, 0, planes))
    got = pta.ExrImage.load_exr_image(p).pixels_buffer.reshape(h, w, 3)
    assert np.array_equal(got[..., 0], planes["beauty.R"].astype(np.float32))


def test_errors_are_results_not_crashes(tmp_path):
    # exr.rs:42-48 returns Err; Scene::get_example_scene then falls back to skybox None (scene.rs:45-60)
    with pytest.raises(pta.InsplorayError):
        pta.ExrImage.load_exr_image(tmp_path / "missing.exr")
    p = tmp_path / "junk.exr"
    p.write_bytes(b"not an exr at all")
    with pytest.raises(pta.InsplorayError):
        pta.ExrImage.load_exr_image(p)
    good = tmp_path / "good.exr"
    pta.ExrImage(_img(16, 16), 16, 16).save(good, compression=3)
    data = good.read_bytes()
    for cut in (10, 200, len(data) - 5):
        p.write_bytes(data[:cut])
        with pytest.raises(pta.InsplorayError):
            pta.ExrImage.load_exr_image(p)
    # unsupported compression is reported by name
    w, h = 4, 2
    planes = {k: np.zeros((h, w)) for k in "BGR"}
    p.write_bytes(_handmade([(k, 2) for k in "BGR"], w, h, 0, 0, planes, comp=0).replace(
        b"compression\0compression\0\x01\0\0\0\0", b"compression\0compression\0\x01\0\0\0\x08"))
    with pytest.raises(pta.InsplorayError, match="DWAA"):
        pta.ExrImage.load_exr_image(p)
    # no RGB layer
    p.write_bytes(_handmade([("Y", 2)], w, h, 0, 0, {"Y": np.zeros((h, w))}))
    with pytest.raises(pta.InsplorayError, match="R, G and B"):
        pta.ExrImage.load_exr_image(p)


def test_example_scene_falls_back_without_asset(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    sc = pta.Scene.get_example_scene()
    assert sc.skybox is None and sc.default_sky_color == [0.6, 0.7, 0.9]
    assert len(sc.spheres) == 2 and len(sc.materials) == 2
    assert sc.spheres[1].position == [0.0, -100.5, 0.0] and sc.spheres[1].radius == 100.0
    assert "Failed loading EXR" in capsys.readouterr().err
    os.makedirs("assets/env")
    pta.ExrImage(_img(8, 4), 8, 4).save("assets/env/default_skybox_1.exr")
    sc = pta.Scene.get_example_scene()
    assert sc.skybox is not None and (sc.skybox.width, sc.skybox.height) == (8, 4)


def _zip_block(raw):
    """OpenEXR ZIP/ZIPS block: de-interleave even/odd bytes, delta-code, deflate; stored raw if that is not smaller."""
    b = np.frombuffer(raw, dtype=np.uint8)
    re = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)
    d = re.copy()
    d[1:] = (re[1:] - re[:-1] + 128 + 256) & 0xFF
    z = zlib.compress(d.astype(np.uint8).tobytes())
    return z if len(z) < len(raw) else raw


def _part_header(chs, w, h, comp, tiles=None, name=None, chunk_count=None):
    hdr = _attr("channels", "chlist", _chlist(chs))
    hdr += _attr("compression", "compression", bytes([comp]))
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    hdr += _attr("lineOrder", "lineOrder", bytes([0]))
    hdr += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hdr += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    if tiles is not None:
        hdr += _attr("tiles", "tiledesc", struct.pack("<IIB", tiles[0], tiles[1], 0))
    if name is not None:  # multi-part attributes
        hdr += _attr("name", "string", name.encode())
        hdr += _attr("type", "string", b"tiledimage" if tiles is not None else b"scanlineimage")
        hdr += _attr("chunkCount", "int", struct.pack("<i", chunk_count))
    return hdr + b"\0"


def _tile_chunks(chs, w, h, planes, comp, tw, th):
    """[(tile header bytes, data)] in the offset table's order (ty major), level 0 only."""
    out = []
    for ty in range((h + th - 1) // th):
        for tx in range((w + tw - 1) // tw):
            x0, y0 = tx * tw, ty * th
            bw, bh = min(tw, w - x0), min(th, h - y0)
            raw = b""
            for ly in range(bh):
                for name, t in chs:
                    v = planes[name][y0 + ly, x0:x0 + bw]
                    raw += v.astype({0: np.uint32, 1: np.float16, 2: np.float32}[t]).tobytes()
            out.append((struct.pack("<iiii", tx, ty, 0, 0), _zip_block(raw) if comp in (2, 3) else raw))
    return out


def _line_chunks(chs, w, h, planes, comp):
    lpb = {0: 1, 2: 1, 3: 16}[comp]
    out = []
    for y0 in range(0, h, lpb):
        raw = b""
        for y in range(y0, min(h, y0 + lpb)):
            for name, t in chs:
                raw += planes[name][y].astype({0: np.uint32, 1: np.float16, 2: np.float32}[t]).tobytes()
        out.append((struct.pack("<i", y0), _zip_block(raw) if comp in (2, 3) else raw))
    return out


def _assemble(version, headers, parts_chunks, multipart):
    """File = magic, version, headers, offset tables (part order), chunks."""
    head = struct.pack("<ii", 20000630, version) + b"".join(headers) + (b"\0" if multipart else b"")
    n_entries = sum(len(c) for c in parts_chunks)
    pos = len(head) + 8 * n_entries
    tables, body = b"", b""
    for pi, chunks in enumerate(parts_chunks):
        for hdr, data in chunks:
            tables += struct.pack("<Q", pos + len(body))
            body += (struct.pack("<i", pi) if multipart else b"") + hdr + struct.pack("<i", len(data)) + data
    return head + tables + body


@pytest.mark.parametrize("comp", [0, 3])
@pytest.mark.parametrize("tile", [(4, 4), (5, 3), (32, 32)])
def test_handmade_tiled_file(tmp_path, comp, tile):
    """Single-part tiled files (version bit 0x200) written here in Python: ragged edge tiles, HALF and UINT channels."""
    w, h = 13, 9
    rng = np.random.default_rng(7)
    planes = {"A": rng.random((h, w)), "B": rng.random((h, w)), "G": rng.integers(0, 500, (h, w)).astype(np.float64),
              "R": rng.random((h, w))}
    chs = [("A", 1), ("B", 1), ("G", 0), ("R", 2)]
    data = _assemble(2 | 0x200, [_part_header(chs, w, h, comp, tiles=tile)], [_tile_chunks(chs, w, h, planes, comp, *tile)], False)
    p = tmp_path / "tiled.exr"
    p.write_bytes(data)
    img = pta.ExrImage.load_exr_image(p)
    assert (img.width, img.height) == (w, h)
    got = img.pixels_buffer.reshape(h, w, 3)
    assert np.array_equal(got[..., 0], planes["R"].astype(np.float32))
    assert np.array_equal(got[..., 1], planes["G"].astype(np.uint32).astype(np.float32))
    assert np.array_equal(got[..., 2], planes["B"].astype(np.float16).astype(np.float32))


@pytest.mark.parametrize("rgb_part_tiled", [False, True])
def test_handmade_multipart_file(tmp_path, rgb_part_tiled):
    """Multi-part file (version bit 0x1000): part 0 holds depth only, part 1 is the first part with R, G and B
    (exr.rs:27 read_first_rgba_layer_from_file), part 2 another RGB layer that must not be picked."""
    w, h = 11, 21
    rng = np.random.default_rng(8)
    z = {"Z": rng.random((h, w))}
    rgb = {k: rng.random((h, w)) for k in "BGR"}
    other = {k: rng.random((h, w)) for k in "BGR"}
    c3 = [(k, 2) for k in "BGR"]
    tile = (4, 8)
    chunks = [_line_chunks([("Z", 2)], w, h, z, 3),
              _tile_chunks(c3, w, h, rgb, 3, *tile) if rgb_part_tiled else _line_chunks(c3, w, h, rgb, 2),
              _line_chunks(c3, w, h, other, 0)]
    headers = [_part_header([("Z", 2)], w, h, 3, name="depth", chunk_count=len(chunks[0])),
               _part_header(c3, w, h, 3 if rgb_part_tiled else 2, tiles=tile if rgb_part_tiled else None, name="beauty",
                            chunk_count=len(chunks[1])),
               _part_header(c3, w, h, 0, name="other", chunk_count=len(chunks[2]))]
    p = tmp_path / "multi.exr"
    p.write_bytes(_assemble(2 | 0x1000, headers, chunks, True))
    got = pta.ExrImage.load_exr_image(p).pixels_buffer.reshape(h, w, 3)
    for k, name in enumerate("RGB"):
        assert np.array_equal(got[..., k], rgb[name].astype(np.float32))


def test_crafted_tile_coordinates_and_counts_are_rejected(tmp_path):
    """Hostile headers: tile origins that overflow 32 bits, tile sizes with the sign bit set, a chunkCount the file cannot
    hold.  Each must come back as an error (the first one used to write 786 KB in front of the image)."""
    w, h = 8, 2
    planes = {k: np.ones((h, w)) for k in "BGR"}
    c3 = [(k, 2) for k in "BGR"]
    p = tmp_path / "evil.exr"
    # tile_w = 65536, tile_h = 1; one chunk claims tx = 65535: x0 = 65535 * 65536 wraps to -65536 in 32 bits
    tw, th = 65536, 1
    chunks = _tile_chunks(c3, w, h, planes, 0, tw, th)
    evil = [(struct.pack("<iiii", 65535, 0, 0, 0), np.ones(65536 * 3, dtype=np.float32).tobytes())] + chunks[1:]
    p.write_bytes(_assemble(2 | 0x200, [_part_header(c3, w, h, 0, tiles=(tw, th))], [evil], False))
    with pytest.raises(pta.InsplorayError, match="outside the data window"):
        pta.ExrImage.load_exr_image(p)
    for tw, th in [(0x80000000, 1), (1, 0xFFFFFFF0), (65537, 4)]:
        p.write_bytes(_assemble(2 | 0x200, [_part_header(c3, w, h, 0, tiles=(tw, th))], [chunks], False))
        with pytest.raises(pta.InsplorayError, match="tile size"):
            pta.ExrImage.load_exr_image(p)
    lines = _line_chunks(c3, w, h, planes, 0)
    for count in (0x7FFFFFFF, 1 << 20):
        p.write_bytes(_assemble(2 | 0x1000, [_part_header(c3, w, h, 0, name="a", chunk_count=count)], [lines], True))
        with pytest.raises(pta.InsplorayError, match="offset table"):
            pta.ExrImage.load_exr_image(p)


@pytest.mark.parametrize("comp", [3, 4, 5])
def test_corrupted_files_never_crash(tmp_path, comp):
    """Bit flips and truncations of valid files: the loader returns an error or an image, it never faults."""
    w, h = 48, 40
    p = tmp_path / "ok.exr"
    pta.ExrImage(_img(w, h, 3), w, h).save(p, compression=comp, pixel_type=1)
    data = bytearray(p.read_bytes())
    rng = np.random.default_rng(comp)
    q = tmp_path / "bad.exr"
    outcomes = {"error": 0, "image": 0}
    for trial in range(300):
        bad = bytearray(data)
        if trial % 3 == 0:
            bad = bad[: int(rng.integers(1, len(bad)))]
        else:
            for _ in range(int(rng.integers(1, 6))):
                i = int(rng.integers(300, len(bad)))          # past the header: hit tables and chunk data
                bad[i] ^= 1 << int(rng.integers(0, 8))
        q.write_bytes(bytes(bad))
        try:
            img = pta.ExrImage.load_exr_image(q)
            assert (img.width, img.height) == (w, h)
            outcomes["image"] += 1
        except pta.InsplorayError:
            outcomes["error"] += 1
    assert outcomes["error"] > 50


def test_png_save_round_trip(tmp_path):
    """ir_png_save ("save image", a reference TODO): the resolved 0xAARRGGBB frame, bottom row first, as an RGB PNG read
    back with an independent decoder (Pillow)."""
    from PIL import Image
    from path_tracer_amd import _capi
    from path_tracer_amd.file_formats import save_png
    rng = np.random.default_rng(3)
    for (w, h) in ((1, 1), (7, 5), (640, 360)):
        argb = rng.integers(0, 2 ** 32, size=w * h, dtype=np.uint64).astype(np.uint32)
        path = tmp_path / f"f_{w}x{h}.png"
        save_png(path, argb, w, h)
        img = np.asarray(Image.open(path))
        assert img.shape == (h, w, 3) and img.dtype == np.uint8
        a = argb.reshape(h, w)[::-1]
        want = np.stack([(a >> 16) & 0xFF, (a >> 8) & 0xFF, a & 0xFF], axis=-1).astype(np.uint8)
        assert np.array_equal(img, want)
    with pytest.raises(_capi.InsplorayError):
        save_png(tmp_path / "no_such_dir" / "x.png", np.zeros(4, dtype=np.uint32), 2, 2)
    with pytest.raises(ValueError):
        save_png(tmp_path / "x.png", np.zeros(5, dtype=np.uint32), 2, 2)


def test_reader_under_address_sanitizer(tmp_path):
    """The reader compiled for the host with -fsanitize=address,undefined (tests/cpp/exr_asan_main.cpp) over tiled,
    multi-part, crafted and randomly corrupted files: no sanitizer report, every file answered with a code."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "path_tracer_amd", "csrc")
    exe = tmp_path / "exr_asan"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan",
                           os.path.join(root, "tests", "cpp", "exr_asan_main.cpp"), os.path.join(csrc, "exr_io.cpp"),
                           os.path.join(csrc, "exr_codecs.cpp"), "-lz", "-o", str(exe)])
    w, h = 13, 9
    rng = np.random.default_rng(11)
    planes = {k: rng.random((h, w)) for k in "BGR"}
    c3 = [(k, 1) for k in "BGR"]
    files = []

    def add(name, data):
        q = tmp_path / name
        q.write_bytes(bytes(data))
        files.append(str(q))

    tiled = _assemble(2 | 0x200, [_part_header(c3, w, h, 3, tiles=(5, 4))], [_tile_chunks(c3, w, h, planes, 3, 5, 4)], False)
    add("tiled.exr", tiled)
    lines = _line_chunks(c3, w, h, planes, 3)
    multi = _assemble(2 | 0x1000, [_part_header(c3, w, h, 3, name="a", chunk_count=len(lines)),
                                   _part_header(c3, w, h, 3, tiles=(5, 4), name="b", chunk_count=6)],
                      [lines, _tile_chunks(c3, w, h, planes, 3, 5, 4)], True)
    add("multi.exr", multi)
    evil_chunks = [(struct.pack("<iiii", 65535, 0, 0, 0), np.ones(65536 * 3, dtype=np.float16).tobytes())]
    add("evil.exr", _assemble(2 | 0x200, [_part_header(c3, 8, 2, 0, tiles=(65536, 1))], [evil_chunks], False))
    add("count.exr", _assemble(2 | 0x1000, [_part_header(c3, w, h, 3, name="a", chunk_count=0x7FFFFFFF)], [lines], True))
    for base, tag in ((tiled, "t"), (multi, "m")):
        for trial in range(150):
            bad = bytearray(base)
            if trial % 4 == 0:
                bad = bad[: int(rng.integers(1, len(bad)))]
            else:
                for _ in range(int(rng.integers(1, 5))):
                    i = int(rng.integers(8, len(bad)))  # headers included: tile sizes, windows, counts
                    bad[i] ^= 1 << int(rng.integers(0, 8))
            add(f"{tag}{trial}.exr", bad)
    for comp in (6, 7):  # B44 / B44A blocks from the writer, whole and damaged (tests/test_exr_b44.py has the format checks)
        q = tmp_path / f"b44_{comp}.exr"
        pta.ExrImage(_img(21, 37, comp), 21, 37).save(q, compression=comp, pixel_type=1)
        base = q.read_bytes()
        files.append(str(q))
        for trial in range(100):
            bad = bytearray(base)
            if trial % 4 == 0:
                bad = bad[: int(rng.integers(1, len(bad)))]
            else:
                for _ in range(int(rng.integers(1, 5))):
                    i = int(rng.integers(8, len(bad)))
                    bad[i] ^= 1 << int(rng.integers(0, 8))
            add(f"b{comp}_{trial}.exr", bad)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([str(exe)] + files, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == len(files)
    assert out[0].startswith("0 13x9") and out[1].startswith("0 13x9") and out[2].startswith("-") and out[3].startswith("-")
