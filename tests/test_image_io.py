"""Image decode / encode rows (L1 data prep in, L0 PNG out), CPU only.

These rows ARE pinned against the reference: tests/golden/png_decode.npz and
png_encode.json hold what the reference's own vendored stb (v2.27 / v1.16,
built from /root/reference/vendor as oracle/_ref) produces, and where that
build is present the comparison is also made live.
"""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


def _decode_fixture():
    return np.load(os.path.join(GOLDEN, "png_decode.npz"))


def _names(data):
    return sorted({k.split("/")[0] for k in data.files})


def test_decode_matches_reference_stb_golden(hmrm):
    """stbi_load(path,&w,&h,&n,req_comp) (hmap.cpp:320-321,341-342) for req_comp 0..4."""
    data = _decode_fixture()
    checked = 0
    for name in _names(data):
        blob = data[name + "/bytes"].tobytes()
        for req in range(5):
            key = f"{name}/req{req}"
            if key not in data.files:
                continue
            arr, n = hmrm.image_load_memory(blob, req)
            assert n == int(data[f"{name}/n{req}"][0]), (name, req)
            assert arr.shape == data[key].shape, (name, req)
            assert np.array_equal(arr, data[key]), (name, req)
            checked += 1
    assert checked > 150


def test_decode_matches_reference_stb_live(hmrm, stb_ref):
    if stb_ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference here); golden vectors cover it")
    import make_golden
    for name, blob in make_golden.png_fixture_files().items():
        for req in range(5):
            exp, n = stb_ref.load(blob, req)
            if name == "p6_16" and req not in (0, 3):
                with pytest.raises(hmrm.HmrmError):
                    hmrm.image_load_memory(blob, req)
                continue
            arr, n2 = hmrm.image_load_memory(blob, req)
            assert n2 == n and np.array_equal(arr, exp), (name, req)


def test_jpeg_decode_matches_reference_stb_golden(hmrm):
    """JPEG maps (README.md: "any format supported by stb_image.h: JPEG, ..."; sample_config.txt uses
    .jpg): IDCT, chroma upsampling and colour conversion must be stb's, pixel for pixel."""
    data = np.load(os.path.join(GOLDEN, "jpeg_decode.npz"))
    checked = 0
    for name in _names(data):
        blob = data[name + "/bytes"].tobytes()
        for req in range(5):
            arr, n = hmrm.image_load_memory(blob, req)
            assert n == int(data[f"{name}/n{req}"][0]), (name, req)
            assert arr.shape == data[f"{name}/req{req}"].shape and np.array_equal(arr, data[f"{name}/req{req}"]), (name, req)
            checked += 1
    assert checked >= 80


def test_jpeg_decode_matches_reference_stb_live(hmrm, stb_ref):
    if stb_ref is None:
        pytest.skip("oracle/_ref not built; golden vectors cover it")
    pytest.importorskip("PIL")
    import make_golden
    files = make_golden.jpeg_fixture_files()
    for name, blob in files.items():
        for req in range(5):
            exp, n = stb_ref.load(blob, req)
            arr, n2 = hmrm.image_load_memory(blob, req)
            assert n2 == n and np.array_equal(arr, exp), (name, req)
    # a truncated file is refused by both
    blob = files["rgb_420_q50"]
    assert stb_ref.load(blob[: len(blob) // 2], 3)[0] is None
    with pytest.raises(hmrm.HmrmError):
        hmrm.image_load_memory(blob[: len(blob) // 2], 3)


def test_jpeg_random_files_match_reference_stb_live(hmrm, stb_ref):
    """120 random JPEGs (sizes 1..80, all PIL subsamplings, baseline / progressive / optimised tables,
    grey / YCbCr / CMYK, restart intervals) against the reference's stb build, every req_comp."""
    if stb_ref is None:
        pytest.skip("oracle/_ref not built; golden vectors cover the fixed set")
    Image = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.RandomState(4242)
    checked = 0
    for trial in range(120):
        w, h = int(rng.randint(1, 81)), int(rng.randint(1, 81))
        mode = ["RGB", "L", "CMYK"][int(rng.choice([0, 0, 0, 1, 2]))]
        ch = {"RGB": 3, "L": 1, "CMYK": 4}[mode]
        base = rng.randint(0, 256, size=(h // 6 + 1, w // 6 + 1, ch)).astype(np.uint8)
        img = np.kron(base, np.ones((6, 6, 1), dtype=np.uint8))[:h, :w]
        img = np.clip(img.astype(np.int32) + rng.randint(-20, 21, size=img.shape), 0, 255).astype(np.uint8)
        pil = Image.fromarray(img[:, :, 0] if ch == 1 else img, mode)
        kw = dict(quality=int(rng.choice([10, 35, 50, 75, 92, 100])), progressive=bool(rng.randint(2)),
                  optimize=bool(rng.randint(2)))
        if mode == "RGB":
            kw["subsampling"] = int(rng.choice([0, 1, 2]))
        if rng.randint(3) == 0:
            kw["restart_marker_blocks"] = int(rng.randint(1, 9))
        buf = io.BytesIO()
        try:
            pil.save(buf, "JPEG", **kw)
        except (TypeError, ValueError, OSError):
            kw.pop("restart_marker_blocks", None)
            buf = io.BytesIO()
            pil.save(buf, "JPEG", **kw)
        blob = buf.getvalue()
        for req in range(5):
            exp, n = stb_ref.load(blob, req)
            assert exp is not None, (trial, kw)
            arr, n2 = hmrm.image_load_memory(blob, req)
            assert n2 == n and np.array_equal(arr, exp), (trial, mode, w, h, kw, req)
            checked += 1
    assert checked == 600


def test_bmp_tga_decode_matches_reference_stb(hmrm, stb_ref):
    """BMP / TGA maps (README.md "Options": "... TGA, BMP ..."): golden vectors from the reference's stb,
    plus the live comparison where that build exists."""
    data = np.load(os.path.join(GOLDEN, "bmp_tga_decode.npz"))
    checked = 0
    for name in _names(data):
        blob = data[name + "/bytes"].tobytes()
        for req in range(5):
            arr, n = hmrm.image_load_memory(blob, req)
            assert n == int(data[f"{name}/n{req}"][0]), (name, req)
            assert arr.shape == data[f"{name}/req{req}"].shape and np.array_equal(arr, data[f"{name}/req{req}"]), (name, req)
            if stb_ref is not None:
                exp, n2 = stb_ref.load(blob, req)
                assert n2 == n and np.array_equal(arr, exp), (name, req)
            checked += 1
    assert checked >= 110
    # refused like stb refuses them: RLE-compressed BMP, truncated header
    import struct
    rle = b"BM" + struct.pack("<IHHI", 0, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 4, 4, 1, 8, 1, 0, 0, 0, 0, 0)
    for blob in (rle, data["bmp24/bytes"].tobytes()[:20]):
        with pytest.raises(hmrm.HmrmError):
            hmrm.image_load_memory(blob, 3)


def test_gif_psd_pic_hdr_decode_matches_reference_stb(hmrm, stb_ref):
    """GIF, PSD, PIC and Radiance HDR maps (README.md "Options" lists them; csrc/legacy_formats.cpp): golden vectors
    from the reference's own stb build for 52 hand-assembled files -- GIF 87a/89a, global / local palettes,
    transparency, interlacing, sub-rectangles with and without a background index, dictionary resets, truncated data;
    PSD raw / PackBits, 1-5 channels, 16 bits, partial alpha over the white matte; PIC raw / pure / mixed run-length
    packets in several channel layouts; HDR run-length and flat scanlines, both magics -- for req_comp 0..4, plus the
    live comparison where that build exists.  Files the reference refuses are refused."""
    data = np.load(os.path.join(GOLDEN, "legacy_decode.npz"))
    checked = refused = 0
    names = _names(data)
    assert {n.split("_")[0] for n in names} == {"gif", "gif87", "gif89", "psd", "pic", "hdr"}
    for name in names:
        blob = data[name + "/bytes"].tobytes()
        for req in range(5):
            want_n = int(data[f"{name}/n{req}"][0])
            if want_n < 0:
                with pytest.raises(hmrm.HmrmError) as e:
                    hmrm.image_load_memory(blob, req)
                assert e.value.code == hmrm.HMRM_E_IMAGE
                refused += 1
                continue
            arr, n = hmrm.image_load_memory(blob, req)
            exp = data[f"{name}/req{req}"]
            assert n == want_n and arr.shape == exp.shape and np.array_equal(arr, exp), (name, req)
            if stb_ref is not None:
                live, n2 = stb_ref.load(blob, req)
                assert n2 == n and np.array_equal(arr, live), (name, req)
            checked += 1
    assert checked >= 250 and refused >= 5
    # files on which the reference's loader crashes or never returns: refused here, promptly
    for blob in (data["pic_rgb_mixed/bytes"].tobytes()[:-9], data["pic_rgb_raw/bytes"].tobytes()[:104],
                 data["hdr_rle/bytes"].tobytes()[:-40]):
        for req in (0, 3, 4):
            with pytest.raises(hmrm.HmrmError) as e:
                hmrm.image_load_memory(blob, req)
            assert e.value.code == hmrm.HMRM_E_IMAGE


def test_map_files_of_every_listed_format_through_config(hmrm, tmp_path):
    """`heightmap x.gif` / `colormap x.psd` etc.: the config path hands any of the formats the reference's README
    names to the decoders (main/hmap.cpp:320-321 req_comp 3, :341-342 req_comp 4)."""
    data = np.load(os.path.join(GOLDEN, "legacy_decode.npz"))
    for name, ext in (("gif_interlaced", "gif"), ("psd_rgba_rle", "psd"), ("pic_rgb_then_alpha", "pic"), ("hdr_rle", "hdr")):
        p = tmp_path / f"map.{ext}"
        p.write_bytes(data[name + "/bytes"].tobytes())
        cfg = hmrm.Config().consume_string(f"heightmap {p}\ncolormap {p}\n")
        assert np.array_equal(cfg.height_rgb(), data[name + "/req3"]), name
        assert np.array_equal(cfg.color_rgba(), data[name + "/req4"]), name
        cfg.close()


def test_jpeg_heightmap_through_config(hmrm, tmp_path):
    """`heightmap x.jpg` / `colormap x.jpg` as in the reference's sample_config.txt."""
    data = np.load(os.path.join(GOLDEN, "jpeg_decode.npz"))
    p = tmp_path / "map.jpg"
    p.write_bytes(data["rgb_420_q50/bytes"].tobytes())
    cfg = hmrm.Config().consume_string(f"heightmap {p}\ncolormap {p}\n")
    assert np.array_equal(cfg.height_rgb(), data["rgb_420_q50/req3"])
    assert np.array_equal(cfg.color_rgba(), data["rgb_420_q50/req4"])


def test_hdr_dimensions_narrow_like_the_reference_loader(hmrm, stb_ref):
    """stb narrows strtol's result to int BEFORE its size check (stb_image.h:7113-7120), so a Radiance
    header that says 2^32 + 5 columns is a 5-column picture there; and 2^31 + 5 is negative: refused.  Same here, live
    against the reference's own stb build where that exists."""
    def hdr(h_txt, w_txt, pixels):
        return (b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " + h_txt + b" +X " + w_txt + b"\n" + bytes(pixels))
    px = [128 + (7 * i) % 100 for i in range(4 * 5 * 2)]
    for h_txt, w_txt, ok in ((b"2", b"5", True), (b"2", str(2 ** 32 + 5).encode(), True), (str(2 ** 32 + 2).encode(), b"5", True),
                             (b"2", str(2 ** 31 + 5).encode(), False), (b"2", b"-5", False)):
        blob = hdr(h_txt, w_txt, px)
        for req in (0, 3, 4):
            try:
                got, n = hmrm.image_load_memory(blob, req)
            except hmrm.HmrmError:
                got = None
            assert (got is not None) == ok, (h_txt, w_txt, req)
            if ok:
                assert got.shape[:2] == (2, 5)
            if stb_ref is not None:
                exp, _ = stb_ref.load(blob, req)
                assert (exp is None) == (got is None), (h_txt, w_txt, req)
                if exp is not None:
                    assert np.array_equal(exp, got)


def test_decode_large_random_png_roundtrip(hmrm, stb_ref):
    """A 300x200 map-like image through zlib level 9 (dynamic Huffman, long matches)."""
    import struct
    import zlib
    rng = np.random.RandomState(5)
    img = np.cumsum(rng.randint(-3, 4, size=(200, 300, 3)), axis=1).astype(np.uint8)
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(200))

    def chunk(tag, d):
        return struct.pack(">I", len(d)) + tag + d + struct.pack(">I", zlib.crc32(tag + d) & 0xFFFFFFFF)

    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 300, 200, 8, 2, 0, 0, 0))
           + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))
    arr, n = hmrm.image_load_memory(png, 3)
    assert n == 3 and np.array_equal(arr, img)
    arr4, _ = hmrm.image_load_memory(png, 4)
    assert np.array_equal(arr4[:, :, :3], img) and (arr4[:, :, 3] == 255).all()
    if stb_ref is not None:
        exp, _ = stb_ref.load(png, 4)
        assert np.array_equal(arr4, exp)


def test_decode_errors(hmrm, tmp_path):
    with pytest.raises(hmrm.HmrmError) as e:
        hmrm.image_load(str(tmp_path / "missing.png"), 3)
    assert e.value.code == hmrm.HMRM_E_IO
    for blob in (b"", b"GIF89a....", b"\xff\xd8\xff\xe0" + b"\0" * 32, b"\x89PNG\r\n\x1a\n" + b"\0" * 40,
                 b"P6\n4 4\n255\n"[:5]):
        with pytest.raises(hmrm.HmrmError) as e:
            hmrm.image_load_memory(blob, 3)
        assert e.value.code == hmrm.HMRM_E_IMAGE
    # truncated IDAT stream
    data = _decode_fixture()
    blob = data["rgb8/bytes"].tobytes()
    with pytest.raises(hmrm.HmrmError):
        hmrm.image_load_memory(blob[: len(blob) // 2], 3)


def test_png_encode_matches_reference_stb_golden(hmrm):
    """SavePNG / stbi_write_png (hmap.cpp:157-160): byte-identical files."""
    import make_golden
    with open(os.path.join(GOLDEN, "png_encode.json")) as f:
        golden = json.load(f)
    imgs = make_golden.encode_test_images()
    assert set(imgs) == set(golden)
    for name, img in imgs.items():
        data = hmrm.png_encode(img)
        assert len(data) == golden[name]["length"], name
        assert hashlib.sha256(data).hexdigest() == golden[name]["sha256"], name


def test_png_encode_matches_reference_stb_live(hmrm, stb_ref):
    if stb_ref is None:
        pytest.skip("oracle/_ref not built; golden hashes cover it")
    rng = np.random.RandomState(3)
    for (h, w, c) in ((1, 1, 1), (2, 3, 3), (31, 17, 4), (64, 64, 4), (5, 300, 2), (200, 150, 3)):
        for kind in ("noise", "smooth", "flat"):
            if kind == "noise":
                img = rng.randint(0, 256, size=(h, w, c)).astype(np.uint8)
            elif kind == "smooth":
                img = (np.add.outer(np.arange(h) * 2, np.arange(w) * 3)[:, :, None] + np.arange(c) * 40).astype(np.uint8)
            else:
                img = np.full((h, w, c), 200, dtype=np.uint8)
            assert hmrm.png_encode(img) == stb_ref.write_png(img), (h, w, c, kind)
    # shapes that stress the match finder: > 32 KiB periods (window edge), buckets that overflow and drop
    # their older half, runs of maximal matches (258), a stream that falls back to stored blocks
    extra = {
        "long period": np.tile(rng.randint(0, 256, size=(1, 10000, 4)).astype(np.uint8), (6, 1, 1)),
        "period 32768": np.tile(rng.randint(0, 256, size=(1, 8192, 4)).astype(np.uint8), (4, 1, 1))[:, :8192],
        "zeros": np.zeros((300, 400, 4), np.uint8),
        "runs": np.repeat(rng.randint(0, 256, size=(120, 40, 3)).astype(np.uint8), 30, axis=1),
        "few symbols": (rng.randint(0, 3, size=(257, 501, 2)) * 100).astype(np.uint8),
        "noisy ramp": ((np.arange(640)[None, :, None] // 3 + np.arange(360)[:, None, None] // 2
                        + rng.randint(0, 16, size=(360, 640, 4))) & 255).astype(np.uint8),
        "one row": rng.randint(0, 2, size=(1, 5000, 3)).astype(np.uint8),
        "one column": rng.randint(0, 256, size=(3000, 1, 1)).astype(np.uint8),
    }
    for name, img in extra.items():
        assert hmrm.png_encode(np.ascontiguousarray(img)) == stb_ref.write_png(np.ascontiguousarray(img)), name


def test_png_roundtrip_and_files(hmrm, tmp_path):
    rng = np.random.RandomState(8)
    img = rng.randint(0, 256, size=(40, 50, 4)).astype(np.uint8)
    p = str(tmp_path / "a.png")
    hmrm.write_png(p, img)
    back, n = hmrm.image_load(p, 4)
    assert n == 4 and np.array_equal(back, img)
    q = str(tmp_path / "a.ppm")
    hmrm.write_ppm(q, img)
    back3, n3 = hmrm.image_load(q, 3)
    assert n3 == 3 and np.array_equal(back3, img[:, :, :3])
    with pytest.raises(hmrm.HmrmError) as e:
        hmrm.write_png(str(tmp_path / "no_such_dir" / "x.png"), img)
    assert e.value.code == hmrm.HMRM_E_IO
