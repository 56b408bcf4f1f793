#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.

  frames.npz      per test scene (tests/scenes.py): RGBA frame, per-ray step count,
                  bits of distance().  Produced by the C ORACLE (oracle/), because
                  the reference's render path can neither be built nor run here
                  (SDL2/glm absent) and ships no golden data of its own.
  png_decode.npz  small PNG/PNM files (bytes) and, for req_comp 0..4, the pixels
                  the REFERENCE's own stb_image v2.27 decodes from them
                  (oracle/_ref/libstb_ref.so, built from /root/reference/vendor).
  jpeg_decode.npz / bmp_tga_decode.npz  the same for JPEG, BMP and TGA files
  png_encode.json sha256 + length of the files the REFERENCE's stb_image_write
                  v1.16 writes for seeded test images.

Run from the repo root in the build container:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scenes  # noqa: E402
import stb_ref  # noqa: E402
from oracle import oracle_py as oracle  # noqa: E402


# ----------------------------------------------------------------- frames ----
def make_frames():
    out = {}
    for case in scenes.cases():
        name, rgb, cmap, params, cam = scenes.build_case(case)
        heights = oracle.update_heightmap(rgb, params)
        cfg = oracle.make_cfg(cam, params, rgb.shape[1], rgb.shape[0])
        fb, total, capped, steps, entry = oracle.render(cfg, heights, cmap, per_pixel=True)
        assert capped == 0
        out[name + "/frame"] = fb
        out[name + "/steps"] = steps.astype(np.uint32)
        out[name + "/entry_bits"] = entry.view(np.uint64)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)
    print("frames.npz:", len(out) // 3, "scenes")


# ------------------------------------------------------------ PNG fixtures ----
def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(rows, bpp, rng):
    """rows: list of bytes objects (packed scanlines); picks a random filter per row."""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for row in rows:
        ft = int(rng.randint(0, 5))
        out.append(ft)
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft]
            out.append((v - pred) & 255)
        prev = row
    return bytes(out)


def _pack_rows(samples, depth):
    """samples: H x (W*channels) integer array -> list of packed scanlines."""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(int(v) for v in r))
        elif depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in r))
        else:
            per = 8 // depth
            buf = bytearray((len(r) + per - 1) // per)
            for i, v in enumerate(r):
                buf[i // per] |= (int(v) & ((1 << depth) - 1)) << (8 - depth * (i % per + 1))
            rows.append(bytes(buf))
    return rows


def make_png(w, h, color, depth, rng, interlace=False, palette=None, trns=None, level=6):
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    maxv = (1 << depth) - 1 if color != 3 else (len(palette) // 3 - 1)
    samples = rng.randint(0, maxv + 1, size=(h, w, channels))
    bpp = max(1, channels * depth // 8)
    raw = b""
    if not interlace:
        raw = _filter_rows(_pack_rows(samples.reshape(h, w * channels), depth), bpp, rng)
    else:
        for xo, yo, xs, ys in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[yo::ys, xo::xs]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            raw += _filter_rows(_pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, rng)
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, int(interlace)))
    if palette is not None:
        png += _chunk(b"PLTE", bytes(palette))
    if trns is not None:
        png += _chunk(b"tRNS", bytes(trns))
    png += _chunk(b"tEXt", b"Comment\x00fixture")  # ancillary chunk must be skipped
    z = zlib.compress(raw, level)
    half = len(z) // 2
    png += _chunk(b"IDAT", z[:half]) + _chunk(b"IDAT", z[half:])  # split IDAT
    png += _chunk(b"IEND", b"")
    return png


def png_fixture_files():
    rng = np.random.RandomState(1234)
    files = {}
    for depth in (1, 2, 4, 8, 16):
        files[f"grey{depth}"] = make_png(13, 7, 0, depth, rng)
        files[f"grey{depth}_i"] = make_png(13, 7, 0, depth, rng, interlace=True)
    for depth in (8, 16):
        files[f"rgb{depth}"] = make_png(11, 9, 2, depth, rng)
        files[f"rgb{depth}_i"] = make_png(11, 9, 2, depth, rng, interlace=True)
        files[f"ga{depth}"] = make_png(9, 5, 4, depth, rng)
        files[f"rgba{depth}"] = make_png(10, 6, 6, depth, rng)
        files[f"rgba{depth}_i"] = make_png(10, 6, 6, depth, rng, interlace=True)
    pal = list(rng.randint(0, 256, size=16 * 3))
    for depth in (1, 2, 4, 8):
        n = min(16, 1 << depth)
        files[f"pal{depth}"] = make_png(12, 8, 3, depth, rng, palette=pal[:n * 3])
        files[f"pal{depth}_trns"] = make_png(12, 8, 3, depth, rng, palette=pal[:n * 3],
                                             trns=list(rng.randint(0, 256, size=max(1, n // 2))))
    files["pal4_i"] = make_png(12, 8, 3, 4, rng, interlace=True, palette=pal)
    # colour-key transparency on grey / RGB
    files["grey8_key"] = make_png(16, 4, 0, 8, rng, trns=[0, 5])
    files["grey2_key"] = make_png(16, 4, 0, 2, rng, trns=[0, 2])
    files["grey16_key"] = make_png(6, 6, 0, 16, rng, trns=[0x12, 0x34])
    files["rgb8_key"] = make_png(5, 5, 2, 8, rng, trns=[0, 1, 0, 2, 0, 3])
    files["one_pixel"] = make_png(1, 1, 2, 8, rng)
    files["wide_stored"] = make_png(70, 3, 6, 8, rng, level=0)  # stored (BTYPE 0) deflate blocks
    # PNM
    files["p6"] = b"P6\n# comment\n7 5\n255\n" + bytes(rng.randint(0, 256, size=7 * 5 * 3).astype(np.uint8))
    files["p5"] = b"P5 6 4 255\n" + bytes(rng.randint(0, 256, size=6 * 4).astype(np.uint8))
    files["p6_16"] = b"P6\n3 2\n65535\n" + bytes(rng.randint(0, 256, size=3 * 2 * 3 * 2).astype(np.uint8))
    return files


def make_png_decode(ref):
    files = png_fixture_files()
    out = {}
    for name, data in files.items():
        out[name + "/bytes"] = np.frombuffer(data, dtype=np.uint8)
        for req in range(5):
            if name == "p6_16" and req not in (0, 3):
                continue  # stb v2.27 mangles 16-bit PNM under channel conversion; not offered
            arr, n = ref.load(data, req)
            assert arr is not None, (name, req, n)
            out[f"{name}/req{req}"] = arr
            out[f"{name}/n{req}"] = np.array([n], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "png_decode.npz"), **out)
    print("png_decode.npz:", len(files), "files")


def encode_test_images():
    rng = np.random.RandomState(99)
    imgs = {}
    imgs["noise_rgba"] = rng.randint(0, 256, size=(37, 53, 4)).astype(np.uint8)
    g = np.add.outer(np.arange(64), np.arange(96)).astype(np.uint8)
    imgs["gradient_rgb"] = np.stack([g, g // 2, 255 - g], axis=2)
    imgs["flat_rgba"] = np.full((20, 30, 4), 77, dtype=np.uint8)
    imgs["grey1"] = rng.randint(0, 4, size=(25, 40, 1)).astype(np.uint8) * 60
    imgs["ga2"] = rng.randint(0, 256, size=(9, 9, 2)).astype(np.uint8)
    imgs["tiny"] = rng.randint(0, 256, size=(1, 1, 4)).astype(np.uint8)
    big = np.zeros((120, 200, 4), dtype=np.uint8)
    big[:, :, 0] = (np.arange(200)[None, :] * 3) & 255
    big[:, :, 1] = (np.arange(120)[:, None] * 5) & 255
    big[40:80, 50:150, 2] = rng.randint(0, 256, size=(40, 100))
    big[:, :, 3] = 255
    imgs["frame_like"] = big
    # a real frame from the oracle
    case = scenes.cases()[0]
    name, rgb, cmap, params, cam = scenes.build_case(case)
    heights = oracle.update_heightmap(rgb, params)
    fb, *_ = oracle.render(oracle.make_cfg(cam, params, rgb.shape[1], rgb.shape[0]), heights, cmap)
    imgs["oracle_frame"] = fb
    # match-finder stress: a period longer than the 32 KiB window, overflowing hash buckets, maximal runs
    r2 = np.random.RandomState(7)
    imgs["long_period"] = np.tile(r2.randint(0, 256, size=(1, 10000, 4)).astype(np.uint8), (6, 1, 1))
    imgs["runs"] = np.repeat(r2.randint(0, 256, size=(120, 40, 3)).astype(np.uint8), 30, axis=1)
    imgs["noisy_ramp"] = ((np.arange(640)[None, :, None] // 3 + np.arange(360)[:, None, None] // 2
                           + r2.randint(0, 16, size=(360, 640, 4))) & 255).astype(np.uint8)
    imgs["zeros"] = np.zeros((300, 400, 4), np.uint8)
    return imgs


def make_png_encode(ref):
    out = {}
    for name, img in encode_test_images().items():
        data = ref.write_png(img)
        out[name] = {"sha256": hashlib.sha256(data).hexdigest(), "length": len(data)}
    with open(os.path.join(HERE, "png_encode.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("png_encode.json:", len(out), "images")


def jpeg_fixture_files():
    """Small JPEGs written with Pillow (only needed when regenerating): baseline and
    progressive, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, grey, CMYK (Adobe APP14), RGB without
    colour transform, restart intervals, optimised Huffman tables, odd sizes."""
    import io
    from PIL import Image
    rng = np.random.RandomState(4321)

    def picture(h, w, c):
        base = rng.randint(0, 256, size=(h // 8 + 2, w // 8 + 2, c)).astype(np.float64)
        ys, xs = np.arange(h) / 8.0, np.arange(w) / 8.0
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
        v = (base[np.ix_(y0, x0)] * (1 - fy) * (1 - fx) + base[np.ix_(y0 + 1, x0)] * fy * (1 - fx)
             + base[np.ix_(y0, x0 + 1)] * (1 - fy) * fx + base[np.ix_(y0 + 1, x0 + 1)] * fy * fx)
        return np.clip(v + rng.randint(-24, 25, size=v.shape), 0, 255).astype(np.uint8)

    def save(arr, mode, **kw):
        buf = io.BytesIO()
        Image.fromarray(arr if arr.shape[2] > 1 else arr[:, :, 0], mode).save(buf, "JPEG", **kw)
        return buf.getvalue()

    files = {}
    rgb = picture(37, 29, 3)
    files["rgb_444_q90"] = save(rgb, "RGB", quality=90, subsampling=0)
    files["rgb_422_q75"] = save(rgb, "RGB", quality=75, subsampling=1)
    files["rgb_420_q50"] = save(rgb, "RGB", quality=50, subsampling=2)
    files["rgb_411_q100"] = save(rgb, "RGB", quality=100, subsampling="4:1:1")
    files["rgb_420_prog"] = save(rgb, "RGB", quality=85, subsampling=2, progressive=True)
    files["rgb_444_prog_opt"] = save(rgb, "RGB", quality=95, subsampling=0, progressive=True, optimize=True)
    files["rgb_420_restart"] = save(picture(45, 50, 3), "RGB", quality=80, subsampling=2, restart_marker_blocks=2)
    files["rgb_422_restart1"] = save(picture(24, 40, 3), "RGB", quality=92, subsampling=1, restart_marker_blocks=1)
    files["rgb_keep_rgb"] = save(rgb, "RGB", quality=90, keep_rgb=True)
    files["rgb_q10"] = save(picture(33, 31, 3), "RGB", quality=10, subsampling=2)
    files["grey_q80"] = save(picture(31, 22, 1), "L", quality=80)
    files["grey_prog"] = save(picture(17, 16, 1), "L", quality=60, progressive=True)
    files["cmyk_q85"] = save(picture(19, 23, 4), "CMYK", quality=85)
    files["cmyk_prog"] = save(picture(16, 16, 4), "CMYK", quality=70, progressive=True)
    files["one_pixel"] = save(picture(1, 1, 3), "RGB", quality=90)
    files["one_block"] = save(picture(8, 8, 3), "RGB", quality=90, subsampling=2)
    files["wide"] = save(picture(9, 70, 3), "RGB", quality=88, subsampling=2)
    return files


def bmp_tga_fixture_files():
    """BMP and TGA variants: written with Pillow where it can, hand-assembled otherwise
    (16-bit 5-5-5 / 5-6-5, 32-bit default / all-zero alpha / shuffled 8-8-8 bitfields, top-down,
    OS/2 header, 4-bit palette; TGA 15/16-bit, 16-bit palette entries, RLE, top-left origin)."""
    import io
    from PIL import Image
    rng = np.random.RandomState(777)

    def pil(arr, mode, fmt, **kw):
        buf = io.BytesIO()
        Image.fromarray(arr, mode).save(buf, fmt, **kw)
        return buf.getvalue()

    files = {}
    rgb = rng.randint(0, 256, size=(13, 17, 3)).astype(np.uint8)
    rgba = rng.randint(0, 256, size=(9, 11, 4)).astype(np.uint8)
    grey = rng.randint(0, 256, size=(10, 7)).astype(np.uint8)
    files["bmp24"] = pil(rgb, "RGB", "BMP")
    files["bmp32"] = pil(rgba, "RGBA", "BMP")
    files["bmp8"] = pil(grey, "L", "BMP")
    quant = Image.fromarray(rgb, "RGB").quantize(16)
    b = io.BytesIO(); quant.save(b, "BMP"); files["bmp_pal"] = b.getvalue()
    b = io.BytesIO(); Image.fromarray((grey > 128).astype(np.uint8) * 255, "L").convert("1").save(b, "BMP")
    files["bmp1"] = b.getvalue()

    def bmp_raw(w, h, bpp, rows, masks=None, hsz=40, topdown=False, palette=b"", compress=0):
        data = b"".join(r + b"\0" * ((-len(r)) & 3) for r in rows)
        extra = b"".join(struct.pack("<I", m) for m in masks) if masks else b""
        if hsz == 12:
            hdr = struct.pack("<IHHHH", 12, w, h, 1, bpp)
        else:
            hdr = struct.pack("<IiiHHIIiiII", 40, w, -h if topdown else h, 1, bpp, compress, len(data), 2835, 2835, 0, 0)
        off = 14 + len(hdr) + len(extra) + len(palette)
        return b"BM" + struct.pack("<IHHI", off + len(data), 0, 0, off) + hdr + extra + palette + data

    w, h = 7, 5
    px16 = rng.randint(0, 65536, size=(h, w)).astype(np.uint16)
    px32 = rng.randint(0, 2 ** 32, size=(h, w), dtype=np.uint64).astype(np.uint32)
    rows24 = lambda: [rng.randint(0, 256, size=w * 3).astype(np.uint8).tobytes() for _ in range(h)]
    files["bmp16_555"] = bmp_raw(w, h, 16, [px16[r].tobytes() for r in range(h)])
    files["bmp16_565"] = bmp_raw(w, h, 16, [px16[r].tobytes() for r in range(h)], masks=(0xF800, 0x07E0, 0x001F), compress=3)
    files["bmp32_default"] = bmp_raw(w, h, 32, [px32[r].tobytes() for r in range(h)])
    files["bmp32_alpha0"] = bmp_raw(w, h, 32, [(px32[r] & 0x00FFFFFF).tobytes() for r in range(h)])
    files["bmp32_bitfields"] = bmp_raw(w, h, 32, [px32[r].tobytes() for r in range(h)],
                                       masks=(0x0000FF00, 0x00FF0000, 0xFF000000), compress=3)
    files["bmp24_topdown"] = bmp_raw(w, h, 24, rows24(), topdown=True)
    files["bmp24_os2"] = bmp_raw(w, h, 24, rows24(), hsz=12)
    files["bmp4"] = bmp_raw(w, h, 4, [rng.randint(0, 256, size=(w + 1) // 2).astype(np.uint8).tobytes() for _ in range(h)],
                            palette=bytes(rng.randint(0, 256, size=16 * 4).astype(np.uint8)))
    files["tga24"] = pil(rgb, "RGB", "TGA")
    files["tga32"] = pil(rgba, "RGBA", "TGA")
    files["tga8"] = pil(grey, "L", "TGA")
    files["tga24_rle"] = pil(np.repeat(rgb[:, :6], 3, axis=1).copy(), "RGB", "TGA", compression="tga_rle")
    files["tga32_rle_top"] = pil(rgba, "RGBA", "TGA", compression="tga_rle", orientation=1)
    b = io.BytesIO(); quant.save(b, "TGA"); files["tga_pal"] = b.getvalue()
    files["tga_la"] = pil(rng.randint(0, 256, size=(6, 5, 2)).astype(np.uint8), "LA", "TGA")

    def tga_raw(w, h, bpp, data, imgtype=2, desc=0, pal=b"", palbits=0):
        entries = len(pal) // (max(palbits, 8) // 8) if pal else 0
        return struct.pack("<BBBHHBHHHHBB", 0, 1 if pal else 0, imgtype, 0, entries, palbits, 0, 0, w, h, bpp, desc) + pal + data

    files["tga16"] = tga_raw(w, h, 16, px16.tobytes())
    files["tga15_top"] = tga_raw(w, h, 15, px16.tobytes(), desc=0x20)
    files["tga_pal16"] = tga_raw(w, h, 8, rng.randint(0, 8, size=w * h).astype(np.uint8).tobytes(), imgtype=1,
                                 pal=rng.randint(0, 65536, size=8).astype(np.uint16).tobytes(), palbits=16)
    return files


def make_bmp_tga_decode(ref):
    files = bmp_tga_fixture_files()
    out = {}
    for name, data in files.items():
        out[name + "/bytes"] = np.frombuffer(data, dtype=np.uint8)
        for req in range(5):
            arr, n = ref.load(data, req)
            assert arr is not None, (name, req, n)
            out[f"{name}/req{req}"] = arr
            out[f"{name}/n{req}"] = np.array([n], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "bmp_tga_decode.npz"), **out)
    print("bmp_tga_decode.npz:", len(files), "files")


def make_jpeg_decode(ref):
    files = jpeg_fixture_files()
    out = {}
    for name, data in files.items():
        out[name + "/bytes"] = np.frombuffer(data, dtype=np.uint8)
        for req in range(5):
            arr, n = ref.load(data, req)
            assert arr is not None, (name, req, n)
            out[f"{name}/req{req}"] = arr
            out[f"{name}/n{req}"] = np.array([n], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "jpeg_decode.npz"), **out)
    print("jpeg_decode.npz:", len(files), "files")



if __name__ == "__main__":
    make_frames()
    ref = stb_ref.load()
    if ref is None:
        print("oracle/_ref/libstb_ref.so missing: run `make -C oracle` where /root/reference exists")
        sys.exit(1)
    make_png_decode(ref)
    make_png_encode(ref)
    make_jpeg_decode(ref)
    make_bmp_tga_decode(ref)
