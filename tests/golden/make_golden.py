#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.

  frames.npz      per test scene (tests/scenes.py): RGBA frame, per-ray step count,
                  bits of distance().  Produced by the C ORACLE (oracle/), because
                  the reference's render path can neither be built nor run here
                  (SDL2/glm absent) and ships no golden data of its own.
  png_decode.npz  small PNG/PNM files (bytes) and, for req_comp 0..4, the pixels
                  the REFERENCE's own stb_image v2.27 decodes from them
                  (oracle/_ref/libstb_ref.so, built from /root/reference/vendor).
  jpeg_decode.npz / bmp_tga_decode.npz / legacy_decode.npz  the same for JPEG, BMP, TGA, GIF, PSD, PIC and HDR files
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



# ---------------------------------------------------- GIF / PSD / PIC / HDR ----
def _gif_lzw(indices, min_bits, with_clear=True, early_end=None):
    """GIF's variable-width LZW encoder (clear code first, dictionary reset at 4096), sub-block packed."""
    clear, end = 1 << min_bits, (1 << min_bits) + 1
    table = {bytes([i]): i for i in range(clear)}
    nxt, width = end + 1, min_bits + 1
    codes = [(clear, width)] if with_clear else []
    cur = b""
    for k, i in enumerate(indices):
        if early_end is not None and k == early_end:
            break
        c = cur + bytes([i])
        if c in table:
            cur = c
            continue
        codes.append((table[cur], width))
        if nxt < 4096:
            table[c] = nxt
            nxt += 1
            if nxt - 1 == (1 << width) and width < 12:
                width += 1
        else:
            codes.append((clear, width))
            table = {bytes([j]): j for j in range(clear)}
            nxt, width = end + 1, min_bits + 1
        cur = bytes([i])
    if cur:
        codes.append((table[cur], width))
    codes.append((end, width))
    acc = nbits = 0
    raw = bytearray()
    for code, w in codes:
        acc |= code << nbits
        nbits += w
        while nbits >= 8:
            raw.append(acc & 255)
            acc >>= 8
            nbits -= 8
    if nbits:
        raw.append(acc & 255)
    out = bytearray([min_bits])
    for i in range(0, len(raw), 255):
        chunk = raw[i:i + 255]
        out.append(len(chunk))
        out += chunk
    out.append(0)
    return bytes(out)


def _gif(w, h, frames, version=b"89a", global_pal=None, bg=0, trailer=True):
    """frames: list of dicts(rect=(x,y,w,h), indices, interlace, local_pal, transparent, min_bits, pre=bytes)."""
    flags = 0
    body = b""
    if global_pal is not None:
        bits = max(1, (len(global_pal) // 3 - 1).bit_length())
        flags = 0x80 | (bits - 1) | ((bits - 1) << 4)
        global_pal = global_pal + b"\0" * (3 * (1 << bits) - len(global_pal))
    out = b"GIF" + version + struct.pack("<HHBBB", w, h, flags, bg, 0) + (global_pal or b"")
    for f in frames:
        out += f.get("pre", b"")
        if f.get("transparent") is not None or f.get("gce"):
            t = f.get("transparent")
            out += b"\x21\xF9\x04" + bytes([(f.get("dispose", 0) << 2) | (1 if t is not None else 0)]) + struct.pack("<H", 7) + bytes([t or 0]) + b"\0"
        x, y, fw, fh = f["rect"]
        lflags = 0x40 if f.get("interlace") else 0
        lp = f.get("local_pal")
        if lp is not None:
            bits = max(1, (len(lp) // 3 - 1).bit_length())
            lflags |= 0x80 | (bits - 1)
            lp = lp + b"\0" * (3 * (1 << bits) - len(lp))
        out += b"\x2C" + struct.pack("<HHHHB", x, y, fw, fh, lflags) + (lp or b"")
        idx = list(f["indices"])
        if f.get("interlace"):
            rows = [idx[r * fw:(r + 1) * fw] for r in range(fh)]
            order = [r for s0, st in ((0, 8), (4, 8), (2, 4), (1, 2)) for r in range(s0, fh, st)]
            idx = [v for r in order for v in rows[r]]
        out += _gif_lzw(idx, f.get("min_bits", 8), early_end=f.get("early_end"))
    return out + (b"\x3B" if trailer else b"")


def _packbits(row):
    out, i = bytearray(), 0
    while i < len(row):
        j = i
        while j + 1 < len(row) and row[j + 1] == row[i] and j - i < 127:
            j += 1
        if j > i:
            out += bytes([257 - (j - i + 1), row[i]])
            i = j + 1
            continue
        j = i
        while j + 1 < len(row) and (j + 2 >= len(row) or row[j + 1] != row[j + 2]) and j - i < 127:
            j += 1
        out += bytes([j - i]) + bytes(row[i:j + 1])
        i = j + 1
    return bytes(out)


def _psd(planes, depth=8, rle=False, mode=3, resources=b"", extra_noop=False):
    """planes: list of HxW arrays (uint8 or uint16) in channel order R,G,B,A,..."""
    h, w = planes[0].shape
    hdr = b"8BPS" + struct.pack(">H6xHIIHH", 1, len(planes), h, w, depth, mode)
    hdr += struct.pack(">I", 0) + struct.pack(">I", len(resources)) + resources + struct.pack(">I", 0)
    if not rle:
        data = b"".join((p.astype(">u2") if depth == 16 else p.astype(np.uint8)).tobytes() for p in planes)
        return hdr + struct.pack(">H", 0) + data
    rows = [_packbits(bytes(p[r].astype(np.uint8))) for p in planes for r in range(h)]
    if extra_noop:
        rows = [b"\x80" + r for r in rows]
    return hdr + struct.pack(">H", 1) + b"".join(struct.pack(">H", len(r)) for r in rows) + b"".join(rows)


def _pic(w, h, packets, rows_of):
    """packets: list of (type, channel mask); rows_of(y, packet index) -> encoded bytes of that packet's row."""
    hdr = b"\x53\x80\xF6\x34" + struct.pack(">f", 3.7) + b"fixture".ljust(80, b"\0") + b"PICT" + struct.pack(">HHfHH", w, h, 1.0, 3, 0)
    for k, (t, mask) in enumerate(packets):
        hdr += bytes([1 if k + 1 < len(packets) else 0, 8, t, mask])
    return hdr + b"".join(rows_of(y, k) for y in range(h) for k in range(len(packets)))


def _rgbe(rgb):
    """float HxWx3 -> uint8 HxWx4 (Ward's RGBE)."""
    m = rgb.max(axis=2)
    out = np.zeros(rgb.shape[:2] + (4,), dtype=np.uint8)
    mant, exp = np.frexp(m)
    scale = np.where(m > 1e-32, mant * 256.0 / np.maximum(m, 1e-38), 0.0)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, exp + 128, 0).astype(np.uint8)
    return out


def _hdr(rgbe, rle=True, magic=b"#?RADIANCE", extra=b"EXPOSURE=1.0\n", raw_run_mix=True):
    h, w = rgbe.shape[:2]
    out = magic + b"\n" + extra + b"FORMAT=32-bit_rle_rgbe\n\n" + b"-Y %d +X %d\n" % (h, w)
    if not rle:
        return out + rgbe.tobytes()
    for y in range(h):
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            row, i = rgbe[y, :, c], 0
            while i < w:
                j = i
                while j + 1 < w and row[j + 1] == row[i] and j - i < 126:
                    j += 1
                if j - i >= 2 or not raw_run_mix:
                    out += bytes([128 + (j - i + 1), row[i]])
                    i = j + 1
                else:
                    j = i
                    while j + 1 < w and j - i < 127 and not (j + 2 < w and row[j + 1] == row[j + 2]):
                        j += 1
                    out += bytes([j - i + 1]) + bytes(row[i:j + 1])
                    i = j + 1
    return out


def legacy_fixture_files():
    """GIF, PSD, PIC and Radiance HDR variants, all hand-assembled (the encoders above are this script's own)."""
    rng = np.random.RandomState(4321)
    files = {}
    pal8 = bytes(rng.randint(0, 256, size=3 * 256).astype(np.uint8))
    pal4 = bytes(rng.randint(0, 256, size=3 * 16).astype(np.uint8))
    w, h = 23, 17
    smooth = (np.add.outer(np.arange(h) * 7, np.arange(w) * 5) % 256).astype(np.uint8)
    noisy = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
    files["gif87_global"] = _gif(w, h, [dict(rect=(0, 0, w, h), indices=noisy.ravel())], version=b"87a", global_pal=pal8)
    files["gif89_transparent"] = _gif(w, h, [dict(rect=(0, 0, w, h), indices=smooth.ravel() % 16, transparent=5, min_bits=4)], global_pal=pal4)
    files["gif_interlaced"] = _gif(w, h, [dict(rect=(0, 0, w, h), indices=noisy.ravel(), interlace=True)], global_pal=pal8)
    files["gif_local_palette"] = _gif(w, h, [dict(rect=(0, 0, w, h), indices=noisy.ravel() % 16, local_pal=pal4, min_bits=4)])
    files["gif_local_transparent"] = _gif(w, h, [dict(rect=(0, 0, w, h), indices=noisy.ravel() % 16, local_pal=pal4, min_bits=4, transparent=3)], global_pal=pal8)
    files["gif_subrect_bg"] = _gif(w, h, [dict(rect=(4, 3, 11, 9), indices=noisy[:9, :11].ravel())], global_pal=pal8, bg=77)
    files["gif_subrect_bg0"] = _gif(w, h, [dict(rect=(4, 3, 11, 9), indices=noisy[:9, :11].ravel())], global_pal=pal8, bg=0)
    files["gif_subrect_transparent_bg"] = _gif(w, h, [dict(rect=(2, 1, 15, 12), indices=noisy[:12, :15].ravel() % 16, transparent=9, min_bits=4, interlace=True)],
                                               global_pal=pal4, bg=9)
    files["gif_two_frames"] = _gif(w, h, [dict(rect=(0, 0, w, h), indices=smooth.ravel()), dict(rect=(1, 1, 5, 5), indices=noisy[:5, :5].ravel(), gce=True, dispose=2)],
                                   global_pal=pal8)
    files["gif_extensions_first"] = _gif(w, h, [dict(rect=(0, 0, w, h), indices=noisy.ravel(), pre=b"\x21\xFE\x05hello\x00" + b"\x21\xFF\x0bNETSCAPE2.0\x03\x01\x00\x00\x00")],
                                         global_pal=pal8)
    big = rng.randint(0, 256, size=(70, 90)).astype(np.uint8)   # > 4096 dictionary entries: the table is reset mid-stream
    files["gif_dictionary_reset"] = _gif(90, 70, [dict(rect=(0, 0, 90, 70), indices=big.ravel())], global_pal=pal8)
    files["gif_2colour"] = _gif(9, 5, [dict(rect=(0, 0, 9, 5), indices=(noisy[:5, :9] & 1).ravel(), min_bits=2)], global_pal=pal8[:6])
    files["gif_runs"] = _gif(64, 8, [dict(rect=(0, 0, 64, 8), indices=np.repeat(np.arange(8, dtype=np.uint8), 64))], global_pal=pal8)  # KwKwK codes
    files["gif_short_data"] = _gif(w, h, [dict(rect=(0, 0, w, h), indices=noisy.ravel(), early_end=200)], global_pal=pal8, bg=3)
    files["gif_truncated"] = files["gif87_global"][:1000]          # ends inside the raster data: decodes as far as it goes
    files["gif_truncated_palette"] = files["gif87_global"][:400]   # ends inside the colour table: refused
    files["gif_interlaced_tiny"] = _gif(5, 3, [dict(rect=(0, 0, 5, 3), indices=noisy[:3, :5].ravel(), interlace=True)], global_pal=pal8)

    pw, ph = 13, 9
    planes = [rng.randint(0, 256, size=(ph, pw)).astype(np.uint8) for _ in range(5)]
    runs = [np.repeat(rng.randint(0, 256, size=(ph, 3)).astype(np.uint8), 5, axis=1)[:, :pw] for _ in range(4)]
    files["psd_rgb_raw"] = _psd(planes[:3])
    files["psd_rgba_raw"] = _psd(planes[:4])
    files["psd_rgb_rle"] = _psd(runs[:3], rle=True)
    files["psd_rgba_rle"] = _psd(runs[:4], rle=True, resources=b"8BIM" + bytes(12))
    files["psd_rgba_rle_noop"] = _psd(planes[:4], rle=True, extra_noop=True)
    files["psd_5_channels"] = _psd(planes[:5])
    files["psd_1_channel"] = _psd(planes[:1])
    files["psd_2_channels_rle"] = _psd(runs[:2], rle=True)
    files["psd_16bit_raw"] = _psd([rng.randint(0, 65536, size=(ph, pw)).astype(np.uint16) for _ in range(4)], depth=16)
    alpha_edge = planes[:3] + [np.tile(np.array([0, 1, 2, 127, 128, 200, 254, 255, 3, 77, 250, 9, 255], dtype=np.uint8), (ph, 1))]
    files["psd_alpha_matte"] = _psd(alpha_edge)
    files["psd_truncated"] = files["psd_rgba_raw"][:-60]

    qw, qh = 11, 7
    pic_rgb = rng.randint(0, 256, size=(qh, qw, 4)).astype(np.uint8)
    pic_runs = np.repeat(rng.randint(0, 256, size=(qh, 3, 4)).astype(np.uint8), 4, axis=1)[:, :qw]

    def raw_rows(img, mask):
        ch = [k for k in range(4) if mask & (0x80 >> k)]
        return lambda y, k: bytes(img[y][:, ch].ravel())

    def pure_rle_rows(img, mask):
        ch = [k for k in range(4) if mask & (0x80 >> k)]

        def f(y, k):
            out, x = bytearray(), 0
            while x < qw:
                j = x
                while j + 1 < qw and (img[y, j + 1, ch] == img[y, x, ch]).all() and j - x < 200:
                    j += 1
                out += bytes([j - x + 1]) + bytes(img[y, x, ch])
                x = j + 1
            return bytes(out)
        return f

    def mixed_rows(img, mask, long_run=False):
        ch = [k for k in range(4) if mask & (0x80 >> k)]

        def f(y, k):
            out, x = bytearray(), 0
            while x < qw:
                j = x
                while j + 1 < qw and (img[y, j + 1, ch] == img[y, x, ch]).all():
                    j += 1
                n = j - x + 1
                if n >= 2:
                    out += (bytes([128]) + struct.pack(">H", n) if long_run else bytes([127 + n])) + bytes(img[y, x, ch])
                    x = j + 1
                else:
                    j = x
                    while j + 1 < qw and not (j + 2 < qw and (img[y, j + 1, ch] == img[y, j + 2, ch]).all()):
                        j += 1
                    out += bytes([j - x]) + bytes(img[y, x:j + 1][:, ch].ravel())
                    x = j + 1
            return bytes(out)
        return f

    files["pic_rgb_raw"] = _pic(qw, qh, [(0, 0xE0)], raw_rows(pic_rgb, 0xE0))
    files["pic_rgba_raw"] = _pic(qw, qh, [(0, 0xF0)], raw_rows(pic_rgb, 0xF0))
    files["pic_rgb_pure_rle"] = _pic(qw, qh, [(1, 0xE0)], pure_rle_rows(pic_runs, 0xE0))
    files["pic_rgb_mixed"] = _pic(qw, qh, [(2, 0xE0)], mixed_rows(pic_runs, 0xE0))
    files["pic_rgb_mixed_long"] = _pic(qw, qh, [(2, 0xE0)], mixed_rows(pic_runs, 0xE0, long_run=True))
    rgbf, af = mixed_rows(pic_runs, 0xE0), raw_rows(pic_rgb, 0x10)
    files["pic_rgb_then_alpha"] = _pic(qw, qh, [(2, 0xE0), (0, 0x10)], lambda y, k: rgbf(y, k) if k == 0 else af(y, k))
    files["pic_red_only"] = _pic(qw, qh, [(0, 0x80)], raw_rows(pic_rgb, 0x80))
    files["pic_alpha_only"] = _pic(qw, qh, [(1, 0x10)], pure_rle_rows(pic_runs, 0x10))

    hw, hh = 19, 6
    lin = np.abs(rng.normal(size=(hh, hw, 3))) * np.array([0.02, 1.0, 30.0])[None, None, :]
    lin[0, :5] = 0.0
    lin[1, 3:12] = lin[1, 3]
    e = _rgbe(lin)
    files["hdr_rle"] = _hdr(e)
    files["hdr_rle_runs_only"] = _hdr(np.repeat(e[:, :4], 5, axis=1)[:, :hw].copy(), raw_run_mix=False)
    files["hdr_flat"] = _hdr(e, rle=False)
    files["hdr_rgbe_magic"] = _hdr(e, magic=b"#?RGBE", extra=b"")
    files["hdr_narrow_flat"] = _hdr(e[:, :7].copy(), rle=True)[: len(_hdr(e[:, :7].copy(), rle=False))] if False else _hdr(e[:, :7].copy(), rle=False)
    files["hdr_bright_dark"] = _hdr(_rgbe(np.array([[[1e-6, 1e-3, 0.5], [1.0, 2.0, 1e4], [0.0, 0.0, 0.0], [0.9999, 1.0001, 0.18], [1e-9, 3e5, 7.0],
                                                    [0.5, 0.5, 0.5], [0.25, 0.125, 0.0625], [123.0, 0.001, 1.0], [1e-12, 1e-12, 1e-12]]])), rle=True)
    mixed = bytearray(_hdr(e))
    files["hdr_second_row_not_rle"] = bytes(mixed[:mixed.index(b"+X %d\n" % hw) + len(b"+X %d\n" % hw)]) + _hdr(e)[len(_hdr(e[:0])) - 0:][:0] + b""  # placeholder, replaced below
    head = _hdr(e[:1])                       # header + the first row, run-length coded
    hdr_only = _hdr(e[:0]).replace(b"-Y 0", b"-Y %d" % hh)
    first_row = head[len(_hdr(e[:0])):]
    files["hdr_second_row_not_rle"] = hdr_only + first_row + e[1:].tobytes()
    # more corners: a 16-bit PSD with PackBits rows (taken as bytes), PSD / flat HDR cut short, an HDR too wide for
    # run-length scanlines, a header line longer than the reader's buffer, GIF code sizes 1 and 3, a GIF without trailer
    files["psd_16bit_rle"] = _psd([np.repeat(rng.randint(0, 256, size=(ph, 4)).astype(np.uint8), 4, axis=1)[:, :pw] for _ in range(3)], depth=16, rle=True)
    files["psd_rle_truncated"] = files["psd_rgba_rle"][:-25]
    # (a PIC cut short, a PIC that ends after its header and an HDR whose run-length data ends early are NOT here: the
    # reference's loader crashes on the first two and never returns from the third -- tests/test_image_io.py only checks
    # that this library refuses them)
    files["hdr_flat_truncated"] = files["hdr_flat"][:-30]
    files["hdr_long_header_line"] = _hdr(e, extra=b"COMMENT=" + b"x" * 1500 + b"\n")
    wide = np.tile(e[:1, :16], (1, 2050, 1))[:, :32770].copy()      # width >= 32768: flat data whatever the file says
    files["hdr_too_wide_for_rle"] = _hdr(wide, rle=False)
    files["gif_code_size_1"] = _gif(7, 3, [dict(rect=(0, 0, 7, 3), indices=(noisy[:3, :7] & 1).ravel(), min_bits=1)], global_pal=pal8[:6])
    files["gif_code_size_3"] = _gif(9, 9, [dict(rect=(0, 0, 9, 9), indices=(noisy[:9, :9] & 7).ravel(), min_bits=3, interlace=True)], global_pal=pal8[:24])
    files["gif_no_trailer"] = _gif(w, h, [dict(rect=(0, 0, w, h), indices=smooth.ravel())], global_pal=pal8, trailer=False)
    files["gif_zero_width_image"] = _gif(w, h, [dict(rect=(3, 3, 0, 4), indices=[])], global_pal=pal8, bg=5)
    return files


def make_legacy_decode(ref):
    files = legacy_fixture_files()
    out = {}
    for name, data in files.items():
        out[name + "/bytes"] = np.frombuffer(data, dtype=np.uint8)
        for req in range(5):
            arr, n = ref.load(data, req)
            if arr is None:   # (a file the reference refuses: recorded as such)
                out[f"{name}/n{req}"] = np.array([-1], dtype=np.int32)
                continue
            out[f"{name}/req{req}"] = arr
            out[f"{name}/n{req}"] = np.array([n], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "legacy_decode.npz"), **out)
    print("legacy_decode.npz:", len(files), "files")


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
    make_legacy_decode(ref)
