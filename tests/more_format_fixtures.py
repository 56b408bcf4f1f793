"""Hand-built GIF / PSD / PIC / HDR files for the decoder tests (writers are the tests' own; the
expected pixels come from the reference's stb build, tests/golden/more_formats.npz)."""
import io
import struct

import numpy as np


# ------------------------------------------------------------------ GIF ----
def _lzw_gif(indices, min_code_size, clear_every=None):
    """Plain GIF LZW encoder (variable code width, clear code first, optional periodic clears)."""
    clear, eoi = 1 << min_code_size, (1 << min_code_size) + 1
    out, acc, nbits = bytearray(), 0, 0

    def put(code, width):
        nonlocal acc, nbits
        acc |= code << nbits
        nbits += width
        while nbits >= 8:
            out.append(acc & 255)
            acc >>= 8
            nbits -= 8

    table = {(i,): i for i in range(clear)}
    nxt, width = eoi + 1, min_code_size + 1
    put(clear, width)
    cur = ()
    count = 0
    for sym in indices:
        sym = int(sym)
        if cur + (sym,) in table:
            cur = cur + (sym,)
            continue
        put(table[cur], width)
        count += 1
        if nxt < 4096:
            table[cur + (sym,)] = nxt
            nxt += 1
            if nxt > (1 << width) and width < 12:
                width += 1
        if (clear_every and count % clear_every == 0) or nxt >= 4096:
            put(clear, width)
            table = {(i,): i for i in range(clear)}
            nxt, width = eoi + 1, min_code_size + 1
        cur = (sym,)
    if cur:
        put(table[cur], width)
    put(eoi, width)
    if nbits:
        out.append(acc & 255)
    blocks = bytearray()
    for i in range(0, len(out), 255):
        chunk = out[i:i + 255]
        blocks.append(len(chunk))
        blocks += chunk
    blocks.append(0)
    return bytes(blocks)


def gif_bytes(w, h, palette, indices, *, version=b"89a", bgindex=0, transparent=None, rect=None, interlace=False,
              local_palette=None, clear_every=None, extra_ext=False, min_code_size=None):
    """One-frame GIF.  palette: (2^k, 3) uint8; indices: (rh, rw) palette indices of the frame rectangle."""
    pal = np.asarray(palette, dtype=np.uint8)
    k = max(1, int(np.ceil(np.log2(len(pal)))))
    assert len(pal) == 1 << k
    out = bytearray(b"GIF" + version)
    out += struct.pack("<HHBBB", w, h, 0x80 | (k - 1) | ((k - 1) << 4), bgindex, 0)
    out += pal.tobytes()
    if extra_ext:
        out += b"\x21\xfe\x05hello\x03abc\x00"  # comment extension with two sub-blocks
        out += b"\x21\xf9\x02\x07\x07"          # graphic control extension of the wrong length: stb skips the
                                                 # two bytes and goes straight back to reading a block tag
    if transparent is not None:
        out += b"\x21\xf9\x04" + struct.pack("<BHB", 0x01, 7, transparent) + b"\x00"
    x, y, rw, rh = rect if rect else (0, 0, w, h)
    idx = np.asarray(indices).reshape(rh, rw)
    flags = 0
    lp = None
    if local_palette is not None:
        lp = np.asarray(local_palette, dtype=np.uint8)
        lk = max(1, int(np.ceil(np.log2(len(lp)))))
        flags |= 0x80 | (lk - 1)
    if interlace:
        flags |= 0x40
        order = list(range(0, rh, 8)) + list(range(4, rh, 8)) + list(range(2, rh, 4)) + list(range(1, rh, 2))
        idx = idx[order]
    out += b"\x2c" + struct.pack("<HHHHB", x, y, rw, rh, flags)
    if lp is not None:
        out += lp.tobytes()
    ncol = len(lp) if lp is not None else len(pal)
    mcs = min_code_size or max(2, int(np.ceil(np.log2(ncol))))
    out += bytes([mcs]) + _lzw_gif(idx.ravel(), mcs, clear_every)
    out += b"\x3b"
    return bytes(out)


# ------------------------------------------------------------------ PSD ----
def _packbits(row):
    out, i, n = bytearray(), 0, len(row)
    while i < n:
        run = 1
        while i + run < n and run < 128 and row[i + run] == row[i]:
            run += 1
        if run >= 3:
            out += bytes([257 - run, row[i]])
            i += run
            continue
        j = i
        while j < n and j - i < 128:
            if j + 2 < n and row[j] == row[j + 1] == row[j + 2]:
                break
            j += 1
        out += bytes([j - i - 1]) + bytes(row[i:j])
        i = j
    return bytes(out)


def psd_bytes(planes, *, depth=8, rle=False, noop=False):
    """planes: (channels, h, w) uint8 or uint16 (depth 16)."""
    planes = np.asarray(planes)
    c, h, w = planes.shape
    out = bytearray(b"8BPS" + struct.pack(">H6xHIIHH", 1, c, h, w, depth, 3))
    out += struct.pack(">I", 0) + struct.pack(">I", 4) + b"\x01\x02\x03\x04" + struct.pack(">I", 0)
    out += struct.pack(">H", 1 if rle else 0)
    if rle:
        assert depth == 8
        rows = [_packbits(planes[ch, y].tobytes()) for ch in range(c) for y in range(h)]
        if noop:
            rows = [b"\x80" + r for r in rows]
        out += b"".join(struct.pack(">H", len(r)) for r in rows) + b"".join(rows)
    elif depth == 16:
        out += planes.astype(">u2").tobytes()
    else:
        out += planes.astype(np.uint8).tobytes()
    return bytes(out)


# ------------------------------------------------------------------ PIC ----
def pic_bytes(img, packets):
    """img: (h, w, 4) uint8 RGBA; packets: list of (type, channel_mask) with type 0 raw, 1 pure RLE, 2 mixed RLE;
    channel mask bits 0x80 R, 0x40 G, 0x20 B, 0x10 A."""
    img = np.asarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    out = bytearray(b"\x53\x80\xF6\x34" + b"\x00" * 84 + b"PICT")
    out += struct.pack(">HHIHH", w, h, 0x3F800000, 3, 0)
    for k, (typ, mask) in enumerate(packets):
        out += bytes([1 if k + 1 < len(packets) else 0, 8, typ, mask])
    for y in range(h):
        for typ, mask in packets:
            chans = [i for i in range(4) if mask & (0x80 >> i)]
            px = [bytes(img[y, x, chans]) for x in range(w)]
            if typ == 0:
                out += b"".join(px)
            elif typ == 1:
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 255 and px[x + run] == px[x]:
                        run += 1
                    out += bytes([run]) + px[x]
                    x += run
            else:
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 300 and px[x + run] == px[x]:
                        run += 1
                    if run >= 2:
                        if run > 128:
                            out += bytes([128]) + struct.pack(">H", run) + px[x]
                        else:
                            out += bytes([run + 127]) + px[x]
                        x += run
                    else:
                        j = x
                        while j < w and j - x < 128 and not (j + 1 < w and px[j] == px[j + 1]):
                            j += 1
                        j = max(j, x + 1)
                        out += bytes([j - x - 1]) + b"".join(px[x:j])
                        x = j
    return bytes(out)


# ------------------------------------------------------------------ HDR ----
def hdr_bytes(rgbe, *, magic=b"#?RADIANCE", rle=True, extra_header=True):
    """rgbe: (h, w, 4) uint8."""
    rgbe = np.asarray(rgbe, dtype=np.uint8)
    h, w, _ = rgbe.shape
    out = bytearray(magic + b"\n")
    if extra_header:
        out += b"# made by the tests\nEXPOSURE=1.0\n"
    out += b"FORMAT=32-bit_rle_rgbe\n\n" + ("-Y %d +X %d\n" % (h, w)).encode()
    if not rle or w < 8 or w >= 32768:
        out += rgbe.tobytes()
        return bytes(out)
    for y in range(h):
        out += bytes([2, 2, w >> 8, w & 255])
        for k in range(4):
            row = rgbe[y, :, k]
            x = 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, row[x]])
                    x += run
                else:
                    j = x
                    while j < w and j - x < 128 and not (j + 2 < w and row[j] == row[j + 1] == row[j + 2]):
                        j += 1
                    j = max(j, x + 1)
                    out += bytes([j - x]) + row[x:j].tobytes()
                    x = j
    return bytes(out)


def fixture_files():
    rng = np.random.RandomState(2026)
    files = {}
    pal16 = rng.randint(0, 256, size=(16, 3)).astype(np.uint8)
    pal4 = rng.randint(0, 256, size=(4, 3)).astype(np.uint8)
    pal256 = rng.randint(0, 256, size=(256, 3)).astype(np.uint8)
    smooth = ((np.add.outer(np.arange(23), np.arange(37)) // 3) % 16).astype(np.uint8)
    files["gif/plain16"] = gif_bytes(37, 23, pal16, smooth)
    files["gif/87a_noise256"] = gif_bytes(40, 31, pal256, rng.randint(0, 256, size=(31, 40)), version=b"87a")
    files["gif/interlaced"] = gif_bytes(37, 23, pal16, smooth, interlace=True)
    files["gif/transparent"] = gif_bytes(37, 23, pal16, smooth, transparent=5)
    files["gif/sub_rect_bg"] = gif_bytes(30, 20, pal16, rng.randint(0, 16, size=(9, 11)), rect=(7, 4, 11, 9), bgindex=3)
    files["gif/sub_rect_bg_transparent"] = gif_bytes(30, 20, pal16, rng.randint(0, 16, size=(9, 11)), rect=(7, 4, 11, 9),
                                                     bgindex=2, transparent=2)
    files["gif/local_palette"] = gif_bytes(16, 16, pal16, rng.randint(0, 4, size=(16, 16)), local_palette=pal4, transparent=1)
    files["gif/many_clears"] = gif_bytes(64, 48, pal256, rng.randint(0, 256, size=(48, 64)), clear_every=50)
    files["gif/table_fills"] = gif_bytes(128, 96, pal256, rng.randint(0, 256, size=(96, 128)))
    files["gif/extensions"] = gif_bytes(20, 10, pal4, rng.randint(0, 4, size=(10, 20)), extra_ext=True, transparent=0)
    files["gif/two_colours_big_code"] = gif_bytes(33, 9, pal4[:2], rng.randint(0, 2, size=(9, 33)), min_code_size=4)

    p3 = rng.randint(0, 256, size=(3, 17, 29)).astype(np.uint8)
    p4 = rng.randint(0, 256, size=(4, 17, 29)).astype(np.uint8)
    p4[3, :6] = 255
    p4[3, 6:9] = 0
    runs = np.repeat(rng.randint(0, 256, size=(4, 12, 5)).astype(np.uint8), 9, axis=2)
    files["psd/rgb8_raw"] = psd_bytes(p3)
    files["psd/rgba8_raw_matte"] = psd_bytes(p4)
    files["psd/rgb8_rle"] = psd_bytes(p3, rle=True)
    files["psd/rgba8_rle_runs"] = psd_bytes(runs, rle=True, noop=True)
    files["psd/rgb16_raw"] = psd_bytes(rng.randint(0, 65536, size=(3, 11, 13)).astype(np.uint16), depth=16)
    files["psd/rgba16_raw"] = psd_bytes(rng.randint(0, 65536, size=(4, 11, 13)).astype(np.uint16), depth=16)
    files["psd/one_channel"] = psd_bytes(p3[:1])
    files["psd/five_channels"] = psd_bytes(rng.randint(0, 256, size=(5, 8, 8)).astype(np.uint8), rle=True)

    img = rng.randint(0, 256, size=(13, 21, 4)).astype(np.uint8)
    flat = np.repeat(rng.randint(0, 256, size=(13, 3, 4)).astype(np.uint8), 7, axis=1)
    wide = np.repeat(rng.randint(0, 256, size=(3, 2, 4)).astype(np.uint8), 200, axis=1)
    files["pic/rgb_raw"] = pic_bytes(img, [(0, 0xE0)])
    files["pic/rgba_raw_two_packets"] = pic_bytes(img, [(0, 0xE0), (0, 0x10)])
    files["pic/rgb_pure_rle"] = pic_bytes(flat, [(1, 0xE0)])
    files["pic/rgba_mixed_rle"] = pic_bytes(flat, [(2, 0xE0), (2, 0x10)])
    files["pic/mixed_rle_noise"] = pic_bytes(img, [(2, 0xF0)])
    files["pic/long_runs"] = pic_bytes(wide, [(2, 0xE0), (1, 0x10)])
    files["pic/red_only"] = pic_bytes(img, [(0, 0x80)])

    rgbe = rng.randint(0, 256, size=(9, 20, 4)).astype(np.uint8)
    rgbe[:, :, 3] = rng.randint(120, 136, size=(9, 20))
    rgbe[2, 3:9, 3] = 0
    soft = np.repeat(rgbe[:, :4], 5, axis=1)
    files["hdr/rle_noise"] = hdr_bytes(rgbe)
    files["hdr/rle_runs"] = hdr_bytes(soft)
    files["hdr/flat"] = hdr_bytes(rgbe, rle=False)
    files["hdr/narrow_flat"] = hdr_bytes(rgbe[:, :5])
    files["hdr/rgbe_magic"] = hdr_bytes(soft, magic=b"#?RGBE", extra_header=False)
    bright = rgbe.copy()
    bright[:, :, 3] = rng.randint(100, 160, size=(9, 20))
    files["hdr/wide_exponents"] = hdr_bytes(bright)
    return files
