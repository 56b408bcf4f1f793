// image_io.cpp -- PNG / PNM decode and PNG / PNM encode, written for this
// project (no third-party code).  The behaviour it has to match is the
// reference's use of stb: stbi_load(path,&w,&h,&n,3|4) for the maps
// (main/hmap.cpp:320-321,341-342; stb_image v2.27) and
// stbi_write_png(path,W,H,4,framebuf,W*4) for frames (main/hmap.cpp:158-160;
// stb_image_write v1.16).  Decoded pixels and encoded files are checked against
// the reference's own stb build (oracle/_ref) by tests/test_image_io.py.
//
// Decode side, stb conventions kept (vendor/stb_image.h):
//   * 16-bit samples are reduced by taking the high byte (:1170-1186)
//   * sub-byte grey is scaled by 0xff/0x55/0x11 (:4720-4770), palette expands to
//     RGB or RGBA (tRNS) (:4901-4937), a tRNS colour key gives alpha 0 (:4851-4874)
//   * channel conversion table of stbi__convert_format (:1735-1781),
//     luma = (77r+150g+29b)>>8 (:1726-1729)
//   * chunk CRCs and the zlib Adler-32 are not verified (stb does not either)
// Encode side: same filter heuristic (vendor/stb_image_write.h:1146-1174) and
// the same LZ77 + fixed-Huffman deflate (:895-1020) so that equal pixels give
// byte-identical files.
#include "image_io.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace hmrm {
namespace {

// ---------------------------------------------------------------- inflate --
struct BitReader {
	const uint8_t *p, *end;
	uint64_t buf = 0;
	int nbits = 0;   // bits held in buf
	int pad = 0;     // how many of them are zero padding appended past the input
	BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
	void fill() {
		while (nbits <= 56) {
			if (p < end) buf |= (uint64_t)*p++ << nbits;
			else pad += 8;
			nbits += 8;
		}
	}
	uint32_t peek(int n) {
		if (nbits < n) fill();
		return (uint32_t)(buf & ((1ull << n) - 1));
	}
	void drop(int n) { buf >>= n; nbits -= n; }
	uint32_t take(int n) {
		if (n == 0) return 0;
		uint32_t v = peek(n);
		drop(n);
		return v;
	}
	// true once bits of the padding (i.e. beyond the input) have been consumed
	bool overrun() const { return nbits < pad; }
};

struct Huff {
	static const int FAST = 10;
	uint16_t fast[1 << FAST]; // (len << 12) | symbol, 0 = not in fast table
	uint16_t count[16];
	uint16_t symbol[320];
	bool build(const uint8_t *lens, int n) {
		memset(count, 0, sizeof count);
		memset(fast, 0, sizeof fast);
		for (int i = 0; i < n; ++i) count[lens[i]]++;
		count[0] = 0;
		int left = 1;
		for (int l = 1; l < 16; ++l) {
			left <<= 1;
			left -= count[l];
			if (left < 0) return false; // over-subscribed
		}
		uint16_t offs[16];
		offs[1] = 0;
		for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
		for (int i = 0; i < n; ++i)
			if (lens[i]) symbol[offs[lens[i]]++] = (uint16_t)i;
		// canonical codes -> fast table (codes are sent MSB first, stream is LSB first)
		uint32_t code = 0;
		uint16_t next[16];
		next[0] = 0;
		for (int l = 1; l < 16; ++l) {
			code = (code + count[l - 1]) << 1;
			next[l] = (uint16_t)code;
		}
		for (int i = 0; i < n; ++i) {
			int l = lens[i];
			if (!l) continue;
			uint32_t c = next[l]++;
			if (l <= FAST) {
				uint32_t rev = 0;
				for (int b = 0; b < l; ++b) rev |= ((c >> b) & 1u) << (l - 1 - b);
				for (uint32_t k = rev; k < (1u << FAST); k += (1u << l))
					fast[k] = (uint16_t)((l << 12) | i);
			}
		}
		return true;
	}
	int decode(BitReader &br) const {
		uint32_t bits = br.peek(15);
		uint16_t f = fast[bits & ((1u << FAST) - 1)];
		if (f) {
			br.drop(f >> 12);
			return f & 0x0fff;
		}
		int code = 0, first = 0, index = 0;
		for (int l = 1; l < 16; ++l) {
			code |= (int)(bits & 1);
			bits >>= 1;
			int c = count[l];
			if (code - c < first) {
				br.drop(l);
				return symbol[index + (code - first)];
			}
			index += c;
			first += c;
			first <<= 1;
			code <<= 1;
		}
		return -1;
	}
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27,
                               31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                               2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129,
                                193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6,
                                6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

bool inflate_raw(BitReader &br, std::vector<uint8_t> *out, std::string *err) {
	Huff lit, dist;
	for (;;) {
		uint32_t final = br.take(1);
		uint32_t type = br.take(2);
		if (type == 0) {
			br.drop(br.nbits & 7);
			uint32_t len = br.take(16), nlen = br.take(16);
			if ((len ^ 0xffffu) != nlen) { *err = "zlib corrupt"; return false; }
			// whole bytes left in the bit buffer first, then straight from the input
			while (len && br.nbits >= 8) { out->push_back((uint8_t)br.take(8)); --len; }
			if (br.overrun()) { *err = "read past buffer"; return false; }
			if (len) { // bit buffer is empty now: copy straight from the input
				if ((size_t)(br.end - br.p) < len) { *err = "read past buffer"; return false; }
				out->insert(out->end(), br.p, br.p + len);
				br.p += len;
			}
		} else if (type == 1 || type == 2) {
			uint8_t lens[320];
			if (type == 1) {
				int i = 0;
				for (; i < 144; ++i) lens[i] = 8;
				for (; i < 256; ++i) lens[i] = 9;
				for (; i < 280; ++i) lens[i] = 7;
				for (; i < 288; ++i) lens[i] = 8;
				lit.build(lens, 288);
				for (i = 0; i < 32; ++i) lens[i] = 5;
				dist.build(lens, 32);
			} else {
				static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5,
				                                  11, 4, 12, 3, 13, 2, 14, 1, 15};
				int hlit = (int)br.take(5) + 257, hdist = (int)br.take(5) + 1,
				    hclen = (int)br.take(4) + 4;
				uint8_t cl[19];
				memset(cl, 0, sizeof cl);
				for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)br.take(3);
				Huff clh;
				if (!clh.build(cl, 19)) { *err = "bad codelengths"; return false; }
				int n = 0, total = hlit + hdist;
				while (n < total) {
					int c = clh.decode(br);
					if (c < 0 || c >= 19) { *err = "bad codelengths"; return false; }
					if (c < 16) { lens[n++] = (uint8_t)c; continue; }
					uint8_t fillv = 0;
					int rep;
					if (c == 16) {
						if (n == 0) { *err = "bad codelengths"; return false; }
						rep = 3 + (int)br.take(2);
						fillv = lens[n - 1];
					} else if (c == 17) rep = 3 + (int)br.take(3);
					else rep = 11 + (int)br.take(7);
					if (total - n < rep) { *err = "bad codelengths"; return false; }
					memset(lens + n, fillv, (size_t)rep);
					n += rep;
				}
				if (!lit.build(lens, hlit)) { *err = "bad codelengths"; return false; }
				if (!dist.build(lens + hlit, hdist)) { *err = "bad codelengths"; return false; }
			}
			for (;;) {
				int s = lit.decode(br);
				if (s < 0) { *err = "bad huffman code"; return false; }
				if (br.overrun()) { *err = "read past buffer"; return false; }
				if (s < 256) { out->push_back((uint8_t)s); continue; }
				if (s == 256) break;
				s -= 257;
				if (s >= 29) { *err = "bad huffman code"; return false; }
				uint32_t len = kLenBase[s] + br.take(kLenExtra[s]);
				int d = dist.decode(br);
				if (d < 0 || d >= 30) { *err = "bad huffman code"; return false; }
				size_t back = kDistBase[d] + br.take(kDistExtra[d]);
				if (back > out->size()) { *err = "bad dist"; return false; }
				size_t from = out->size() - back;
				for (uint32_t k = 0; k < len; ++k) out->push_back((*out)[from + k]);
				if (br.overrun()) { *err = "read past buffer"; return false; }
			}
		} else {
			*err = "zlib corrupt";
			return false;
		}
		if (br.overrun()) { *err = "read past buffer"; return false; }
		if (final) return true;
	}
}

// ------------------------------------------------------- channel convert --
inline uint8_t luma8(int r, int g, int b) { return (uint8_t)(((r * 77) + (g * 150) + (29 * b)) >> 8); }
inline uint16_t luma16(int r, int g, int b) { return (uint16_t)(((r * 77) + (g * 150) + (29 * b)) >> 8); }

// T = uint8_t or uint16_t; full = all-ones alpha
template <typename T>
std::vector<T> convert_channels(const std::vector<T> &src, int from, int to, size_t npix, T full) {
	if (from == to) return src;
	std::vector<T> dst(npix * (size_t)to);
	for (size_t i = 0; i < npix; ++i) {
		const T *s = &src[i * (size_t)from];
		T *d = &dst[i * (size_t)to];
		T r, g, b, a;
		if (from <= 2) { r = g = b = s[0]; a = (from == 2) ? s[1] : full; }
		else { r = s[0]; g = s[1]; b = s[2]; a = (from == 4) ? s[3] : full; }
		if (to <= 2) {
			d[0] = (from <= 2) ? s[0]
			       : (sizeof(T) == 1 ? (T)luma8(r, g, b) : (T)luma16(r, g, b));
			if (to == 2) d[1] = a;
		} else {
			d[0] = r; d[1] = g; d[2] = b;
			if (to == 4) d[3] = a;
		}
	}
	return dst;
}

// ------------------------------------------------------------------- PNG --
inline uint32_t be32(const uint8_t *p) {
	return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
inline int paeth_predict(int a, int b, int c) {
	int p = a + b - c;
	int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	if (pa <= pb && pa <= pc) return a;
	if (pb <= pc) return b;
	return c;
}

// Undo PNG row filters for one (sub)image; returns packed samples exactly as
// stored (bit depth untouched).  `bpp` = bytes per complete pixel (min 1).
bool unfilter(const uint8_t *raw, size_t raw_len, size_t row_bytes, uint32_t rows, int bpp,
              std::vector<uint8_t> *out, std::string *err) {
	if (raw_len < (row_bytes + 1) * (size_t)rows) { *err = "not enough pixels"; return false; }
	out->assign(row_bytes * (size_t)rows, 0);
	std::vector<uint8_t> zero(row_bytes, 0);
	for (uint32_t y = 0; y < rows; ++y) {
		const uint8_t *in = raw + (row_bytes + 1) * (size_t)y;
		int ft = *in++;
		uint8_t *cur = out->data() + row_bytes * (size_t)y;
		const uint8_t *up = y ? cur - row_bytes : zero.data();
		if (ft > 4) { *err = "invalid filter"; return false; }
		for (size_t i = 0; i < row_bytes; ++i) {
			int a = i >= (size_t)bpp ? cur[i - bpp] : 0;
			int b = up[i];
			int c = i >= (size_t)bpp ? up[i - bpp] : 0;
			int v = in[i];
			switch (ft) {
			case 1: v += a; break;
			case 2: v += b; break;
			case 3: v += (a + b) >> 1; break;
			case 4: v += paeth_predict(a, b, c); break;
			default: break;
			}
			cur[i] = (uint8_t)v;
		}
	}
	return true;
}

// Expand packed rows of `depth`-bit samples (n per pixel) into one uint16 per
// sample.  Sub-byte grey is scaled to 0..255 when `scale_grey`.
void expand_samples(const std::vector<uint8_t> &packed, size_t row_bytes, uint32_t w, uint32_t h,
                    int n, int depth, bool scale_grey, std::vector<uint16_t> *out) {
	static const int scale_tab[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
	const int scale = (scale_grey && depth <= 8) ? scale_tab[depth] : 1; // 16-bit grey is not scaled
	out->resize((size_t)w * h * n);
	for (uint32_t y = 0; y < h; ++y) {
		const uint8_t *row = packed.data() + row_bytes * (size_t)y;
		uint16_t *dst = out->data() + (size_t)y * w * n;
		const size_t cnt = (size_t)w * n;
		if (depth == 8) for (size_t i = 0; i < cnt; ++i) dst[i] = row[i];
		else if (depth == 16) for (size_t i = 0; i < cnt; ++i) dst[i] = (uint16_t)((row[2 * i] << 8) | row[2 * i + 1]);
		else {
			const int per = 8 / depth, mask = (1 << depth) - 1;
			for (size_t i = 0; i < cnt; ++i) {
				int shift = 8 - depth * (int)(i % per + 1);
				// stb multiplies in uint8 arithmetic: (stbi_uc)(scale * v)
				dst[i] = (uint8_t)(scale * ((row[i / per] >> shift) & mask));
			}
		}
	}
}

bool decode_png(const uint8_t *bytes, size_t len, int req_comp, Image *img, std::string *err) {
	static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
	if (len < 8 || memcmp(bytes, sig, 8) != 0) { *err = "bad png sig"; return false; }
	size_t pos = 8;
	bool first = true, have_idat = false;
	uint32_t w = 0, h = 0;
	int depth = 0, color = 0, interlace = 0, img_n = 0, pal_img_n = 0;
	uint8_t palette[1024];
	uint32_t pal_len = 0;
	bool has_trans = false;
	uint8_t tc8[3] = {0, 0, 0};
	uint16_t tc16[3] = {0, 0, 0};
	std::vector<uint8_t> idat;
	static const int scale_tab[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};

	for (;;) {
		if (pos + 8 > len) { *err = "outofdata"; return false; }
		uint32_t clen = be32(bytes + pos);
		const uint8_t *ty = bytes + pos + 4;
		pos += 8;
		uint32_t type = be32(ty);
		// stb reads past-the-end bytes as zeros; we refuse truncated chunks instead,
		// except that a missing trailing CRC is tolerated.
		if (clen > len - pos) { *err = "outofdata"; return false; }
		const uint8_t *d = bytes + pos;
		if (type == 0x43674249u) { *err = "CgBI (iPhone) PNG not supported"; return false; }
		if (type == 0x49484452u) { // IHDR
			if (!first) { *err = "multiple IHDR"; return false; }
			first = false;
			if (clen != 13) { *err = "bad IHDR len"; return false; }
			w = be32(d); h = be32(d + 4);
			if (w > (1u << 24) || h > (1u << 24)) { *err = "too large"; return false; }
			depth = d[8];
			if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) { *err = "1/2/4/8/16-bit only"; return false; }
			color = d[9];
			if (color > 6) { *err = "bad ctype"; return false; }
			if (color == 3 && depth == 16) { *err = "bad ctype"; return false; }
			if (color == 3) pal_img_n = 3;
			else if (color & 1) { *err = "bad ctype"; return false; }
			if (d[10]) { *err = "bad comp method"; return false; }
			if (d[11]) { *err = "bad filter method"; return false; }
			interlace = d[12];
			if (interlace > 1) { *err = "bad interlace method"; return false; }
			if (!w || !h) { *err = "0-pixel image"; return false; }
			if (!pal_img_n) {
				img_n = ((color & 2) ? 3 : 1) + ((color & 4) ? 1 : 0);
				if ((1u << 30) / w / (uint32_t)img_n < h) { *err = "too large"; return false; }
			} else {
				img_n = 1;
				if ((1u << 30) / w / 4 < h) { *err = "too large"; return false; }
			}
		} else if (first) {
			*err = "first not IHDR";
			return false;
		} else if (type == 0x504c5445u) { // PLTE
			if (clen > 256 * 3) { *err = "invalid PLTE"; return false; }
			pal_len = clen / 3;
			if (pal_len * 3 != clen) { *err = "invalid PLTE"; return false; }
			for (uint32_t i = 0; i < pal_len; ++i) {
				palette[i * 4 + 0] = d[i * 3 + 0];
				palette[i * 4 + 1] = d[i * 3 + 1];
				palette[i * 4 + 2] = d[i * 3 + 2];
				palette[i * 4 + 3] = 255;
			}
		} else if (type == 0x74524e53u) { // tRNS
			if (have_idat) { *err = "tRNS after IDAT"; return false; }
			if (pal_img_n) {
				if (pal_len == 0) { *err = "tRNS before PLTE"; return false; }
				if (clen > pal_len) { *err = "bad tRNS len"; return false; }
				pal_img_n = 4;
				for (uint32_t i = 0; i < clen; ++i) palette[i * 4 + 3] = d[i];
			} else {
				if (!(img_n & 1)) { *err = "tRNS with alpha"; return false; }
				if (clen != (uint32_t)img_n * 2) { *err = "bad tRNS len"; return false; }
				has_trans = true;
				for (int k = 0; k < img_n; ++k) {
					uint16_t v = (uint16_t)((d[2 * k] << 8) | d[2 * k + 1]);
					tc16[k] = v;
					tc8[k] = (uint8_t)((uint8_t)(v & 255) * (depth <= 8 ? scale_tab[depth] : 1));
				}
			}
		} else if (type == 0x49444154u) { // IDAT
			if (pal_img_n && !pal_len) { *err = "no PLTE"; return false; }
			idat.insert(idat.end(), d, d + clen);
			have_idat = true;
		} else if (type == 0x49454e44u) { // IEND
			break;
		} else if ((type & (1u << 29)) == 0) {
			*err = std::string((const char *)ty, 4) + " PNG chunk not known";
			return false;
		}
		pos += clen;
		pos += 4; // CRC, not verified
		if (pos > len) pos = len;
	}
	if (!have_idat) { *err = "no IDAT"; return false; }

	std::vector<uint8_t> raw;
	raw.reserve(((size_t)w * depth * img_n + 7) / 8 * h + h);
	if (!zlib_inflate(idat.data(), idat.size(), &raw, err)) return false;
	idat.clear();
	idat.shrink_to_fit();

	// filtered scanlines -> one uint16 per sample
	std::vector<uint16_t> samples((size_t)w * h * img_n);
	const int bpp = (img_n * depth + 7) / 8;
	const bool scale_grey = (color == 0);
	if (!interlace) {
		size_t row_bytes = ((size_t)img_n * w * depth + 7) >> 3;
		std::vector<uint8_t> packed;
		if (!unfilter(raw.data(), raw.size(), row_bytes, h, bpp, &packed, err)) return false;
		expand_samples(packed, row_bytes, w, h, img_n, depth, scale_grey, &samples);
	} else {
		static const int xorig[7] = {0, 4, 0, 2, 0, 1, 0}, yorig[7] = {0, 0, 4, 0, 2, 0, 1};
		static const int xspc[7] = {8, 8, 4, 4, 2, 2, 1}, yspc[7] = {8, 8, 8, 4, 4, 2, 2};
		size_t off = 0;
		for (int p = 0; p < 7; ++p) {
			uint32_t pw = (w - xorig[p] + xspc[p] - 1) / xspc[p];
			uint32_t ph = (h - yorig[p] + yspc[p] - 1) / yspc[p];
			if ((int)w <= xorig[p] || (int)h <= yorig[p]) pw = ph = 0;
			if (!pw || !ph) continue;
			size_t row_bytes = ((size_t)img_n * pw * depth + 7) >> 3;
			if (off > raw.size()) { *err = "not enough pixels"; return false; }
			std::vector<uint8_t> packed;
			if (!unfilter(raw.data() + off, raw.size() - off, row_bytes, ph, bpp, &packed, err)) return false;
			std::vector<uint16_t> sub;
			expand_samples(packed, row_bytes, pw, ph, img_n, depth, scale_grey, &sub);
			for (uint32_t j = 0; j < ph; ++j)
				for (uint32_t i = 0; i < pw; ++i) {
					size_t oy = (size_t)j * yspc[p] + yorig[p], ox = (size_t)i * xspc[p] + xorig[p];
					memcpy(&samples[(oy * w + ox) * img_n], &sub[((size_t)j * pw + i) * img_n],
					       sizeof(uint16_t) * (size_t)img_n);
				}
			off += (row_bytes + 1) * (size_t)ph;
		}
	}
	raw.clear();
	raw.shrink_to_fit();

	const size_t npix = (size_t)w * h;
	img->w = (int32_t)w;
	img->h = (int32_t)h;

	if (pal_img_n) {
		// palette index -> RGB(A); stb expands straight to req_comp when it is 3 or 4
		int out_n = pal_img_n;
		if (req_comp >= 3) out_n = req_comp;
		std::vector<uint8_t> px(npix * (size_t)out_n);
		for (size_t i = 0; i < npix; ++i) {
			const uint8_t *e = &palette[(samples[i] & 0xff) * 4];
			memcpy(&px[i * (size_t)out_n], e, (size_t)out_n);
		}
		img->comp_in_file = pal_img_n;
		int final_n = req_comp ? req_comp : out_n;
		img->px = convert_channels<uint8_t>(px, out_n, final_n, npix, 255);
		img->comp = final_n;
		return true;
	}

	// colour-key transparency adds an alpha channel to grey / RGB
	int src_n = img_n;
	if (has_trans) {
		std::vector<uint16_t> with_a(npix * (size_t)(img_n + 1));
		const uint16_t full = depth == 16 ? 65535 : 255;
		for (size_t i = 0; i < npix; ++i) {
			bool match = true;
			for (int k = 0; k < img_n; ++k) {
				uint16_t key = depth == 16 ? tc16[k] : tc8[k];
				with_a[i * (size_t)(img_n + 1) + k] = samples[i * (size_t)img_n + k];
				if (samples[i * (size_t)img_n + k] != key) match = false;
			}
			with_a[i * (size_t)(img_n + 1) + img_n] = match ? 0 : full;
		}
		samples.swap(with_a);
		src_n = img_n + 1;
	}
	img->comp_in_file = src_n;
	const int final_n = req_comp ? req_comp : src_n;
	if (depth == 16) {
		std::vector<uint16_t> conv = convert_channels<uint16_t>(samples, src_n, final_n, npix, 65535);
		img->px.resize(conv.size());
		for (size_t i = 0; i < conv.size(); ++i) img->px[i] = (uint8_t)((conv[i] >> 8) & 0xff);
	} else {
		std::vector<uint8_t> s8(samples.size());
		for (size_t i = 0; i < samples.size(); ++i) s8[i] = (uint8_t)samples[i];
		img->px = convert_channels<uint8_t>(s8, src_n, final_n, npix, 255);
	}
	img->comp = final_n;
	return true;
}

// ------------------------------------------------------------------- PNM --
bool decode_pnm(const uint8_t *bytes, size_t len, int req_comp, Image *img, std::string *err) {
	if (len < 3 || bytes[0] != 'P' || (bytes[1] != '5' && bytes[1] != '6')) { *err = "not pnm"; return false; }
	const int n = bytes[1] == '6' ? 3 : 1;
	size_t pos = 2;
	auto get = [&]() -> int { return pos < len ? bytes[pos++] : 0; };
	auto eof = [&]() { return pos >= len; };
	auto is_space = [](int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
	int c = get();
	auto skip_ws = [&]() {
		for (;;) {
			while (!eof() && is_space(c)) c = get();
			if (eof() || c != '#') break;
			while (!eof() && c != '\n' && c != '\r') c = get();
		}
	};
	auto integer = [&]() {
		long v = 0;
		while (!eof() && c >= '0' && c <= '9') {
			v = v * 10 + (c - '0');
			if (v > (1L << 30)) v = (1L << 30);
			c = get();
		}
		return (int)v;
	};
	skip_ws();
	int w = integer();
	skip_ws();
	int h = integer();
	skip_ws();
	int maxv = integer();
	if (maxv > 65535) { *err = "max value > 65535"; return false; }
	if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24)) { *err = "bad pnm size"; return false; }
	const int bytes_per = maxv > 255 ? 2 : 1;
	const size_t npix = (size_t)w * (size_t)h;
	const size_t need = npix * (size_t)n * (size_t)bytes_per;
	// `c` already holds the single whitespace byte that ends the header; pos is at the raster
	std::vector<uint8_t> data(need, 0);
	size_t avail = len > pos ? len - pos : 0;
	memcpy(data.data(), bytes + pos, avail < need ? avail : need); // short files: rest stays 0 (stb leaves it unread)
	img->w = w;
	img->h = h;
	img->comp_in_file = n;
	const int final_n = req_comp ? req_comp : n;
	if (bytes_per == 2) {
		// stb v2.27 reads the big-endian samples as native little-endian uint16 and keeps
		// ">> 8" of that, i.e. the SECOND byte of each sample; its channel conversion on
		// such data is not meaningful, so only the no-conversion case is offered.
		if (final_n != n) { *err = "16-bit PNM with channel conversion not supported"; return false; }
		img->px.resize(npix * (size_t)n);
		for (size_t i = 0; i < npix * (size_t)n; ++i) img->px[i] = data[2 * i + 1];
	} else {
		img->px = convert_channels<uint8_t>(data, n, final_n, npix, 255);
	}
	img->comp = final_n;
	return true;
}

} // namespace

bool zlib_inflate(const uint8_t *src, size_t len, std::vector<uint8_t> *out, std::string *err) {
	if (len < 2) { *err = "bad zlib header"; return false; }
	int cmf = src[0], flg = src[1];
	if ((cmf * 256 + flg) % 31 != 0) { *err = "bad zlib header"; return false; }
	if (flg & 32) { *err = "no preset dict"; return false; }
	if ((cmf & 15) != 8) { *err = "bad compression"; return false; }
	BitReader br(src + 2, src + len);
	return inflate_raw(br, out, err);
}

std::vector<uint8_t> convert_channels8(const std::vector<uint8_t> &src, int from, int to, size_t npix) {
	return convert_channels<uint8_t>(src, from, to, npix, 255);
}

// Format detection in the order the reference's loader tries them (stb_image v2.27 stbi__load_main): the formats
// with a real signature first -- PNG, BMP, GIF, PSD, PIC -- then JPEG, PNM and Radiance HDR, and TGA last (it has no
// signature, only a header that must be plausible).  All nine formats the reference's README lists for maps.
bool decode_image(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	if (req_comp < 0 || req_comp > 4) { *err = "bad req_comp"; return false; }
	static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
	if (len >= 8 && memcmp(bytes, sig, 8) == 0) return decode_png(bytes, len, req_comp, out, err);
	if (looks_like_bmp(bytes, len)) return decode_bmp(bytes, len, req_comp, out, err);
	if (looks_like_gif(bytes, len)) return decode_gif(bytes, len, req_comp, out, err);
	if (looks_like_psd(bytes, len)) return decode_psd(bytes, len, req_comp, out, err);
	if (looks_like_pic(bytes, len)) return decode_pic(bytes, len, req_comp, out, err);
	if (len >= 2 && bytes[0] == 0xff && bytes[1] == 0xd8) return decode_jpeg(bytes, len, req_comp, out, err);
	if (len >= 2 && bytes[0] == 'P' && (bytes[1] == '5' || bytes[1] == '6'))
		return decode_pnm(bytes, len, req_comp, out, err);
	if (looks_like_hdr(bytes, len)) return decode_hdr(bytes, len, req_comp, out, err);
	if (looks_like_tga(bytes, len)) return decode_tga(bytes, len, req_comp, out, err);
	*err = "unknown image type (supported: PNG, JPEG, BMP, GIF, PSD, PIC, HDR, TGA, binary PGM/PPM)";
	return false;
}

bool load_image_file(const char *path, int req_comp, Image *out, std::string *err) {
	FILE *f = fopen(path, "rb");
	if (!f) { *err = "can't fopen"; return false; }
	std::vector<uint8_t> buf;
	uint8_t chunk[1 << 16];
	size_t got;
	while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
	fclose(f);
	return decode_image(buf.data(), buf.size(), req_comp, out, err);
}

// ---------------------------------------------------------------- deflate --
// Same token stream as stb_image_write's stbi_zlib_compress (:895-1020), bit for bit, written
// for speed: the recording path encodes one 4K frame per orbit step on host threads and is
// bound by this function, not by the GPU.  What must be preserved: the hash, the 16384
// buckets that keep between `quality` and 2*quality most recent starts (oldest half dropped
// when full), the candidate order (oldest first, a later equal-length match replaces an
// earlier one), the one-step lazy rule, and the fixed-Huffman coding.  What is free: how a
// match length is counted (8 bytes at a time), rejecting a candidate from the one byte that
// would have to extend the best match so far, table-driven codes, a 64-bit bit buffer.
namespace {
struct BitWriter {
	uint8_t *p = nullptr; // write cursor into a buffer the caller sized for the worst case
	uint64_t buf = 0;
	int count = 0;
	inline void add(uint32_t code, int bits) { // bits <= 25 here; flushed below 32
		buf |= (uint64_t)code << count;
		count += bits;
		if (count >= 32) {
			p[0] = (uint8_t)buf; p[1] = (uint8_t)(buf >> 8); p[2] = (uint8_t)(buf >> 16); p[3] = (uint8_t)(buf >> 24);
			p += 4;
			buf >>= 32;
			count -= 32;
		}
	}
	void flush_bytes() { // whole bytes only; count ends < 8
		while (count >= 8) {
			*p++ = (uint8_t)buf;
			buf >>= 8;
			count -= 8;
		}
	}
	static uint32_t rev(uint32_t code, int bits) {
		uint32_t r = 0;
		while (bits--) { r = (r << 1) | (code & 1); code >>= 1; }
		return r;
	}
};

struct DeflateTables {
	uint16_t lit_code[288];
	uint8_t lit_bits[288];
	uint8_t len_sym[259];  // match length 3..258 -> index into lengthc
	uint8_t dist_sym[32768]; // distance 1..32767 -> index into distc
	uint16_t lengthc[30] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
	                        35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 259};
	uint8_t lengtheb[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
	uint16_t distc[31] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
	                      257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 32768};
	uint8_t disteb[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
	uint8_t dist_code5[30]; // the 5-bit distance symbol, bit-reversed
	DeflateTables() {
		for (int n = 0; n < 288; ++n) { // fixed literal/length code of RFC 1951 3.2.6
			if (n <= 143) { lit_code[n] = (uint16_t)BitWriter::rev(0x30 + n, 8); lit_bits[n] = 8; }
			else if (n <= 255) { lit_code[n] = (uint16_t)BitWriter::rev(0x190 + n - 144, 9); lit_bits[n] = 9; }
			else if (n <= 279) { lit_code[n] = (uint16_t)BitWriter::rev(n - 256, 7); lit_bits[n] = 7; }
			else { lit_code[n] = (uint16_t)BitWriter::rev(0xc0 + n - 280, 8); lit_bits[n] = 8; }
		}
		for (int len = 3; len <= 258; ++len) { // stb: for (j = 0; best > lengthc[j+1]-1; ++j);
			int j = 0;
			while (len > lengthc[j + 1] - 1) ++j;
			len_sym[len] = (uint8_t)j;
		}
		dist_sym[0] = 0;
		for (int d = 1; d < 32768; ++d) {
			int j = 0;
			while (d > distc[j + 1] - 1) ++j;
			dist_sym[d] = (uint8_t)j;
		}
		for (int j = 0; j < 30; ++j) dist_code5[j] = (uint8_t)BitWriter::rev((uint32_t)j, 5);
	}
};
const DeflateTables &deflate_tables() {
	static const DeflateTables t;
	return t;
}

inline uint32_t hash3(const uint8_t *d) {
	uint32_t h = d[0] + (d[1] << 8) + (d[2] << 16);
	h ^= h << 3; h += h >> 5; h ^= h << 4; h += h >> 17; h ^= h << 25; h += h >> 6;
	return h;
}
// stbiw__zlib_countm: common prefix of a and b, at most min(limit, 258)
inline int match_len(const uint8_t *a, const uint8_t *b, int limit) {
	const int lim = limit < 258 ? limit : 258;
	int i = 0;
	while (i + 8 <= lim) {
		uint64_t x, y;
		memcpy(&x, a + i, 8);
		memcpy(&y, b + i, 8);
		if (x != y) return i + (__builtin_ctzll(x ^ y) >> 3); // (little endian: lowest differing byte)
		i += 8;
	}
	while (i < lim && a[i] == b[i]) ++i;
	return i;
}
} // namespace

void zlib_deflate_stb(const uint8_t *data, int data_len, int quality, std::vector<uint8_t> *out) {
	const DeflateTables &T = deflate_tables();
	const int NB = 16384;
	if (quality < 5) quality = 5;
	const int cap = 2 * quality;
	// worst case of the fixed code: 9 bits per literal, plus header, end-of-block, padding, Adler-32
	out->assign((size_t)data_len + (size_t)data_len / 8 + 64, 0);
	(*out)[0] = 0x78;
	(*out)[1] = 0x5e;
	BitWriter bw;
	bw.p = out->data() + 2;
	bw.add(1, 1); // BFINAL
	bw.add(1, 2); // BTYPE = fixed Huffman
	// bucket = up to `cap` (position, first three bytes) pairs, oldest first.  The three bytes
	// let a scan drop hash collisions -- candidates that cannot reach the minimum match of 3 --
	// without touching the data, four at a time.
	const int stride = (cap + 3) & ~3;
	std::vector<int> pos_tab((size_t)NB * (size_t)stride);
	std::vector<uint32_t> tag_tab((size_t)NB * (size_t)stride, 0xffffffffu);
	std::vector<uint8_t> fill((size_t)NB, 0);
	auto first3 = [&](int at) { return (uint32_t)data[at] | ((uint32_t)data[at + 1] << 8) | ((uint32_t)data[at + 2] << 16); };
	// indices j < n (ascending, as a bit mask) whose tag equals `want`
	auto same3 = [&](const uint32_t *tags, int n, uint32_t want) -> uint32_t {
		uint32_t m = 0;
#if defined(__SSE2__)
		const __m128i w = _mm_set1_epi32((int)want);
		for (int g = 0; g < n; g += 4) {
			const __m128i t = _mm_loadu_si128((const __m128i *)(tags + g));
			m |= (uint32_t)_mm_movemask_ps(_mm_castsi128_ps(_mm_cmpeq_epi32(t, w))) << g;
		}
#else
		for (int j = 0; j < n; ++j) m |= (uint32_t)(tags[j] == want) << j;
#endif
		return n >= 32 ? m : (m & ((1u << n) - 1u));
	};

	int i = 0;
	while (i < data_len - 3) {
		if (i + 8 < data_len - 3) { // the buckets are 2 MB of random accesses: fetch the one a few bytes ahead
			const size_t hp = (size_t)(hash3(data + i + 8) & (NB - 1)) * (size_t)stride;
			__builtin_prefetch(&pos_tab[hp]);
			__builtin_prefetch(&tag_tab[hp]);
		}
		const int h = (int)(hash3(data + i) & (NB - 1));
		int best = 3, bestloc = -1;
		int *hl = &pos_tab[(size_t)h * (size_t)stride];
		uint32_t *ht = &tag_tab[(size_t)h * (size_t)stride];
		int n = fill[(size_t)h];
		const int limit = data_len - i;
		const uint32_t cur3 = first3(i);
		for (uint32_t m = same3(ht, n, cur3); m; m &= m - 1) {
			const int c = hl[__builtin_ctz(m)];
			if (c > i - 32768) {
				// d >= best needs the first `best` bytes to agree: look at the last of them first
				if (best - 1 < limit && data[c + best - 1] != data[i + best - 1]) continue;
				const int d = match_len(data + c, data + i, limit);
				if (d >= best) { best = d; bestloc = c; }
			}
		}
		if (n == cap) { // keep the newer half
			memmove(hl, hl + quality, sizeof(int) * (size_t)quality);
			memmove(ht, ht + quality, sizeof(uint32_t) * (size_t)quality);
			n = quality;
		}
		hl[n] = i;
		ht[n] = cur3;
		fill[(size_t)h] = (uint8_t)(n + 1);
		if (bestloc >= 0) { // lazy: a longer match starting one byte later wins (then emit a literal)
			const int h2 = (int)(hash3(data + i + 1) & (NB - 1));
			const int *hl2 = &pos_tab[(size_t)h2 * (size_t)stride];
			const int limit2 = data_len - i - 1;
			if (best < limit2 && best < 258) { // else no e > best exists
				for (uint32_t m = same3(&tag_tab[(size_t)h2 * (size_t)stride], fill[(size_t)h2], first3(i + 1)); m; m &= m - 1) {
					const int c = hl2[__builtin_ctz(m)];
					if (c > i - 32767) {
						// e > best needs best+1 agreeing bytes
						if (data[c + best] != data[i + 1 + best]) continue;
						const int e = match_len(data + c, data + i + 1, limit2);
						if (e > best) { bestloc = -1; break; }
					}
				}
			}
		}
		if (bestloc >= 0) {
			const int d = i - bestloc;
			const int j = T.len_sym[best];
			bw.add(T.lit_code[j + 257], T.lit_bits[j + 257]);
			if (T.lengtheb[j]) bw.add((uint32_t)(best - T.lengthc[j]), T.lengtheb[j]);
			const int k = T.dist_sym[d];
			bw.add(T.dist_code5[k], 5);
			if (T.disteb[k]) bw.add((uint32_t)(d - T.distc[k]), T.disteb[k]);
			i += best;
		} else {
			bw.add(T.lit_code[data[i]], T.lit_bits[data[i]]);
			++i;
		}
	}
	for (; i < data_len; ++i) bw.add(T.lit_code[data[i]], T.lit_bits[data[i]]);
	bw.add(T.lit_code[256], T.lit_bits[256]);
	bw.flush_bytes();
	if (bw.count) bw.add(0, 8 - bw.count); // pad the last byte with zero bits
	bw.flush_bytes();
	out->resize((size_t)(bw.p - out->data()));

	if ((long)out->size() > (long)data_len + 2 + ((data_len + 32766) / 32767) * 5) {
		out->resize(2);
		for (int j = 0; j < data_len;) {
			int blocklen = data_len - j;
			if (blocklen > 32767) blocklen = 32767;
			out->push_back((uint8_t)(data_len - j == blocklen));
			out->push_back((uint8_t)blocklen);
			out->push_back((uint8_t)(blocklen >> 8));
			out->push_back((uint8_t)~blocklen);
			out->push_back((uint8_t)(~blocklen >> 8));
			out->insert(out->end(), data + j, data + j + blocklen);
			j += blocklen;
		}
	}
	// Adler-32, reduced every 5552 bytes (the largest run that cannot overflow 32 bits)
	uint32_t s1 = 1, s2 = 0;
	for (int j = 0; j < data_len;) {
		const int run = data_len - j < 5552 ? data_len - j : 5552;
		for (int k = 0; k < run; ++k) {
			s1 += data[j + k];
			s2 += s1;
		}
		s1 %= 65521;
		s2 %= 65521;
		j += run;
	}
	out->push_back((uint8_t)(s2 >> 8));
	out->push_back((uint8_t)s2);
	out->push_back((uint8_t)(s1 >> 8));
	out->push_back((uint8_t)s1);
}

// -------------------------------------------------------------------- PNG --
namespace {
uint32_t crc32_png(const uint8_t *p, size_t n) {
	static uint32_t table[256];
	static bool init = false;
	if (!init) {
		for (uint32_t i = 0; i < 256; ++i) {
			uint32_t c = i;
			for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
			table[i] = c;
		}
		init = true;
	}
	uint32_t crc = ~0u;
	for (size_t i = 0; i < n; ++i) crc = (crc >> 8) ^ table[p[i] ^ (crc & 0xff)];
	return ~crc;
}
void put32(std::vector<uint8_t> *o, uint32_t v) {
	o->push_back((uint8_t)(v >> 24));
	o->push_back((uint8_t)(v >> 16));
	o->push_back((uint8_t)(v >> 8));
	o->push_back((uint8_t)v);
}
void put_chunk(std::vector<uint8_t> *o, const char tag[4], const uint8_t *data, size_t n) {
	put32(o, (uint32_t)n);
	size_t start = o->size();
	o->insert(o->end(), tag, tag + 4);
	if (n) o->insert(o->end(), data, data + n);
	put32(o, crc32_png(o->data() + start, n + 4));
}

// Row filters as stb applies them (stb_image_write.h:1073-1126).  kind: 0 none, 1 sub, 2 up,
// 3 average, 4 paeth; on the first row "up" degenerates to none, "average" to left>>1 and
// "paeth" to sub.  `n` = bytes per pixel.  The encoder needs, per row, the sum of |int8(v)|
// of every filter (first minimum wins) and then the bytes of the winner only: one pass
// for the five sums, one pass to write.
inline int abs_i8(int v) {
	const int s = (int)(int8_t)v;
	return s < 0 ? -s : s;
}
void filter_sums(const uint8_t *row, const uint8_t *prev, int row_bytes, int n, int est[5]) {
	int e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
	if (prev) {
		for (int i = 0; i < n && i < row_bytes; ++i) { // no left neighbour: left = ul = 0
			const int v = row[i], up = prev[i];
			e0 += abs_i8(v);
			e1 += abs_i8(v);
			e2 += abs_i8(v - up);
			e3 += abs_i8(v - (up >> 1));
			e4 += abs_i8(v - paeth_predict(0, up, 0));
		}
		for (int i = n; i < row_bytes; ++i) {
			const int v = row[i], left = row[i - n], up = prev[i], ul = prev[i - n];
			e0 += abs_i8(v);
			e1 += abs_i8(v - left);
			e2 += abs_i8(v - up);
			e3 += abs_i8(v - ((left + up) >> 1));
			e4 += abs_i8(v - paeth_predict(left, up, ul));
		}
	} else {
		for (int i = 0; i < n && i < row_bytes; ++i) {
			const int a = abs_i8(row[i]);
			e0 += a; e1 += a; e2 += a; e3 += a; e4 += a;
		}
		for (int i = n; i < row_bytes; ++i) {
			const int v = row[i], left = row[i - n];
			e0 += abs_i8(v);
			e1 += abs_i8(v - left);
			e2 += abs_i8(v);
			e3 += abs_i8(v - (left >> 1));
			e4 += abs_i8(v - left);
		}
	}
	est[0] = e0; est[1] = e1; est[2] = e2; est[3] = e3; est[4] = e4;
}
void filter_row(const uint8_t *row, const uint8_t *prev, int row_bytes, int n, int kind, uint8_t *dst) {
	if (!prev) {
		static const int first_row[5] = {0, 1, 0, 5, 6};
		kind = first_row[kind];
	}
	const int head = n < row_bytes ? n : row_bytes;
	switch (kind) {
	case 0:
		memcpy(dst, row, (size_t)row_bytes);
		break;
	case 1:
	case 6:
		for (int i = 0; i < head; ++i) dst[i] = row[i];
		for (int i = n; i < row_bytes; ++i) dst[i] = (uint8_t)(row[i] - row[i - n]);
		break;
	case 2:
		for (int i = 0; i < row_bytes; ++i) dst[i] = (uint8_t)(row[i] - prev[i]);
		break;
	case 3:
		for (int i = 0; i < head; ++i) dst[i] = (uint8_t)(row[i] - (prev[i] >> 1));
		for (int i = n; i < row_bytes; ++i) dst[i] = (uint8_t)(row[i] - ((row[i - n] + prev[i]) >> 1));
		break;
	case 4:
		for (int i = 0; i < head; ++i) dst[i] = (uint8_t)(row[i] - paeth_predict(0, prev[i], 0));
		for (int i = n; i < row_bytes; ++i) dst[i] = (uint8_t)(row[i] - paeth_predict(row[i - n], prev[i], prev[i - n]));
		break;
	case 5:
		for (int i = 0; i < head; ++i) dst[i] = row[i];
		for (int i = n; i < row_bytes; ++i) dst[i] = (uint8_t)(row[i] - (row[i - n] >> 1));
		break;
	default:
		break;
	}
}
} // namespace

bool encode_png(int32_t w, int32_t h, int32_t comp, const uint8_t *data, size_t stride_bytes,
                std::vector<uint8_t> *out) {
	if (w <= 0 || h <= 0 || comp < 1 || comp > 4 || !data) return false;
	if (stride_bytes == 0) stride_bytes = (size_t)w * comp;
	const int row_bytes = w * comp;
	if ((long long)(row_bytes + 1) * h > 0x7fffffffLL) return false; // stb's int arithmetic limit
	std::vector<uint8_t> filt((size_t)(row_bytes + 1) * h);
	for (int y = 0; y < h; ++y) {
		const uint8_t *row = data + stride_bytes * (size_t)y;
		const uint8_t *prev = y ? row - stride_bytes : nullptr;
		int est[5];
		filter_sums(row, prev, row_bytes, comp, est);
		int best = 0, best_val = 0x7fffffff;
		for (int ft = 0; ft < 5; ++ft)
			if (est[ft] < best_val) { best_val = est[ft]; best = ft; }
		uint8_t *dst = &filt[(size_t)(row_bytes + 1) * y];
		dst[0] = (uint8_t)best;
		filter_row(row, prev, row_bytes, comp, best, dst + 1);
	}
	std::vector<uint8_t> z;
	zlib_deflate_stb(filt.data(), (int)filt.size(), 8, &z);
	filt.clear();
	filt.shrink_to_fit();

	static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
	static const int ctype[5] = {-1, 0, 4, 2, 6};
	out->clear();
	out->reserve(z.size() + 64);
	out->insert(out->end(), sig, sig + 8);
	uint8_t ihdr[13];
	ihdr[0] = (uint8_t)(w >> 24); ihdr[1] = (uint8_t)(w >> 16); ihdr[2] = (uint8_t)(w >> 8); ihdr[3] = (uint8_t)w;
	ihdr[4] = (uint8_t)(h >> 24); ihdr[5] = (uint8_t)(h >> 16); ihdr[6] = (uint8_t)(h >> 8); ihdr[7] = (uint8_t)h;
	ihdr[8] = 8;
	ihdr[9] = (uint8_t)ctype[comp];
	ihdr[10] = ihdr[11] = ihdr[12] = 0;
	put_chunk(out, "IHDR", ihdr, 13);
	put_chunk(out, "IDAT", z.data(), z.size());
	put_chunk(out, "IEND", nullptr, 0);
	return true;
}

bool encode_pnm(int32_t w, int32_t h, int32_t comp, const uint8_t *data, size_t stride_bytes,
                std::vector<uint8_t> *out) {
	if (w <= 0 || h <= 0 || comp < 1 || comp > 4 || !data) return false;
	if (stride_bytes == 0) stride_bytes = (size_t)w * comp;
	const int out_n = comp >= 3 ? 3 : 1;
	char header[64];
	int hl = snprintf(header, sizeof header, "P%c\n%d %d\n255\n", out_n == 3 ? '6' : '5', w, h);
	out->clear();
	out->reserve((size_t)hl + (size_t)w * h * out_n);
	out->insert(out->end(), header, header + hl);
	for (int y = 0; y < h; ++y) {
		const uint8_t *row = data + stride_bytes * (size_t)y;
		for (int x = 0; x < w; ++x)
			for (int k = 0; k < out_n; ++k) out->push_back(row[(size_t)x * comp + k]);
	}
	return true;
}

bool write_file(const char *path, const uint8_t *data, size_t len) {
	FILE *f = fopen(path, "wb");
	if (!f) return false;
	size_t put = fwrite(data, 1, len, f);
	int rc = fclose(f);
	return put == len && rc == 0;
}

} // namespace hmrm
