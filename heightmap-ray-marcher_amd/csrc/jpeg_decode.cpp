// jpeg_decode.cpp -- baseline and progressive JPEG decoder (8-bit, Huffman),
// written for this project.  The maps of the reference are usually photographs
// (README.md: "The image can be any format supported by stb_image.h: JPEG, ...";
// sample_config.txt uses .jpg), loaded with stbi_load(path,&w,&h,&n,3|4)
// (main/hmap.cpp:320-321,341-342).  A JPEG decoder is only a drop-in if it produces
// the same pixels as that loader, and JPEG leaves IDCT precision, chroma upsampling
// and colour conversion to the implementation.  So the arithmetic below follows what
// stb_image v2.27 does (vendor/stb_image.h), and tests/test_image_io.py compares the
// result with the reference's own stb build (oracle/_ref) pixel for pixel:
//   * coefficients are kept in 16-bit storage after dequantisation (:2198,:3039-3044)
//   * IDCT: the integer "islow" variant at 12 fractional bits, column pass rounded to
//     2 extra bits, row pass rounded once, +128, clamp (:2406-2493)
//   * chroma upsampling: triangle filters (3:1 weights) for 2x horizontal, 2x
//     vertical and 2x2, nearest for other ratios, sample row chosen per output row
//     as in :3873-3887
//   * YCbCr -> RGB in 20-bit fixed point with stb's reduced-precision constants
//     (:3604-3630); Adobe APP14 transform flags, CMYK/YCCK through the 8x8 "blinn"
//     multiply (:3805-3809, :3905-3928)
//   * channel conversion to req_comp as load_jpeg_image does it (:3811-3972)
// Not supported (as in stb): arithmetic coding, 12-bit, lossless, hierarchical.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "image_io.hpp"

namespace hmrm {
namespace {

const uint8_t kDezigzag[64 + 15] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    // guard entries so that a corrupt run cannot index out of range
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

const int kMarkerNone = 0xff;

struct Huffman {
	uint16_t code[256];
	uint8_t values[256];
	uint8_t size[257];
	uint32_t maxcode[18];
	int delta[17];
	bool build(const int count[16]) {
		int k = 0;
		for (int i = 0; i < 16; ++i)
			for (int j = 0; j < count[i]; ++j) {
				if (k >= 256) return false;
				size[k++] = (uint8_t)(i + 1);
			}
		size[k] = 0;
		uint32_t c = 0;
		k = 0;
		for (int j = 1; j <= 16; ++j) {
			delta[j] = k - (int)c;
			if (size[k] == j) {
				while (size[k] == j) code[k++] = (uint16_t)(c++);
				if (c - 1 >= (1u << j)) return false;
			}
			maxcode[j] = c << (16 - j);
			c <<= 1;
		}
		maxcode[17] = 0xffffffffu;
		return true;
	}
};

struct Component {
	int id = 0, h = 0, v = 0, tq = 0, hd = 0, ha = 0, dc_pred = 0;
	int x = 0, y = 0, w2 = 0, h2 = 0;
	std::vector<uint8_t> data;   // w2 x h2 samples
	std::vector<int16_t> coeff;  // progressive: 64 per block, coeff_w blocks per row
	int coeff_w = 0;
};

struct Decoder {
	const uint8_t *p, *end;
	std::string err;
	Huffman huff_dc[4], huff_ac[4];
	uint16_t dequant[4][64];
	int img_x = 0, img_y = 0, img_n = 0;
	int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
	Component comp[4];
	uint32_t code_buffer = 0;
	int code_bits = 0;
	int marker = kMarkerNone;
	bool nomore = false;
	bool progressive = false;
	int spec_start = 0, spec_end = 0, succ_high = 0, succ_low = 0, eob_run = 0;
	bool jfif = false;
	int app14 = -1, rgb = 0;
	int scan_n = 0, order[4];
	int restart_interval = 0, todo = 0;

	bool fail(const char *m) { if (err.empty()) err = m; return false; }
	bool at_eof() const { return p >= end; }
	int get8() { return p < end ? *p++ : 0; } // past the end reads as zeros, like stb's reader
	int get16() { int a = get8(); return (a << 8) | get8(); }
	void skip(int n) { if (n > 0) p = (end - p) < n ? end : p + n; }

	// ---- entropy-coded segment bit reader (MSB first, 0xFF00 stuffing, markers stop it)
	void grow() {
		do {
			unsigned b = nomore ? 0u : (unsigned)get8();
			if (b == 0xff) {
				int c = get8();
				while (c == 0xff) c = get8();
				if (c != 0) {
					marker = c;
					nomore = true;
					return;
				}
			}
			const int sh = 24 - code_bits; // > 31 only after a corrupt stream drove code_bits negative
			if (sh < 32) code_buffer |= b << sh;
			code_bits += 8;
		} while (code_bits <= 24);
	}
	int huff_decode(const Huffman &h) {
		if (code_bits < 16) grow();
		const uint32_t temp = code_buffer >> 16;
		int k;
		for (k = 1; k < 17; ++k) // (a table that was never defined has maxcode all zero: k runs to 17)
			if (temp < h.maxcode[k]) break;
		if (k == 17) { code_bits -= 16; return -1; }
		if (k > code_bits) return -1;
		const int c = (int)((code_buffer >> (32 - k)) & ((1u << k) - 1)) + h.delta[k];
		if (c < 0 || c > 255) return -1;
		code_bits -= k;
		code_buffer <<= k;
		return h.values[c];
	}
	static uint32_t rotl(uint32_t v, int n) { n &= 31; return n ? (v << n) | (v >> (32 - n)) : v; }
	int extend_receive(int n) { // JPEG RECEIVE + EXTEND
		if (code_bits < n) grow();
		const int sgn = (int)(code_buffer >> 31);
		uint32_t k = rotl(code_buffer, n);
		const uint32_t mask = (1u << n) - 1;
		code_buffer = k & ~mask;
		k &= mask;
		code_bits -= n;
		const int bias = -(1 << n) + 1;
		return (int)k + (sgn ? 0 : bias);
	}
	int get_bits(int n) {
		if (code_bits < n) grow();
		uint32_t k = rotl(code_buffer, n);
		const uint32_t mask = (1u << n) - 1;
		code_buffer = k & ~mask;
		k &= mask;
		code_bits -= n;
		return (int)k;
	}
	bool get_bit() {
		if (code_bits < 1) grow();
		const uint32_t k = code_buffer;
		code_buffer <<= 1;
		--code_bits;
		return (k & 0x80000000u) != 0;
	}

	// ---- block decoders
	bool decode_block(int16_t data[64], const Huffman &hdc, const Huffman &hac, int b, const uint16_t *dq) {
		if (code_bits < 16) grow();
		const int t = huff_decode(hdc);
		if (t < 0 || t > 15) return fail("bad huffman code");
		memset(data, 0, 64 * sizeof(int16_t));
		const int diff = t ? extend_receive(t) : 0;
		const int dc = comp[b].dc_pred + diff;
		comp[b].dc_pred = dc;
		data[0] = (int16_t)(dc * dq[0]);
		int k = 1;
		do {
			const int rs = huff_decode(hac);
			if (rs < 0) return fail("bad huffman code");
			const int s = rs & 15, r = rs >> 4;
			if (s == 0) {
				if (rs != 0xf0) break;
				k += 16;
			} else {
				k += r;
				const int zig = kDezigzag[k++];
				data[zig] = (int16_t)(extend_receive(s) * dq[zig]);
			}
		} while (k < 64);
		return true;
	}
	bool decode_block_prog_dc(int16_t data[64], const Huffman &hdc, int b) {
		if (spec_end != 0) return fail("can't merge dc and ac");
		if (code_bits < 16) grow();
		if (succ_high == 0) {
			memset(data, 0, 64 * sizeof(int16_t));
			const int t = huff_decode(hdc);
			if (t < 0 || t > 15) return fail("can't merge dc and ac");
			const int diff = t ? extend_receive(t) : 0;
			const int dc = comp[b].dc_pred + diff;
			comp[b].dc_pred = dc;
			data[0] = (int16_t)(dc * (1 << succ_low));
		} else if (get_bit()) {
			data[0] = (int16_t)(data[0] + (int16_t)(1 << succ_low));
		}
		return true;
	}
	bool decode_block_prog_ac(int16_t data[64], const Huffman &hac) {
		if (spec_start == 0) return fail("can't merge dc and ac");
		if (succ_high == 0) {
			const int shift = succ_low;
			if (eob_run) { --eob_run; return true; }
			int k = spec_start;
			do {
				const int rs = huff_decode(hac);
				if (rs < 0) return fail("bad huffman code");
				const int s = rs & 15, r = rs >> 4;
				if (s == 0) {
					if (r < 15) {
						eob_run = 1 << r;
						if (r) eob_run += get_bits(r);
						--eob_run;
						break;
					}
					k += 16;
				} else {
					k += r;
					const int zig = kDezigzag[k++];
					data[zig] = (int16_t)(extend_receive(s) * (1 << shift));
				}
			} while (k <= spec_end);
		} else {
			const int16_t bit = (int16_t)(1 << succ_low);
			auto refine = [&](int16_t *q) {
				if (get_bit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
			};
			if (eob_run) {
				--eob_run;
				for (int k = spec_start; k <= spec_end; ++k) {
					int16_t *q = &data[kDezigzag[k]];
					if (*q != 0) refine(q);
				}
			} else {
				int k = spec_start;
				do {
					const int rs = huff_decode(hac);
					if (rs < 0) return fail("bad huffman code");
					int s = rs & 15, r = rs >> 4;
					if (s == 0) {
						if (r < 15) {
							eob_run = (1 << r) - 1;
							if (r) eob_run += get_bits(r);
							r = 64; // force end of block
						}
					} else {
						if (s != 1) return fail("bad huffman code");
						s = get_bit() ? bit : -bit;
					}
					while (k <= spec_end) {
						int16_t *q = &data[kDezigzag[k++]];
						if (*q != 0) {
							refine(q);
						} else {
							if (r == 0) { *q = (int16_t)s; break; }
							--r;
						}
					}
				} while (k <= spec_end);
			}
		}
		return true;
	}

	// ---- inverse DCT (integer, 12 fractional bits)
	static inline uint8_t clamp255(int x) { return (unsigned)x > 255u ? (x < 0 ? 0 : 255) : (uint8_t)x; }
	static void idct_block(uint8_t *out, int stride, const int16_t d[64]) {
#define HMRM_F2F(x) ((int)(((x) * 4096 + 0.5)))
#define HMRM_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                       \
	int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                 \
	p2 = s2; p3 = s6;                                                                        \
	p1 = (p2 + p3) * HMRM_F2F(0.5411961f);                                                   \
	t2 = p1 + p3 * HMRM_F2F(-1.847759065f);                                                  \
	t3 = p1 + p2 * HMRM_F2F(0.765366865f);                                                   \
	p2 = s0; p3 = s4;                                                                        \
	t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                            \
	x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                  \
	t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                      \
	p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                                  \
	p5 = (p3 + p4) * HMRM_F2F(1.175875602f);                                                 \
	t0 = t0 * HMRM_F2F(0.298631336f); t1 = t1 * HMRM_F2F(2.053119869f);                      \
	t2 = t2 * HMRM_F2F(3.072711026f); t3 = t3 * HMRM_F2F(1.501321110f);                      \
	p1 = p5 + p1 * HMRM_F2F(-0.899976223f); p2 = p5 + p2 * HMRM_F2F(-2.562915447f);          \
	p3 = p3 * HMRM_F2F(-1.961570560f); p4 = p4 * HMRM_F2F(-0.390180644f);                    \
	t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
		int val[64];
		for (int i = 0; i < 8; ++i) {
			const int16_t *c = d + i;
			int *v = val + i;
			if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
				const int dcterm = c[0] * 4;
				v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
			} else {
				HMRM_IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
				x0 += 512; x1 += 512; x2 += 512; x3 += 512;
				v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
				v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
				v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
				v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
			}
		}
		for (int i = 0; i < 8; ++i) {
			const int *v = val + i * 8;
			uint8_t *o = out + (size_t)i * stride;
			HMRM_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
			x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17);
			x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
			o[0] = clamp255((x0 + t3) >> 17); o[7] = clamp255((x0 - t3) >> 17);
			o[1] = clamp255((x1 + t2) >> 17); o[6] = clamp255((x1 - t2) >> 17);
			o[2] = clamp255((x2 + t1) >> 17); o[5] = clamp255((x2 - t1) >> 17);
			o[3] = clamp255((x3 + t0) >> 17); o[4] = clamp255((x3 - t0) >> 17);
		}
#undef HMRM_IDCT_1D
#undef HMRM_F2F
	}

	// ---- markers and headers
	int get_marker() {
		if (marker != kMarkerNone) { int x = marker; marker = kMarkerNone; return x; }
		int x = get8();
		if (x != 0xff) return kMarkerNone;
		while (x == 0xff) x = get8();
		return x;
	}
	void reset_entropy() {
		code_bits = 0;
		code_buffer = 0;
		nomore = false;
		for (auto &c : comp) c.dc_pred = 0;
		marker = kMarkerNone;
		todo = restart_interval ? restart_interval : 0x7fffffff;
		eob_run = 0;
	}
	bool process_marker(int m) {
		switch (m) {
		case kMarkerNone: return fail("expected marker");
		case 0xDD:
			if (get16() != 4) return fail("bad DRI len");
			restart_interval = get16();
			return true;
		case 0xDB: {
			int L = get16() - 2;
			while (L > 0) {
				const int q = get8(), prec = q >> 4, t = q & 15;
				if (prec != 0 && prec != 1) return fail("bad DQT type");
				if (t > 3) return fail("bad DQT table");
				for (int i = 0; i < 64; ++i) dequant[t][kDezigzag[i]] = (uint16_t)(prec ? get16() : get8());
				L -= prec ? 129 : 65;
			}
			return L == 0 ? true : fail("bad DQT len");
		}
		case 0xC4: {
			int L = get16() - 2;
			while (L > 0) {
				int sizes[16], n = 0;
				const int q = get8(), tc = q >> 4, th = q & 15;
				if (tc > 1 || th > 3) return fail("bad DHT header");
				for (int i = 0; i < 16; ++i) { sizes[i] = get8(); n += sizes[i]; }
				if (n > 256) return fail("bad DHT header");
				L -= 17;
				Huffman &h = tc == 0 ? huff_dc[th] : huff_ac[th];
				if (!h.build(sizes)) return fail("bad code lengths");
				for (int i = 0; i < n; ++i) h.values[i] = (uint8_t)get8();
				L -= n;
			}
			return L == 0 ? true : fail("bad DHT len");
		}
		default: break;
		}
		if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
			int L = get16();
			if (L < 2) return fail(m == 0xFE ? "bad COM len" : "bad APP len");
			L -= 2;
			if (m == 0xE0 && L >= 5) {
				static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
				bool ok = true;
				for (int i = 0; i < 5; ++i) if (get8() != tag[i]) ok = false;
				L -= 5;
				if (ok) jfif = true;
			} else if (m == 0xEE && L >= 12) {
				static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
				bool ok = true;
				for (int i = 0; i < 6; ++i) if (get8() != tag[i]) ok = false;
				L -= 6;
				if (ok) {
					get8(); get16(); get16();
					app14 = get8();
					L -= 6;
				}
			}
			skip(L);
			return true;
		}
		return fail("unknown marker");
	}
	bool process_frame_header() {
		const int Lf = get16();
		if (Lf < 11) return fail("bad SOF len");
		if (get8() != 8) return fail("only 8-bit");
		img_y = get16();
		if (img_y == 0) return fail("no header height");
		img_x = get16();
		if (img_x == 0) return fail("0 width");
		const int c = get8();
		if (c != 3 && c != 1 && c != 4) return fail("bad component count");
		img_n = c;
		if (Lf != 8 + 3 * img_n) return fail("bad SOF len");
		rgb = 0;
		for (int i = 0; i < img_n; ++i) {
			static const uint8_t rgb_ids[3] = {'R', 'G', 'B'};
			comp[i].id = get8();
			if (img_n == 3 && comp[i].id == rgb_ids[i]) ++rgb;
			const int q = get8();
			comp[i].h = q >> 4;
			comp[i].v = q & 15;
			if (!comp[i].h || comp[i].h > 4) return fail("bad H");
			if (!comp[i].v || comp[i].v > 4) return fail("bad V");
			comp[i].tq = get8();
			if (comp[i].tq > 3) return fail("bad TQ");
		}
		if ((int64_t)img_x * img_y * img_n > ((int64_t)1 << 30)) return fail("too large");
		h_max = v_max = 1;
		for (int i = 0; i < img_n; ++i) {
			if (comp[i].h > h_max) h_max = comp[i].h;
			if (comp[i].v > v_max) v_max = comp[i].v;
		}
		for (int i = 0; i < img_n; ++i) {
			if (h_max % comp[i].h != 0) return fail("bad H");
			if (v_max % comp[i].v != 0) return fail("bad V");
		}
		const int mcu_w = h_max * 8, mcu_h = v_max * 8;
		mcu_x = (img_x + mcu_w - 1) / mcu_w;
		mcu_y = (img_y + mcu_h - 1) / mcu_h;
		for (int i = 0; i < img_n; ++i) {
			Component &k = comp[i];
			k.x = (img_x * k.h + h_max - 1) / h_max;
			k.y = (img_y * k.v + v_max - 1) / v_max;
			k.w2 = mcu_x * k.h * 8;
			k.h2 = mcu_y * k.v * 8;
			k.data.assign((size_t)k.w2 * k.h2, 0);
			if (progressive) {
				k.coeff_w = k.w2 / 8;
				k.coeff.assign((size_t)k.w2 * k.h2, 0);
			}
		}
		return true;
	}
	bool decode_header() {
		jfif = false;
		app14 = -1;
		marker = kMarkerNone;
		int m = get_marker();
		if (m != 0xd8) return fail("no SOI");
		m = get_marker();
		while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
			if (!process_marker(m)) return false;
			m = get_marker();
			while (m == kMarkerNone) {
				if (at_eof()) return fail("no SOF");
				m = get_marker();
			}
		}
		progressive = (m == 0xc2);
		return process_frame_header();
	}
	bool process_scan_header() {
		const int Ls = get16();
		scan_n = get8();
		if (scan_n < 1 || scan_n > 4 || scan_n > img_n) return fail("bad SOS component count");
		if (Ls != 6 + 2 * scan_n) return fail("bad SOS len");
		for (int i = 0; i < scan_n; ++i) {
			const int id = get8(), q = get8();
			int which = 0;
			for (; which < img_n; ++which) if (comp[which].id == id) break;
			if (which == img_n) return fail("bad SOS component");
			comp[which].hd = q >> 4;
			comp[which].ha = q & 15;
			if (comp[which].hd > 3) return fail("bad DC huff");
			if (comp[which].ha > 3) return fail("bad AC huff");
			order[i] = which;
		}
		spec_start = get8();
		spec_end = get8();
		const int aa = get8();
		succ_high = aa >> 4;
		succ_low = aa & 15;
		if (progressive) {
			if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13)
				return fail("bad SOS");
		} else {
			if (spec_start != 0 || succ_high != 0 || succ_low != 0) return fail("bad SOS");
			spec_end = 63;
		}
		return true;
	}
	// returns false on hard error; *stop set when a non-restart marker ends the scan early
	bool restart_check(bool *stop) {
		if (--todo <= 0) {
			if (code_bits < 24) grow();
			if (!(marker >= 0xd0 && marker <= 0xd7)) { *stop = true; return true; }
			reset_entropy();
		}
		return true;
	}
	bool parse_entropy_coded_data() {
		reset_entropy();
		bool stop = false;
		int16_t block[64];
		if (scan_n == 1) {
			const int n = order[0];
			Component &c = comp[n];
			const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
			for (int j = 0; j < h; ++j)
				for (int i = 0; i < w; ++i) {
					if (!progressive) {
						if (!decode_block(block, huff_dc[c.hd], huff_ac[c.ha], n, dequant[c.tq])) return false;
						idct_block(&c.data[(size_t)c.w2 * j * 8 + i * 8], c.w2, block);
					} else {
						int16_t *data = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
						if (spec_start == 0) {
							if (!decode_block_prog_dc(data, huff_dc[c.hd], n)) return false;
						} else if (!decode_block_prog_ac(data, huff_ac[c.ha])) return false;
					}
					if (!restart_check(&stop)) return false;
					if (stop) return true;
				}
			return true;
		}
		for (int j = 0; j < mcu_y; ++j)
			for (int i = 0; i < mcu_x; ++i) {
				for (int k = 0; k < scan_n; ++k) {
					const int n = order[k];
					Component &c = comp[n];
					for (int y = 0; y < c.v; ++y)
						for (int x = 0; x < c.h; ++x) {
							const int bx = i * c.h + x, by = j * c.v + y;
							if (!progressive) {
								if (!decode_block(block, huff_dc[c.hd], huff_ac[c.ha], n, dequant[c.tq])) return false;
								idct_block(&c.data[(size_t)c.w2 * by * 8 + bx * 8], c.w2, block);
							} else {
								int16_t *data = &c.coeff[64 * ((size_t)bx + (size_t)by * c.coeff_w)];
								if (!decode_block_prog_dc(data, huff_dc[c.hd], n)) return false;
							}
						}
				}
				if (!restart_check(&stop)) return false;
				if (stop) return true;
			}
		return true;
	}
	void finish_progressive() {
		for (int n = 0; n < img_n; ++n) {
			Component &c = comp[n];
			const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
			for (int j = 0; j < h; ++j)
				for (int i = 0; i < w; ++i) {
					int16_t *data = &c.coeff[64 * ((size_t)i + (size_t)j * c.coeff_w)];
					for (int q = 0; q < 64; ++q) data[q] = (int16_t)(data[q] * dequant[c.tq][q]);
					idct_block(&c.data[(size_t)c.w2 * j * 8 + i * 8], c.w2, data);
				}
		}
	}
	bool decode_image() {
		restart_interval = 0;
		if (!decode_header()) return false;
		int m = get_marker();
		while (m != 0xd9) {
			if (m == 0xda) {
				if (!process_scan_header()) return false;
				if (!parse_entropy_coded_data()) return false;
				if (marker == kMarkerNone) {
					while (!at_eof()) { // zeros after the scan data
						if (get8() == 255) { marker = get8(); break; }
					}
				}
			} else if (m == 0xdc) {
				const int Ld = get16(), NL = get16();
				if (Ld != 4) return fail("bad DNL len");
				if (NL != img_y) return fail("bad DNL height");
			} else if (!process_marker(m)) {
				return false;
			}
			m = get_marker();
		}
		if (progressive) finish_progressive();
		return true;
	}
};

// ---- chroma upsampling of one row; `near` is the sample row closer to the output row
inline uint8_t div4(int x) { return (uint8_t)(x >> 2); }
inline uint8_t div16(int x) { return (uint8_t)(x >> 4); }

const uint8_t *resample_row(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int hs, int vs) {
	if (hs == 1 && vs == 1) return near;
	if (hs == 1 && vs == 2) {
		for (int i = 0; i < w; ++i) out[i] = div4(3 * near[i] + far[i] + 2);
		return out;
	}
	if (hs == 2 && vs == 1) {
		if (w == 1) { out[0] = out[1] = near[0]; return out; }
		out[0] = near[0];
		out[1] = div4(near[0] * 3 + near[1] + 2);
		int i;
		for (i = 1; i < w - 1; ++i) {
			const int n = 3 * near[i] + 2;
			out[i * 2 + 0] = div4(n + near[i - 1]);
			out[i * 2 + 1] = div4(n + near[i + 1]);
		}
		out[i * 2 + 0] = div4(near[w - 2] * 3 + near[w - 1] + 2);
		out[i * 2 + 1] = near[w - 1];
		return out;
	}
	if (hs == 2 && vs == 2) {
		if (w == 1) { out[0] = out[1] = div4(3 * near[0] + far[0] + 2); return out; }
		int t1 = 3 * near[0] + far[0];
		out[0] = div4(t1 + 2);
		for (int i = 1; i < w; ++i) {
			const int t0 = t1;
			t1 = 3 * near[i] + far[i];
			out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
			out[i * 2] = div16(3 * t1 + t0 + 8);
		}
		out[w * 2 - 1] = div4(t1 + 2);
		return out;
	}
	for (int i = 0; i < w; ++i)
		for (int j = 0; j < hs; ++j) out[i * hs + j] = near[i];
	return out;
}

inline void ycbcr_to_rgb(uint8_t *out, int y, int cb_, int cr_) {
#define HMRM_F2FIX(x) (((int)((x)*4096.0f + 0.5f)) << 8)
	const int y_fixed = (y << 20) + (1 << 19);
	const int cr = cr_ - 128, cb = cb_ - 128;
	int r = y_fixed + cr * HMRM_F2FIX(1.40200f);
	int g = y_fixed + (cr * -HMRM_F2FIX(0.71414f)) + ((cb * -HMRM_F2FIX(0.34414f)) & 0xffff0000);
	int b = y_fixed + cb * HMRM_F2FIX(1.77200f);
#undef HMRM_F2FIX
	r >>= 20; g >>= 20; b >>= 20;
	if ((unsigned)r > 255) r = r < 0 ? 0 : 255;
	if ((unsigned)g > 255) g = g < 0 ? 0 : 255;
	if ((unsigned)b > 255) b = b < 0 ? 0 : 255;
	out[0] = (uint8_t)r; out[1] = (uint8_t)g; out[2] = (uint8_t)b;
}
inline uint8_t blinn(uint8_t x, uint8_t y) {
	const unsigned t = (unsigned)x * y + 128;
	return (uint8_t)((t + (t >> 8)) >> 8);
}
inline uint8_t luma(int r, int g, int b) { return (uint8_t)(((r * 77) + (g * 150) + (29 * b)) >> 8); }

} // namespace

bool decode_jpeg(const uint8_t *bytes, size_t len, int req_comp, Image *img, std::string *err) {
	std::vector<Decoder> holder(1); // the decoder is large: keep it off the stack
	Decoder &z = holder[0];
	z.p = bytes;
	z.end = bytes + len;
	memset(z.dequant, 0, sizeof z.dequant);
	for (auto &h : z.huff_dc) { memset(&h, 0, sizeof h); }
	for (auto &h : z.huff_ac) { memset(&h, 0, sizeof h); }
	if (!z.decode_image()) {
		*err = z.err.empty() ? "Corrupt JPEG" : z.err;
		return false;
	}
	const int n = req_comp ? req_comp : (z.img_n >= 3 ? 3 : 1);
	const bool is_rgb = z.img_n == 3 && (z.rgb == 3 || (z.app14 == 0 && !z.jfif));
	const int decode_n = (z.img_n == 3 && n < 3 && !is_rgb) ? 1 : z.img_n;
	const int W = z.img_x, H = z.img_y;

	struct Resample { int hs, vs, ystep, w_lores, ypos; size_t line0, line1; };
	Resample rs[4];
	std::vector<uint8_t> linebuf[4];
	for (int k = 0; k < decode_n; ++k) {
		linebuf[k].assign((size_t)W + 3 + 8, 0);
		rs[k].hs = z.h_max / z.comp[k].h;
		rs[k].vs = z.v_max / z.comp[k].v;
		rs[k].ystep = rs[k].vs >> 1;
		rs[k].w_lores = (W + rs[k].hs - 1) / rs[k].hs;
		rs[k].ypos = 0;
		rs[k].line0 = rs[k].line1 = 0;
	}
	img->w = W;
	img->h = H;
	img->comp = n;
	img->comp_in_file = z.img_n >= 3 ? 3 : 1;
	img->px.assign((size_t)n * W * H, 0);
	const uint8_t *co[4] = {nullptr, nullptr, nullptr, nullptr};
	for (int j = 0; j < H; ++j) {
		uint8_t *out = &img->px[(size_t)n * W * j];
		for (int k = 0; k < decode_n; ++k) {
			Resample &r = rs[k];
			const Component &c = z.comp[k];
			const bool y_bot = r.ystep >= (r.vs >> 1);
			const uint8_t *l0 = c.data.data() + r.line0, *l1 = c.data.data() + r.line1;
			co[k] = resample_row(linebuf[k].data(), y_bot ? l1 : l0, y_bot ? l0 : l1, r.w_lores, r.hs, r.vs);
			if (++r.ystep >= r.vs) {
				r.ystep = 0;
				r.line0 = r.line1;
				if (++r.ypos < c.y) r.line1 += (size_t)c.w2;
			}
		}
		if (n >= 3) {
			if (z.img_n == 3) {
				for (int i = 0; i < W; ++i, out += n) {
					if (is_rgb) { out[0] = co[0][i]; out[1] = co[1][i]; out[2] = co[2][i]; }
					else ycbcr_to_rgb(out, co[0][i], co[1][i], co[2][i]);
					if (n == 4) out[3] = 255;
				}
			} else if (z.img_n == 4) {
				for (int i = 0; i < W; ++i, out += n) {
					const uint8_t m = co[3][i];
					if (z.app14 == 0) { // CMYK
						out[0] = blinn(co[0][i], m); out[1] = blinn(co[1][i], m); out[2] = blinn(co[2][i], m);
					} else {
						ycbcr_to_rgb(out, co[0][i], co[1][i], co[2][i]);
						if (z.app14 == 2) { // YCCK
							out[0] = blinn((uint8_t)(255 - out[0]), m);
							out[1] = blinn((uint8_t)(255 - out[1]), m);
							out[2] = blinn((uint8_t)(255 - out[2]), m);
						}
					}
					if (n == 4) out[3] = 255;
				}
			} else {
				for (int i = 0; i < W; ++i, out += n) {
					out[0] = out[1] = out[2] = co[0][i];
					if (n == 4) out[3] = 255;
				}
			}
		} else {
			for (int i = 0; i < W; ++i, out += n) {
				if (is_rgb) out[0] = luma(co[0][i], co[1][i], co[2][i]);
				else if (z.img_n == 4 && z.app14 == 0)
					out[0] = luma(blinn(co[0][i], co[3][i]), blinn(co[1][i], co[3][i]), blinn(co[2][i], co[3][i]));
				else if (z.img_n == 4 && z.app14 == 2) out[0] = blinn((uint8_t)(255 - co[0][i]), co[3][i]);
				else out[0] = co[0][i];
				if (n == 2) out[1] = 255;
			}
		}
	}
	return true;
}

} // namespace hmrm
