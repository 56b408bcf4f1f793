// jpeg_decode.cpp -- JPEG (ITU-T T.81) decoder for height / colour maps: baseline and extended
// sequential Huffman (SOF0/SOF1) and progressive Huffman (SOF2), 8 bits per sample, 1, 3 or 4
// components, restart intervals, JFIF / Adobe colour signalling.
//
// Written for this project from the standard (T.81 Annex B marker syntax, Annex F/G entropy coding,
// Annex A.3.3 IDCT definition) with its own structure: the file is first cut into marker segments,
// every scan's entropy-coded bytes are un-stuffed and split at the restart markers up front, the
// coefficients of every scan type land in one coefficient store per component, and one
// reconstruction pass (dequantise, IDCT, upsample, colour) runs at the end.
//
// ONE thing is not free: the reference loads its maps with stb_image v2.27 (main/hmap.cpp:320-321,
// 341-342), and a JPEG file does not define its decoded pixels to the last bit -- the IDCT's
// fixed-point arithmetic, the chroma upsampling filter and the YCbCr matrix's rounding are the
// decoder's choice.  Heights come from these pixels, so the numerical choices below are stb_image's,
// restated as arithmetic (each marked "pixel contract"); tests/golden/jpeg_decode.npz holds what the
// reference's build of stb produces and tests/test_image_io.py compares with it, also live.
#include <cstring>
#include <string>
#include <vector>

#include "image_io.hpp"

namespace hmrm {
namespace {

// Zig-zag position -> natural (row-major) index, T.81 Figure A.6.
const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Fail {
	std::string why;
};

// ------------------------------------------------------------------ bits ----
// MSB-first reader over un-stuffed entropy-coded bytes.  Past the end it delivers zero bits (a
// truncated interval then decodes as whatever zeros mean, as common decoders do) and counts them.
class BitSource {
public:
	BitSource(const uint8_t *p, size_t n) : p_(p), n_(n) {}
	inline void fill() {
		while (have_ <= 56) {
			const uint64_t b = pos_ < n_ ? p_[pos_] : 0;
			++pos_;
			acc_ |= b << (56 - have_);
			have_ += 8;
		}
	}
	inline uint32_t peek(int n) { // n in 1..16
		if (have_ < n) fill();
		return (uint32_t)(acc_ >> (64 - n));
	}
	inline void drop(int n) {
		acc_ <<= n;
		have_ -= n;
	}
	inline uint32_t take(int n) {
		if (n == 0) return 0;
		const uint32_t v = peek(n);
		drop(n);
		return v;
	}

private:
	const uint8_t *p_;
	size_t n_, pos_ = 0;
	uint64_t acc_ = 0;
	int have_ = 0;
};

// --------------------------------------------------------------- Huffman ----
// Canonical code from BITS/HUFFVAL (T.81 Annex C): a direct table for codes of up to kQuick bits,
// and per-length first-code / first-index bounds (Figure F.16's MINCODE / VALPTR) for the rest.
class HuffmanTable {
public:
	bool defined = false;
	void build(const uint8_t counts[16], const uint8_t *symbols, int nsym) {
		std::memset(quick_, 0xff, sizeof quick_);
		std::memcpy(sym_, symbols, (size_t)nsym);
		uint32_t code = 0;
		int index = 0;
		for (int len = 1; len <= 16; ++len) {
			first_code_[len] = code;
			first_index_[len] = index;
			for (int k = 0; k < counts[len - 1]; ++k, ++index, ++code) {
				if (code >= (1u << len)) throw Fail{"bad code lengths"};
				if (len <= kQuick) {
					const uint32_t lo = code << (kQuick - len), span = 1u << (kQuick - len);
					for (uint32_t f = 0; f < span; ++f) quick_[lo + f] = (uint16_t)((len << 8) | symbols[index]);
				}
			}
			end_code_[len] = code; // one past the last code of this length
			code <<= 1;
		}
		defined = true;
	}
	inline int decode(BitSource &bits) const {
		const uint32_t look = bits.peek(16);
		const uint16_t q = quick_[look >> (16 - kQuick)];
		if (q != 0xffff) {
			bits.drop(q >> 8);
			return q & 0xff;
		}
		for (int len = kQuick + 1; len <= 16; ++len) {
			const uint32_t c = look >> (16 - len);
			if (c < end_code_[len] && c >= first_code_[len]) {
				bits.drop(len);
				return sym_[first_index_[len] + (int)(c - first_code_[len])];
			}
		}
		throw Fail{"bad huffman code"};
	}

private:
	static constexpr int kQuick = 9;
	uint16_t quick_[1 << kQuick];
	uint8_t sym_[256];
	uint32_t first_code_[17], end_code_[17];
	int first_index_[17];
};

// T.81 F.2.2.1 EXTEND: the s-bit magnitude category value -> signed coefficient.
inline int extend(uint32_t v, int s) { return s == 0 ? 0 : (v < (1u << (s - 1)) ? (int)v - (int)((1u << s) - 1) : (int)v); }

// ----------------------------------------------------------------- frame ----
struct Plane {
	int id = 0, hs = 1, vs = 1, tq = 0;
	int width = 0, height = 0;           // samples of this component (ceil of the scaled frame size)
	int blocks_w = 0, blocks_h = 0;      // 8x8 blocks covering width x height
	int store_w = 0, store_h = 0;        // blocks in the coefficient store (padded to whole MCUs)
	std::vector<int16_t> coef;           // store_w * store_h * 64, natural order, NOT dequantised
	std::vector<uint8_t> samples;        // store_w*8 x store_h*8 after reconstruction
	int dc_pred = 0;
	int16_t *block(int bx, int by) { return &coef[((size_t)by * store_w + bx) * 64]; }
};

struct Scan {
	int ncomp = 0;
	int plane[4] = {0, 0, 0, 0};
	int dc_table[4] = {0, 0, 0, 0}, ac_table[4] = {0, 0, 0, 0};
	int ss = 0, se = 63, ah = 0, al = 0;
};

class JpegDecoder {
public:
	JpegDecoder(const uint8_t *data, size_t len) : d_(data), n_(len) {}

	void decode(int req_comp, Image *out) {
		parse();
		reconstruct();
		emit(req_comp, out);
	}

private:
	const uint8_t *d_;
	size_t n_, at_ = 0;
	// tables
	uint16_t quant_[4][64] = {}; // natural order
	HuffmanTable dc_[4], ac_[4];
	// frame
	bool have_frame_ = false, progressive_ = false;
	int width_ = 0, height_ = 0, ncomp_ = 0, hmax_ = 1, vmax_ = 1, mcus_w_ = 0, mcus_h_ = 0;
	Plane planes_[4];
	int restart_interval_ = 0;
	bool jfif_ = false;
	int adobe_transform_ = -1;
	int rgb_ids_ = 0; // components whose ids spell 'R','G','B'
	bool saw_scan_ = false;

	// ---- byte access ----
	uint8_t u8() {
		if (at_ >= n_) throw Fail{"truncated"};
		return d_[at_++];
	}
	int u16() {
		const int hi = u8();
		return (hi << 8) | u8();
	}
	// next marker code; fill bytes (0xff repeated) are allowed in front of it
	int next_marker() {
		if (u8() != 0xff) throw Fail{"expected marker"};
		int m = u8();
		while (m == 0xff) m = u8();
		return m;
	}

	// ---- container ----
	void parse() {
		if (n_ < 4 || d_[0] != 0xff || d_[1] != 0xd8) throw Fail{"no SOI"};
		at_ = 2;
		for (;;) {
			const int m = next_marker();
			if (m == 0xd9) { // EOI
				if (!have_frame_ || !saw_scan_) throw Fail{"no image data"};
				return;
			}
			if (m == 0xda) {
				scan_segment();
				continue;
			}
			if (m == 0xc0 || m == 0xc1 || m == 0xc2) {
				frame_segment(m == 0xc2);
				continue;
			}
			if ((m >= 0xc3 && m <= 0xcf && m != 0xc4) || m == 0x01 || (m >= 0xd0 && m <= 0xd7) || m == 0x00)
				throw Fail{m >= 0xc3 && m <= 0xcf ? "unsupported JPEG process (lossless / arithmetic / hierarchical)"
				                                  : "unexpected marker"};
			const int len = u16();
			if (len < 2 || at_ + (size_t)(len - 2) > n_) throw Fail{"bad segment length"};
			const size_t end = at_ + (size_t)(len - 2);
			switch (m) {
			case 0xdb: quant_segment(end); break;
			case 0xc4: huffman_segment(end); break;
			case 0xdd:
				if (len != 4) throw Fail{"bad DRI length"};
				restart_interval_ = u16();
				break;
			case 0xdc: { // DNL: only legal after the first scan, and must repeat the frame's height
				if (len != 4) throw Fail{"bad DNL length"};
				if (u16() != height_) throw Fail{"bad DNL height"};
				break;
			}
			case 0xe0:
				if (len >= 7 && std::memcmp(d_ + at_, "JFIF\0", 5) == 0) jfif_ = true;
				break;
			case 0xee:
				if (len >= 14 && std::memcmp(d_ + at_, "Adobe\0", 6) == 0) adobe_transform_ = d_[at_ + 11];
				break;
			default:
				if (!((m >= 0xe0 && m <= 0xef) || m == 0xfe)) throw Fail{"unknown marker"};
			}
			at_ = end;
		}
	}

	void quant_segment(size_t end) {
		while (at_ < end) {
			const int pq_tq = u8();
			const int precision = pq_tq >> 4, slot = pq_tq & 15;
			if (precision > 1) throw Fail{"bad DQT precision"};
			if (slot > 3) throw Fail{"bad DQT table"};
			for (int k = 0; k < 64; ++k) quant_[slot][kNatural[k]] = (uint16_t)(precision ? u16() : u8());
		}
		if (at_ != end) throw Fail{"bad DQT length"};
	}

	void huffman_segment(size_t end) {
		while (at_ < end) {
			const int tc_th = u8();
			const int cls = tc_th >> 4, slot = tc_th & 15;
			if (cls > 1 || slot > 3) throw Fail{"bad DHT header"};
			uint8_t counts[16];
			int total = 0;
			for (int k = 0; k < 16; ++k) total += counts[k] = u8();
			if (total > 256) throw Fail{"bad DHT header"};
			uint8_t symbols[256];
			for (int k = 0; k < total; ++k) symbols[k] = u8();
			(cls == 0 ? dc_ : ac_)[slot].build(counts, symbols, total);
		}
		if (at_ != end) throw Fail{"bad DHT length"};
	}

	void frame_segment(bool progressive) {
		if (have_frame_) throw Fail{"multiple frames"};
		const int len = u16();
		if (len < 11) throw Fail{"bad SOF length"};
		if (u8() != 8) throw Fail{"only 8-bit samples"};
		height_ = u16();
		width_ = u16();
		if (height_ == 0) throw Fail{"no header height"}; // (height by DNL is refused, as the reference's loader does)
		if (width_ == 0) throw Fail{"0 width"};
		ncomp_ = u8();
		if (ncomp_ != 1 && ncomp_ != 3 && ncomp_ != 4) throw Fail{"bad component count"};
		if (len != 8 + 3 * ncomp_) throw Fail{"bad SOF length"};
		if ((int64_t)width_ * height_ > ((int64_t)1 << 28)) throw Fail{"too large"};
		static const char rgb[3] = {'R', 'G', 'B'};
		for (int c = 0; c < ncomp_; ++c) {
			Plane &p = planes_[c];
			p.id = u8();
			if (ncomp_ == 3 && p.id == rgb[c]) ++rgb_ids_;
			const int hv = u8();
			p.hs = hv >> 4;
			p.vs = hv & 15;
			if (p.hs < 1 || p.hs > 4 || p.vs < 1 || p.vs > 4) throw Fail{"bad sampling factor"};
			p.tq = u8();
			if (p.tq > 3) throw Fail{"bad quantisation table index"};
			hmax_ = p.hs > hmax_ ? p.hs : hmax_;
			vmax_ = p.vs > vmax_ ? p.vs : vmax_;
		}
		for (int c = 0; c < ncomp_; ++c) // (the upsampler below only scales by whole factors)
			if (hmax_ % planes_[c].hs != 0 || vmax_ % planes_[c].vs != 0) throw Fail{"bad sampling factor"};
		mcus_w_ = (width_ + 8 * hmax_ - 1) / (8 * hmax_);
		mcus_h_ = (height_ + 8 * vmax_ - 1) / (8 * vmax_);
		for (int c = 0; c < ncomp_; ++c) {
			Plane &p = planes_[c];
			p.width = (width_ * p.hs + hmax_ - 1) / hmax_;
			p.height = (height_ * p.vs + vmax_ - 1) / vmax_;
			p.blocks_w = (p.width + 7) / 8;
			p.blocks_h = (p.height + 7) / 8;
			p.store_w = mcus_w_ * p.hs;
			p.store_h = mcus_h_ * p.vs;
			p.coef.assign((size_t)p.store_w * p.store_h * 64, 0);
		}
		progressive_ = progressive;
		have_frame_ = true;
	}

	// ---- scans ----
	void scan_segment() {
		if (!have_frame_) throw Fail{"scan before frame"};
		const int len = u16();
		Scan sc;
		sc.ncomp = u8();
		if (sc.ncomp < 1 || sc.ncomp > ncomp_) throw Fail{"bad SOS component count"};
		if (len != 6 + 2 * sc.ncomp) throw Fail{"bad SOS length"};
		for (int k = 0; k < sc.ncomp; ++k) {
			const int id = u8(), tables = u8();
			int which = -1;
			for (int c = 0; c < ncomp_; ++c)
				if (planes_[c].id == id) {
					which = c;
					break;
				}
			if (which < 0) throw Fail{"scan names an unknown component"};
			sc.plane[k] = which;
			sc.dc_table[k] = tables >> 4;
			sc.ac_table[k] = tables & 15;
			if (sc.dc_table[k] > 3 || sc.ac_table[k] > 3) throw Fail{"bad huffman table index"};
		}
		sc.ss = u8();
		sc.se = u8();
		const int a = u8();
		sc.ah = a >> 4;
		sc.al = a & 15;
		if (progressive_) {
			if (sc.ss > 63 || sc.se > 63 || sc.ss > sc.se || sc.ah > 13 || sc.al > 13) throw Fail{"bad SOS"};
			if (sc.ss == 0 && sc.se != 0) throw Fail{"DC and AC in one progressive scan"};
			if (sc.ss > 0 && sc.ncomp != 1) throw Fail{"interleaved AC scan"};
		} else {
			if (sc.ss != 0 || sc.ah != 0 || sc.al != 0) throw Fail{"bad SOS"};
			sc.se = 63;
		}

		// The entropy-coded segment runs up to the next marker that is neither a stuffed 0xff00 nor a
		// restart marker.  Un-stuff it once, remembering where each restart interval begins.
		std::vector<uint8_t> bytes;
		std::vector<size_t> starts(1, 0);
		size_t i = at_;
		for (; i < n_; ++i) {
			const uint8_t b = d_[i];
			if (b != 0xff) {
				bytes.push_back(b);
				continue;
			}
			if (i + 1 >= n_) { // a lone 0xff at the very end: nothing follows, the marker search below fails
				i = n_;
				break;
			}
			const uint8_t nxt = d_[i + 1];
			if (nxt == 0x00) {
				bytes.push_back(0xff);
				++i;
			} else if (nxt >= 0xd0 && nxt <= 0xd7) {
				starts.push_back(bytes.size());
				++i;
			} else if (nxt == 0xff) {
				continue; // fill byte in front of a marker
			} else {
				break; // a real marker: the segment ends in front of it
			}
		}
		at_ = i;
		starts.push_back(bytes.size());
		run_scan(sc, bytes, starts);
		saw_scan_ = true;
	}

	void run_scan(const Scan &sc, const std::vector<uint8_t> &bytes, const std::vector<size_t> &starts) {
		for (int k = 0; k < sc.ncomp; ++k) {
			const bool need_dc = sc.ss == 0 && sc.ah == 0, need_ac = sc.se > 0;
			if (need_dc && !dc_[sc.dc_table[k]].defined) throw Fail{"missing DC huffman table"};
			if (need_ac && !ac_[sc.ac_table[k]].defined) throw Fail{"missing AC huffman table"};
		}
		// units of the scan: MCUs when interleaved, the component's own blocks otherwise (T.81 A.2)
		const bool interleaved = sc.ncomp > 1;
		Plane &solo = planes_[sc.plane[0]];
		const int units_w = interleaved ? mcus_w_ : solo.blocks_w, units_h = interleaved ? mcus_h_ : solo.blocks_h;
		const int64_t total = (int64_t)units_w * units_h;
		const int64_t per_interval = restart_interval_ > 0 ? restart_interval_ : total;
		int64_t unit = 0;
		for (size_t iv = 0; iv + 1 < starts.size() && unit < total; ++iv) {
			BitSource bits(bytes.data() + starts[iv], starts[iv + 1] - starts[iv]);
			for (int c = 0; c < ncomp_; ++c) planes_[c].dc_pred = 0;
			int eob_run = 0;
			for (int64_t k = 0; k < per_interval && unit < total; ++k, ++unit) {
				const int ux = (int)(unit % units_w), uy = (int)(unit / units_w);
				if (!interleaved) {
					one_block(sc, 0, solo.block(ux, uy), bits, eob_run);
					continue;
				}
				for (int s = 0; s < sc.ncomp; ++s) {
					Plane &p = planes_[sc.plane[s]];
					for (int by = 0; by < p.vs; ++by)
						for (int bx = 0; bx < p.hs; ++bx) one_block(sc, s, p.block(ux * p.hs + bx, uy * p.vs + by), bits, eob_run);
				}
			}
		}
		// (fewer intervals than the frame needs: the blocks not reached keep their zero coefficients)
	}

	void one_block(const Scan &sc, int s, int16_t *blk, BitSource &bits, int &eob_run) {
		Plane &p = planes_[sc.plane[s]];
		if (!progressive_) {
			sequential_block(p, dc_[sc.dc_table[s]], ac_[sc.ac_table[s]], blk, bits);
		} else if (sc.ss == 0) {
			if (sc.ah == 0) {
				const int t = dc_[sc.dc_table[s]].decode(bits);
				if (t > 15) throw Fail{"bad huffman code"};
				p.dc_pred += extend(bits.take(t), t);
				blk[0] = (int16_t)(p.dc_pred * (1 << sc.al)); // (wraps like a 16-bit store on damaged streams)
			} else if (bits.take(1)) {
				blk[0] = (int16_t)(blk[0] + (1 << sc.al));
			}
		} else if (sc.ah == 0) {
			ac_first(sc, ac_[sc.ac_table[s]], blk, bits, eob_run);
		} else {
			ac_refine(sc, ac_[sc.ac_table[s]], blk, bits, eob_run);
		}
	}

	// T.81 F.2.2: DC difference, then run/size pairs up to EOB.
	void sequential_block(Plane &p, const HuffmanTable &dc, const HuffmanTable &ac, int16_t *blk, BitSource &bits) {
		const int t = dc.decode(bits);
		if (t > 15) throw Fail{"bad huffman code"};
		p.dc_pred += extend(bits.take(t), t);
		blk[0] = (int16_t)p.dc_pred;
		for (int k = 1; k < 64;) {
			const int rs = ac.decode(bits), run = rs >> 4, size = rs & 15;
			if (size == 0) {
				if (run != 15) break; // EOB
				k += 16;              // ZRL
				continue;
			}
			k += run;
			// (a run past the block's end is a damaged stream: the value lands on the last coefficient)
			blk[k < 64 ? kNatural[k] : 63] = (int16_t)extend(bits.take(size), size);
			++k;
		}
	}

	// T.81 G.1.2.2: first pass over a spectral band, with end-of-band runs.
	void ac_first(const Scan &sc, const HuffmanTable &ac, int16_t *blk, BitSource &bits, int &eob_run) {
		if (eob_run > 0) {
			--eob_run;
			return;
		}
		for (int k = sc.ss; k <= sc.se;) {
			const int rs = ac.decode(bits), run = rs >> 4, size = rs & 15;
			if (size == 0) {
				if (run < 15) { // EOBn: this block and 2^run - 1 + extra more end here
					eob_run = (1 << run) - 1;
					if (run) eob_run += (int)bits.take(run);
					return;
				}
				k += 16;
				continue;
			}
			k += run;
			blk[k < 64 ? kNatural[k] : 63] = (int16_t)(extend(bits.take(size), size) * (1 << sc.al));
			++k;
		}
	}

	// T.81 G.1.2.3: refinement of a band -- one correction bit for every coefficient that is already
	// non-zero, new +-1 coefficients placed after `run` still-zero positions.
	void ac_refine(const Scan &sc, const HuffmanTable &ac, int16_t *blk, BitSource &bits, int &eob_run) {
		const int plus = 1 << sc.al, minus = -plus;
		auto correct = [&](int16_t &c) {
			if (bits.take(1) && (c & plus) == 0) c = (int16_t)(c + (c > 0 ? plus : minus));
		};
		int k = sc.ss;
		if (eob_run == 0) {
			while (k <= sc.se) {
				const int rs = ac.decode(bits), size = rs & 15;
				int run = rs >> 4, value = 0;
				if (size == 0) {
					if (run < 15) {
						eob_run = (1 << run) - 1;
						if (run) eob_run += (int)bits.take(run);
						++eob_run; // (counts this block too: handled by the tail below)
						break;
					}
					// ZRL: 16 zero positions, i.e. run = 15 and a zero "new" coefficient
				} else {
					if (size != 1) throw Fail{"bad huffman code"};
					value = bits.take(1) ? plus : minus;
				}
				while (k <= sc.se) {
					int16_t &c = blk[kNatural[k++]];
					if (c != 0) {
						correct(c);
					} else if (run == 0) {
						c = (int16_t)value;
						break;
					} else {
						--run;
					}
				}
			}
		}
		if (eob_run > 0) {
			for (; k <= sc.se; ++k) {
				int16_t &c = blk[kNatural[k]];
				if (c != 0) correct(c);
			}
			--eob_run;
		}
	}

	// ---- reconstruction ----
	// PIXEL CONTRACT (stb_image v2.27): the 8x8 inverse DCT in 32-bit fixed point -- an even/odd
	// factorisation with 12-bit constants, columns first keeping two extra bits (>> 10 after + 512),
	// then rows (>> 17 after + 65536 + (128 << 17), i.e. rounding and the +128 level shift), clamped to
	// 0..255; a column whose AC terms are all zero is just its DC term * 4.  The constants are computed
	// the way that decoder spells them, (int)(c * 4096 + 0.5) on a float literal: negative ones
	// therefore round towards zero.
	static constexpr int fix(float c) { return (int)(c * 4096 + 0.5); }
	struct Odd {
		int a, b, c, d;
	};
	static inline void even_part(int s0, int s2, int s4, int s6, int e[4]) {
		const int z = (s2 + s6) * fix(0.5411961f);
		const int lo = z + s6 * fix(-1.847759065f), hi = z + s2 * fix(0.765366865f);
		const int sum = (s0 + s4) * 4096, diff = (s0 - s4) * 4096;
		e[0] = sum + hi;
		e[3] = sum - hi;
		e[1] = diff + lo;
		e[2] = diff - lo;
	}
	static inline Odd odd_part(int s1, int s3, int s5, int s7) {
		const int p3 = s7 + s3, p4 = s5 + s1, p1 = s7 + s1, p2 = s5 + s3;
		const int p5 = (p3 + p4) * fix(1.175875602f);
		const int q1 = p5 + p1 * fix(-0.899976223f), q2 = p5 + p2 * fix(-2.562915447f);
		const int q3 = p3 * fix(-1.961570560f), q4 = p4 * fix(-0.390180644f);
		Odd o;
		o.a = s7 * fix(0.298631336f) + q1 + q3;
		o.b = s5 * fix(2.053119869f) + q2 + q4;
		o.c = s3 * fix(3.072711026f) + q2 + q3;
		o.d = s1 * fix(1.501321110f) + q1 + q4;
		return o;
	}
	static inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

	static void idct8x8(const int16_t *in, const uint16_t *q, uint8_t *out, int stride) {
		int d[64], mid[64];
		for (int i = 0; i < 64; ++i) d[i] = (int16_t)(in[i] * q[i]); // dequantised coefficient as a 16-bit value
		for (int x = 0; x < 8; ++x) {
			const int *c = d + x;
			int *m = mid + x;
			if (!(c[8] | c[16] | c[24] | c[32] | c[40] | c[48] | c[56])) {
				const int dc = c[0] * 4;
				for (int y = 0; y < 8; ++y) m[8 * y] = dc;
				continue;
			}
			int e[4];
			even_part(c[0], c[16], c[32], c[48], e);
			const Odd o = odd_part(c[8], c[24], c[40], c[56]);
			for (int k = 0; k < 4; ++k) e[k] += 512;
			m[0] = (e[0] + o.d) >> 10;
			m[56] = (e[0] - o.d) >> 10;
			m[8] = (e[1] + o.c) >> 10;
			m[48] = (e[1] - o.c) >> 10;
			m[16] = (e[2] + o.b) >> 10;
			m[40] = (e[2] - o.b) >> 10;
			m[24] = (e[3] + o.a) >> 10;
			m[32] = (e[3] - o.a) >> 10;
		}
		for (int y = 0; y < 8; ++y) {
			const int *r = mid + 8 * y;
			uint8_t *o8 = out + (size_t)y * stride;
			int e[4];
			even_part(r[0], r[2], r[4], r[6], e);
			const Odd o = odd_part(r[1], r[3], r[5], r[7]);
			for (int k = 0; k < 4; ++k) e[k] += 65536 + (128 << 17);
			o8[0] = clamp8((e[0] + o.d) >> 17);
			o8[7] = clamp8((e[0] - o.d) >> 17);
			o8[1] = clamp8((e[1] + o.c) >> 17);
			o8[6] = clamp8((e[1] - o.c) >> 17);
			o8[2] = clamp8((e[2] + o.b) >> 17);
			o8[5] = clamp8((e[2] - o.b) >> 17);
			o8[3] = clamp8((e[3] + o.a) >> 17);
			o8[4] = clamp8((e[3] - o.a) >> 17);
		}
	}

	void reconstruct() {
		for (int c = 0; c < ncomp_; ++c) {
			Plane &p = planes_[c];
			const int stride = p.store_w * 8;
			p.samples.assign((size_t)stride * p.store_h * 8, 0);
			for (int by = 0; by < p.store_h; ++by)
				for (int bx = 0; bx < p.store_w; ++bx)
					idct8x8(p.block(bx, by), quant_[p.tq], &p.samples[(size_t)by * 8 * stride + (size_t)bx * 8], stride);
		}
	}

	// PIXEL CONTRACT (stb_image v2.27): chroma upsampling.  Factor 2 uses the "triangle" filter -- each
	// output sample is 3/4 of the nearest input sample and 1/4 of the next nearest, in both directions
	// for 2x2 (then with 4 fractional bits: (3a + b + 8) >> 4), with +2 >> 2 rounding in one direction;
	// the first and last output of a row copy the edge sample (2x1) or use only the vertical blend
	// (2x2).  Other factors repeat samples.  Vertically the next-nearest row is the one above for even
	// output rows and the one below for odd ones, clamped to the component's own rows.
	void upsample_row(const Plane &p, int row, std::vector<uint8_t> &line, std::vector<uint8_t> &scratch) const {
		const int hx = hmax_ / p.hs, vx = vmax_ / p.vs;
		const int stride = p.store_w * 8;
		const int w = (width_ + hx - 1) / hx; // input samples that contribute to the visible row
		const int near_row = row / vx;
		const uint8_t *a = &p.samples[(size_t)near_row * stride];
		line.resize((size_t)w * hx + 8);
		if (hx == 1 && vx == 1) {
			std::memcpy(line.data(), a, (size_t)w);
			return;
		}
		int far_row = near_row;
		if (vx == 2) {
			far_row = (row & 1) ? near_row + 1 : near_row - 1;
			if (far_row < 0) far_row = 0;
			if (far_row > p.height - 1) far_row = p.height - 1;
		}
		const uint8_t *b = &p.samples[(size_t)far_row * stride];
		if (hx == 1 && vx == 2) {
			for (int i = 0; i < w; ++i) line[(size_t)i] = (uint8_t)((3 * a[i] + b[i] + 2) >> 2);
		} else if (hx == 2 && vx == 1) {
			if (w == 1) {
				line[0] = line[1] = a[0];
				return;
			}
			line[0] = a[0];
			line[1] = (uint8_t)((a[0] * 3 + a[1] + 2) >> 2);
			for (int i = 1; i < w - 1; ++i) {
				const int n = 3 * a[i] + 2;
				line[(size_t)2 * i] = (uint8_t)((n + a[i - 1]) >> 2);
				line[(size_t)2 * i + 1] = (uint8_t)((n + a[i + 1]) >> 2);
			}
			// (pixel contract: this one sample weighs its LEFT neighbour 3/4, unlike every other one)
			line[(size_t)2 * w - 2] = (uint8_t)((3 * a[w - 2] + a[w - 1] + 2) >> 2);
			line[(size_t)2 * w - 1] = a[w - 1];
		} else if (hx == 2 && vx == 2) {
			if (w == 1) {
				line[0] = line[1] = (uint8_t)((3 * a[0] + b[0] + 2) >> 2);
				return;
			}
			int prev = 3 * a[0] + b[0];
			line[0] = (uint8_t)((prev + 2) >> 2);
			for (int i = 1; i < w; ++i) {
				const int cur = 3 * a[i] + b[i];
				line[(size_t)2 * i - 1] = (uint8_t)((3 * prev + cur + 8) >> 4);
				line[(size_t)2 * i] = (uint8_t)((3 * cur + prev + 8) >> 4);
				prev = cur;
			}
			line[(size_t)2 * w - 1] = (uint8_t)((prev + 2) >> 2);
		} else {
			// any other whole factor: each sample of the nearest row repeated hx times
			for (int i = 0; i < w; ++i)
				for (int k = 0; k < hx; ++k) line[(size_t)i * hx + k] = a[i];
		}
		(void)scratch;
	}

	// PIXEL CONTRACT (stb_image v2.27): YCbCr -> RGB with 20 fractional bits; coefficients are
	// round(c * 4096) << 8, luma carries the rounding half, and the Cb term of green is masked to its
	// upper 16 bits before the sum.
	static constexpr int cfix(float c) { return ((int)(c * 4096.0f + 0.5f)) << 8; }
	static inline void ycc_to_rgb(int y, int cb, int cr, uint8_t *rgb) {
		const int yf = (y << 20) + (1 << 19);
		cb -= 128;
		cr -= 128;
		const int r = yf + cr * cfix(1.40200f);
		const int g = yf + cr * -cfix(0.71414f) + (int)((unsigned)(cb * -cfix(0.34414f)) & 0xffff0000u);
		const int b = yf + cb * cfix(1.77200f);
		rgb[0] = clamp8(r >> 20);
		rgb[1] = clamp8(g >> 20);
		rgb[2] = clamp8(b >> 20);
	}
	// PIXEL CONTRACT (stb_image v2.27): x*y/255 as (t + (t >> 8)) >> 8 with t = x*y + 128 (CMYK / YCCK);
	// luma of an RGB triple as (77 r + 150 g + 29 b) >> 8.
	static inline uint8_t mul255(int x, int y) {
		const unsigned t = (unsigned)(x * y + 128);
		return (uint8_t)((t + (t >> 8)) >> 8);
	}
	static inline uint8_t luma(int r, int g, int b) { return (uint8_t)((r * 77 + g * 150 + b * 29) >> 8); }

	void emit(int req_comp, Image *out) {
		const int n = req_comp ? req_comp : (ncomp_ >= 3 ? 3 : 1);
		const bool is_rgb = ncomp_ == 3 && (rgb_ids_ == 3 || (adobe_transform_ == 0 && !jfif_));
		// grey output of a YCbCr file needs the luma plane only
		const int used = (ncomp_ == 3 && n < 3 && !is_rgb) ? 1 : ncomp_;
		out->w = width_;
		out->h = height_;
		out->comp = n;
		out->comp_in_file = ncomp_ >= 3 ? 3 : 1;
		out->px.assign((size_t)width_ * height_ * n, 0);
		std::vector<uint8_t> line[4], scratch;
		for (int y = 0; y < height_; ++y) {
			for (int c = 0; c < used; ++c) upsample_row(planes_[c], y, line[c], scratch);
			uint8_t *o = &out->px[(size_t)y * width_ * n];
			const uint8_t *c0 = line[0].data(), *c1 = line[1].data(), *c2 = line[2].data(), *c3 = line[3].data();
			for (int x = 0; x < width_; ++x, o += n) {
				uint8_t rgb[3];
				if (ncomp_ == 1 || used == 1) {
					rgb[0] = rgb[1] = rgb[2] = c0[x];
					if (n < 3) {
						o[0] = c0[x];
						if (n == 2) o[1] = 255;
						continue;
					}
				} else if (ncomp_ == 3) {
					if (is_rgb) {
						rgb[0] = c0[x];
						rgb[1] = c1[x];
						rgb[2] = c2[x];
					} else {
						ycc_to_rgb(c0[x], c1[x], c2[x], rgb);
					}
				} else { // four components
					const int k = c3[x];
					if (adobe_transform_ == 0) { // CMYK stored inverted
						rgb[0] = mul255(c0[x], k);
						rgb[1] = mul255(c1[x], k);
						rgb[2] = mul255(c2[x], k);
					} else if (adobe_transform_ == 2) { // YCCK
						if (n < 3) {
							// (grey from YCCK: the inverted luma times K, no colour conversion)
							o[0] = mul255(255 - c0[x], k);
							if (n == 2) o[1] = 255;
							continue;
						}
						ycc_to_rgb(c0[x], c1[x], c2[x], rgb);
						rgb[0] = mul255(255 - rgb[0], k);
						rgb[1] = mul255(255 - rgb[1], k);
						rgb[2] = mul255(255 - rgb[2], k);
					} else { // no Adobe marker: treated as YCbCr, fourth channel ignored
						if (n < 3) {
							o[0] = c0[x];
							if (n == 2) o[1] = 255;
							continue;
						}
						ycc_to_rgb(c0[x], c1[x], c2[x], rgb);
					}
				}
				if (n >= 3) {
					o[0] = rgb[0];
					o[1] = rgb[1];
					o[2] = rgb[2];
					if (n == 4) o[3] = 255;
				} else {
					o[0] = luma(rgb[0], rgb[1], rgb[2]);
					if (n == 2) o[1] = 255;
				}
			}
		}
	}
};

} // namespace

bool decode_jpeg(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	if (req_comp < 0 || req_comp > 4) {
		*err = "bad req_comp";
		return false;
	}
	try {
		JpegDecoder dec(bytes, len);
		dec.decode(req_comp, out);
	} catch (const Fail &f) {
		*err = f.why;
		return false;
	} catch (const std::bad_alloc &) {
		*err = "out of memory";
		return false;
	}
	return true;
}

} // namespace hmrm
