// bmp_tga_decode.cpp -- Windows BMP and Truevision TGA decoders, written for this
// project.  Two more of the formats the reference accepts for its maps through
// stbi_load (README.md "Options": "JPEG, PNG, TGA, BMP, ..."; main/hmap.cpp:320-321,
// 341-342).  Conventions follow stb_image v2.27 so that the decoded pixels are the
// same (checked against the reference's own stb build in tests/test_image_io.py):
//   BMP (vendor/stb_image.h:5282-5655): 1/4/8-bit palettes, 16/24/32-bit direct colour
//     with default or BI_BITFIELDS masks (channel = masked bits scaled to 8 bits by bit
//     replication), bottom-up unless the height is negative, an all-zero alpha channel of a
//     32-bit file reads as opaque, RLE / embedded JPEG/PNG refused;
//   TGA (:5661-5990): types 1/2/3 and their RLE forms 9/10/11, 8/15/16/24/32 bits, 15/16-bit
//     pixels and palette entries as 5-5-5 RGB scaled by 255/31, bottom-up unless bit 5 of the
//     descriptor is set, BGR(A) -> RGB(A).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "image_io.hpp"

namespace hmrm {
namespace {

struct Reader {
	const uint8_t *base, *p, *end;
	int get8() { return p < end ? *p++ : 0; } // past the end reads as zeros, like stb's reader
	int get16le() { int a = get8(); return a | (get8() << 8); }
	uint32_t get32le() { uint32_t a = (uint32_t)get16le(); return a | ((uint32_t)get16le() << 16); }
	void skip(int n) {
		if (n < 0) { p = end; return; } // stb: a negative skip jumps to the end of the data
		p = (end - p) < n ? end : p + n;
	}
	long consumed() const { return (long)(p - base); }
};

int high_bit(uint32_t z) {
	if (z == 0) return -1;
	int n = 0;
	if (z >= 0x10000) { n += 16; z >>= 16; }
	if (z >= 0x00100) { n += 8; z >>= 8; }
	if (z >= 0x00010) { n += 4; z >>= 4; }
	if (z >= 0x00004) { n += 2; z >>= 2; }
	if (z >= 0x00002) { n += 1; }
	return n;
}
int bit_count(uint32_t a) {
	int n = 0;
	for (; a; a &= a - 1) ++n;
	return n;
}
// `bits` masked bits whose top bit sits `shift` above bit 7 -> 8 bits by bit replication
int scale_masked(uint32_t v, int shift, int bits) {
	static const unsigned mul_table[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01};
	static const unsigned shift_table[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
	if (shift < 0) v <<= -shift;
	else v >>= shift;
	v >>= (8 - bits);
	return (int)((unsigned)v * mul_table[bits]) >> shift_table[bits];
}

} // namespace

bool looks_like_bmp(const uint8_t *b, size_t len) {
	if (len < 18 || b[0] != 'B' || b[1] != 'M') return false;
	const uint32_t sz = (uint32_t)b[14] | ((uint32_t)b[15] << 8) | ((uint32_t)b[16] << 16) | ((uint32_t)b[17] << 24);
	return sz == 12 || sz == 40 || sz == 56 || sz == 108 || sz == 124;
}

bool decode_bmp(const uint8_t *bytes, size_t len, int req_comp, Image *img, std::string *err) {
	Reader s{bytes, bytes, bytes + len};
	auto fail = [&](const char *m) { *err = m; return false; };
	if (s.get8() != 'B' || s.get8() != 'M') return fail("not BMP");
	s.get32le(); s.get16le(); s.get16le();
	const int offset = (int)s.get32le();
	const int hsz = (int)s.get32le();
	uint32_t mr = 0, mg = 0, mb = 0, ma = 0, all_a = 255;
	int extra_read = 14;
	if (offset < 0) return fail("bad BMP");
	if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return fail("unknown BMP");
	int img_x, img_y_signed;
	if (hsz == 12) { img_x = s.get16le(); img_y_signed = s.get16le(); }
	else { img_x = (int)s.get32le(); img_y_signed = (int)s.get32le(); }
	if (s.get16le() != 1) return fail("bad BMP");
	const int bpp = s.get16le();
	auto mask_defaults = [&](int compress) {
		if (compress == 3) return;
		if (compress == 0) {
			if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
			else if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; all_a = 0; }
			else mr = mg = mb = ma = 0;
		}
	};
	if (hsz != 12) {
		const int compress = (int)s.get32le();
		if (compress == 1 || compress == 2) return fail("BMP RLE");
		if (compress >= 4) return fail("BMP JPEG/PNG");
		if (compress == 3 && bpp != 16 && bpp != 32) return fail("bad BMP");
		s.get32le(); s.get32le(); s.get32le(); s.get32le(); s.get32le();
		if (hsz == 40 || hsz == 56) {
			if (hsz == 56) { s.get32le(); s.get32le(); s.get32le(); s.get32le(); }
			if (bpp == 16 || bpp == 32) {
				if (compress == 0) mask_defaults(compress);
				else if (compress == 3) {
					mr = s.get32le(); mg = s.get32le(); mb = s.get32le();
					extra_read += 12;
					if (mr == mg && mg == mb) return fail("bad BMP");
				} else return fail("bad BMP");
			}
		} else {
			mr = s.get32le(); mg = s.get32le(); mb = s.get32le(); ma = s.get32le();
			if (compress != 3) mask_defaults(compress);
			s.get32le();
			for (int i = 0; i < 12; ++i) s.get32le();
			if (hsz == 124) { s.get32le(); s.get32le(); s.get32le(); s.get32le(); }
		}
	}
	const bool flip = img_y_signed > 0;
	const int img_y = std::abs(img_y_signed);
	if (img_x <= 0 || img_y <= 0 || img_x > (1 << 24) || img_y > (1 << 24)) return fail("too large");
	int psize = 0;
	if (hsz == 12) { if (bpp < 24) psize = (offset - extra_read - 24) / 3; }
	else if (bpp < 16) psize = (offset - extra_read - hsz) >> 2;
	if (psize == 0 && offset != s.consumed()) return fail("bad offset");
	const int img_n = (bpp == 24 && ma == 0xff000000u) ? 3 : (ma ? 4 : 3);
	const int target = (req_comp && req_comp >= 3) ? req_comp : img_n;
	if ((int64_t)img_x * img_y * target > ((int64_t)1 << 30)) return fail("too large");
	std::vector<uint8_t> out((size_t)img_x * img_y * target);
	size_t z = 0;
	if (bpp < 16) {
		if (psize <= 0 || psize > 256) return fail("invalid"); // (stb reads an uninitialised palette for psize < 0)
		uint8_t pal[256][4];
		for (int i = 0; i < psize; ++i) {
			pal[i][2] = (uint8_t)s.get8(); pal[i][1] = (uint8_t)s.get8(); pal[i][0] = (uint8_t)s.get8();
			if (hsz != 12) s.get8();
			pal[i][3] = 255;
		}
		for (int i = psize; i < 256; ++i) pal[i][0] = pal[i][1] = pal[i][2] = 0, pal[i][3] = 255;
		s.skip(offset - extra_read - hsz - psize * (hsz == 12 ? 3 : 4));
		int width;
		if (bpp == 1) width = (img_x + 7) >> 3;
		else if (bpp == 4) width = (img_x + 1) >> 1;
		else if (bpp == 8) width = img_x;
		else return fail("bad bpp");
		const int pad = (-width) & 3;
		auto put = [&](int c) {
			out[z++] = pal[c][0]; out[z++] = pal[c][1]; out[z++] = pal[c][2];
			if (target == 4) out[z++] = 255;
		};
		for (int j = 0; j < img_y; ++j) {
			if (bpp == 1) {
				int bit = 7, v = s.get8();
				for (int i = 0; i < img_x; ++i) {
					put((v >> bit) & 1);
					if (i + 1 == img_x) break;
					if (--bit < 0) { bit = 7; v = s.get8(); }
				}
			} else {
				for (int i = 0; i < img_x; i += 2) {
					int v = s.get8(), v2 = 0;
					if (bpp == 4) { v2 = v & 15; v >>= 4; }
					put(v);
					if (i + 1 == img_x) break;
					put(bpp == 8 ? s.get8() : v2);
				}
			}
			s.skip(pad);
		}
	} else {
		s.skip(offset - extra_read - hsz);
		int width = bpp == 24 ? 3 * img_x : (bpp == 16 ? 2 * img_x : 0);
		const int pad = (-width) & 3;
		int easy = 0;
		if (bpp == 24) easy = 1;
		else if (bpp == 32 && mb == 0xff && mg == 0xff00 && mr == 0x00ff0000 && ma == 0xff000000u) easy = 2;
		int rshift = 0, gshift = 0, bshift = 0, ashift = 0, rcount = 0, gcount = 0, bcount = 0, acount = 0;
		if (!easy) {
			if (!mr || !mg || !mb) return fail("bad masks");
			rshift = high_bit(mr) - 7; rcount = bit_count(mr);
			gshift = high_bit(mg) - 7; gcount = bit_count(mg);
			bshift = high_bit(mb) - 7; bcount = bit_count(mb);
			ashift = high_bit(ma) - 7; acount = bit_count(ma);
			if (rcount > 8 || gcount > 8 || bcount > 8 || acount > 8) return fail("bad masks");
		}
		for (int j = 0; j < img_y; ++j) {
			for (int i = 0; i < img_x; ++i) {
				unsigned a;
				if (easy) {
					out[z + 2] = (uint8_t)s.get8(); out[z + 1] = (uint8_t)s.get8(); out[z + 0] = (uint8_t)s.get8();
					z += 3;
					a = easy == 2 ? (unsigned)s.get8() : 255u;
				} else {
					const uint32_t v = bpp == 16 ? (uint32_t)s.get16le() : s.get32le();
					out[z++] = (uint8_t)scale_masked(v & mr, rshift, rcount);
					out[z++] = (uint8_t)scale_masked(v & mg, gshift, gcount);
					out[z++] = (uint8_t)scale_masked(v & mb, bshift, bcount);
					a = ma ? (unsigned)scale_masked(v & ma, ashift, acount) : 255u;
				}
				all_a |= a;
				if (target == 4) out[z++] = (uint8_t)a;
			}
			s.skip(pad);
		}
	}
	if (target == 4 && all_a == 0)
		for (size_t i = 3; i < out.size(); i += 4) out[i] = 255;
	if (flip) {
		const size_t row = (size_t)img_x * target;
		for (int j = 0; j < img_y >> 1; ++j)
			for (size_t i = 0; i < row; ++i) std::swap(out[(size_t)j * row + i], out[(size_t)(img_y - 1 - j) * row + i]);
	}
	img->w = img_x;
	img->h = img_y;
	img->comp_in_file = img_n;
	const int final_n = req_comp ? req_comp : target;
	img->px = convert_channels8(out, target, final_n, (size_t)img_x * img_y);
	img->comp = final_n;
	return true;
}

// ------------------------------------------------------------------- TGA ----
static int tga_components(int bits, bool is_grey, bool *rgb16) {
	*rgb16 = false;
	switch (bits) {
	case 8: return 1;
	case 16: if (is_grey) return 2; // fallthrough
	case 15: *rgb16 = true; return 3;
	case 24: case 32: return bits / 8;
	default: return 0;
	}
}

bool looks_like_tga(const uint8_t *b, size_t len) {
	Reader s{b, b, b + len};
	s.get8();
	const int color_type = s.get8();
	if (color_type > 1) return false;
	int sz = s.get8();
	if (color_type == 1) {
		if (sz != 1 && sz != 9) return false;
		s.skip(4);
		sz = s.get8();
		if (sz != 8 && sz != 15 && sz != 16 && sz != 24 && sz != 32) return false;
		s.skip(4);
	} else {
		if (sz != 2 && sz != 3 && sz != 10 && sz != 11) return false;
		s.skip(9);
	}
	if (s.get16le() < 1) return false;
	if (s.get16le() < 1) return false;
	sz = s.get8();
	if (color_type == 1 && sz != 8 && sz != 16) return false;
	if (sz != 8 && sz != 15 && sz != 16 && sz != 24 && sz != 32) return false;
	return true;
}

bool decode_tga(const uint8_t *bytes, size_t len, int req_comp, Image *img, std::string *err) {
	Reader s{bytes, bytes, bytes + len};
	auto fail = [&](const char *m) { *err = m; return false; };
	const int id_len = s.get8();
	const int indexed = s.get8();
	int image_type = s.get8();
	const int pal_start = s.get16le(), pal_len = s.get16le(), pal_bits = s.get8();
	s.get16le(); s.get16le();
	const int W = s.get16le(), H = s.get16le();
	const int bpp = s.get8();
	int inverted = s.get8();
	bool rle = false;
	if (image_type >= 8) { image_type -= 8; rle = true; }
	inverted = 1 - ((inverted >> 5) & 1);
	bool rgb16 = false;
	const int comp = indexed ? tga_components(pal_bits, false, &rgb16) : tga_components(bpp, image_type == 3, &rgb16);
	if (!comp) return fail("bad format");
	if (W <= 0 || H <= 0) return fail("bad format");
	if ((int64_t)W * H * comp > ((int64_t)1 << 30)) return fail("too large");
	std::vector<uint8_t> data((size_t)W * H * comp, 0);
	s.skip(id_len);
	auto read_rgb16 = [&](uint8_t *out) {
		const unsigned px = (unsigned)s.get16le();
		out[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31);
		out[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31);
		out[2] = (uint8_t)(((px & 31) * 255) / 31);
	};
	if (!indexed && !rle && !rgb16) {
		for (int i = 0; i < H; ++i) {
			const int row = inverted ? H - i - 1 : i;
			uint8_t *dst = &data[(size_t)row * W * comp];
			const size_t n = (size_t)W * comp, avail = (size_t)(s.end - s.p);
			if (avail >= n) { memcpy(dst, s.p, n); s.p += n; } // a short row is left untouched (stb: getn fails)
		}
	} else {
		std::vector<uint8_t> palette;
		if (indexed) {
			if (pal_len == 0) return fail("bad palette");
			s.skip(pal_start);
			palette.assign((size_t)pal_len * comp, 0);
			if (rgb16) {
				for (int i = 0; i < pal_len; ++i) read_rgb16(&palette[(size_t)i * comp]);
			} else {
				const size_t n = palette.size();
				if ((size_t)(s.end - s.p) < n) return fail("bad palette");
				memcpy(palette.data(), s.p, n);
				s.p += n;
			}
		}
		uint8_t raw[4] = {0, 0, 0, 0};
		int rle_count = 0, rle_repeating = 0;
		bool read_next = true;
		for (size_t i = 0; i < (size_t)W * H; ++i) {
			if (rle) {
				if (rle_count == 0) {
					const int cmd = s.get8();
					rle_count = 1 + (cmd & 127);
					rle_repeating = cmd >> 7;
					read_next = true;
				} else if (!rle_repeating) read_next = true;
			} else read_next = true;
			if (read_next) {
				if (indexed) {
					int idx = bpp == 8 ? s.get8() : s.get16le();
					if (idx >= pal_len) idx = 0;
					for (int j = 0; j < comp; ++j) raw[j] = palette[(size_t)idx * comp + j];
				} else if (rgb16) read_rgb16(raw);
				else for (int j = 0; j < comp; ++j) raw[j] = (uint8_t)s.get8();
				read_next = false;
			}
			for (int j = 0; j < comp; ++j) data[i * comp + j] = raw[j];
			if (rle) --rle_count;
		}
		if (inverted) {
			const size_t row = (size_t)W * comp;
			for (int j = 0; j * 2 < H; ++j)
				for (size_t i = 0; i < row; ++i) std::swap(data[(size_t)j * row + i], data[(size_t)(H - 1 - j) * row + i]);
		}
	}
	if (comp >= 3 && !rgb16)
		for (size_t i = 0; i < (size_t)W * H; ++i) std::swap(data[i * comp], data[i * comp + 2]);
	img->w = W;
	img->h = H;
	img->comp_in_file = comp;
	const int final_n = req_comp ? req_comp : comp;
	img->px = convert_channels8(data, comp, final_n, (size_t)W * H);
	img->comp = final_n;
	return true;
}

} // namespace hmrm
