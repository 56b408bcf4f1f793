// bmp_tga_decode.cpp -- Windows / OS2 BMP and Truevision TGA decoders for height / colour maps.
//
// Written for this project from the file formats (BITMAPFILEHEADER + BITMAPCOREHEADER / INFOHEADER /
// V4 / V5 headers; the 18-byte TGA header of the Truevision 2.0 specification), structured as:
// parse the header into a plain description, then one pixel-fetch routine per storage class.
//
// The reference loads its maps with stb_image v2.27 (main/hmap.cpp:320-321, 341-342), so where the
// formats leave room the choices are that loader's ("pixel contract" below): which variants are
// accepted, how sub-byte channel fields are widened to 8 bits (bit replication), that a 32-bit BMP
// whose alpha bytes are all zero is opaque, that a BMP reports 4 channels exactly when it has an alpha
// mask, that 16-bit TGA pixels are 5-5-5 scaled by 255/31.  tests/golden/bmp_tga_decode.npz holds what
// the reference's build of stb produces for 23 variants; tests/test_image_io.py compares, also live.
#include <cstring>
#include <string>
#include <vector>

#include "image_io.hpp"

namespace hmrm {
namespace {

// Bounds-checked little-endian cursor.
class Cursor {
public:
	Cursor(const uint8_t *p, size_t n) : p_(p), n_(n) {}
	bool ok() const { return !bad_; }
	size_t pos() const { return at_; }
	size_t left() const { return at_ <= n_ ? n_ - at_ : 0; }
	uint32_t u8() {
		if (at_ >= n_) {
			bad_ = true;
			return 0;
		}
		return p_[at_++];
	}
	uint32_t u16() {
		const uint32_t lo = u8();
		return lo | (u8() << 8);
	}
	uint32_t u32() {
		const uint32_t lo = u16();
		return lo | (u16() << 16);
	}
	void skip(int64_t k) {
		if (k < 0 || (uint64_t)k > left()) {
			bad_ = true;
			at_ = n_;
			return;
		}
		at_ += (size_t)k;
	}
	const uint8_t *here() const { return p_ + at_; }

private:
	const uint8_t *p_;
	size_t n_, at_ = 0;
	bool bad_ = false;
};

bool fail(std::string *err, const char *why) {
	*err = why;
	return false;
}

bool finish(Image *img, std::vector<uint8_t> &px, int w, int h, int have, int file_comp, int req_comp) {
	img->w = w;
	img->h = h;
	img->comp_in_file = file_comp;
	const int want = req_comp ? req_comp : have;
	img->px = want == have ? std::move(px) : convert_channels8(px, have, want, (size_t)w * h);
	img->comp = want;
	return true;
}

void flip_rows(std::vector<uint8_t> &px, int h, size_t row_bytes) {
	std::vector<uint8_t> tmp(row_bytes);
	for (int y = 0; y < h / 2; ++y) {
		uint8_t *a = &px[(size_t)y * row_bytes], *b = &px[(size_t)(h - 1 - y) * row_bytes];
		std::memcpy(tmp.data(), a, row_bytes);
		std::memcpy(a, b, row_bytes);
		std::memcpy(b, tmp.data(), row_bytes);
	}
}

// ------------------------------------------------------------------- BMP ----
// One channel of a packed pixel: the field selected by `mask`, widened to 8 bits by repeating its
// bit pattern (pixel contract; fields wider than 8 bits keep their top 8).
struct Field {
	uint32_t mask = 0;
	int shift = 0, bits = 0;
	bool set(uint32_t m) {
		mask = m;
		shift = bits = 0;
		if (!m) return true;
		while (!((m >> shift) & 1u)) ++shift;
		uint32_t run = m >> shift;
		while (run & 1u) {
			++bits;
			run >>= 1;
		}
		return run == 0; // one contiguous run of ones
	}
	uint8_t get(uint32_t px) const {
		uint32_t v = (px & mask) >> shift;
		if (bits >= 8) return (uint8_t)(v >> (bits - 8));
		uint32_t out = 0;
		for (int filled = 0; filled < 8; filled += bits) {
			const int room = 8 - filled;
			out |= room >= bits ? v << (room - bits) : v >> (bits - room);
		}
		return (uint8_t)out;
	}
};

struct BmpInfo {
	int width = 0, height = 0;
	bool bottom_up = true;
	int bpp = 0, header = 0;
	uint32_t compression = 0;
	uint32_t data_offset = 0;
	Field r, g, b, a;
	bool has_alpha_mask = false;
	bool plain_bgra = false;   // 32-bit with the standard byte masks
	bool default_masks32 = false; // 32-bit without bit-field masks of its own: alpha that is zero everywhere means opaque
};

bool bmp_header(Cursor &c, BmpInfo *bi, std::string *err) {
	if (c.u8() != 'B' || c.u8() != 'M') return fail(err, "not BMP");
	c.u32(); // file size
	c.u32(); // reserved
	bi->data_offset = c.u32();
	bi->header = (int)c.u32();
	const int hs = bi->header;
	if (hs != 12 && hs != 40 && hs != 56 && hs != 108 && hs != 124) return fail(err, "unknown BMP header");
	int w, h;
	if (hs == 12) {
		w = (int)c.u16();
		h = (int)c.u16();
	} else {
		w = (int)c.u32();
		h = (int)c.u32();
	}
	if (c.u16() != 1) return fail(err, "bad BMP planes");
	bi->bpp = (int)c.u16();
	uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
	bool masks_from_file = false;
	if (hs != 12) {
		bi->compression = c.u32();
		if (bi->compression == 1 || bi->compression == 2) return fail(err, "BMP RLE is not supported");
		if (bi->compression >= 4) return fail(err, "BMP with embedded JPEG/PNG is not supported");
		if (bi->compression == 3 && bi->bpp != 16 && bi->bpp != 32) return fail(err, "bad BMP bitfields");
		c.skip(20); // image size, resolution x/y, colours used / important
		if (hs == 40 || hs == 56) {
			if (hs == 56) c.skip(16);
			if ((bi->bpp == 16 || bi->bpp == 32) && bi->compression == 3) {
				mr = c.u32();
				mg = c.u32();
				mb = c.u32();
				masks_from_file = true;
				if (mr == mg && mg == mb) return fail(err, "bad BMP masks");
			}
		} else {
			mr = c.u32();
			mg = c.u32();
			mb = c.u32();
			ma = c.u32();
			masks_from_file = bi->compression == 3;
			c.skip(4 + 48);          // colour space type + end points and gammas
			if (hs == 124) c.skip(16); // intent, profile data / size, reserved
		}
	}
	if (!c.ok()) return fail(err, "truncated BMP header");
	if (!masks_from_file) { // pixel contract: 5-5-5 for 16 bits, B,G,R,A bytes for 32 bits
		mr = mg = mb = ma = 0;
		if (bi->bpp == 16) {
			mr = 31u << 10;
			mg = 31u << 5;
			mb = 31u;
		} else if (bi->bpp == 32) {
			mr = 0xffu << 16;
			mg = 0xffu << 8;
			mb = 0xffu;
			ma = 0xffu << 24;
		}
	}
	if (bi->bpp == 16 || bi->bpp == 32) {
		if (!mr || !mg || !mb) return fail(err, "bad BMP masks");
		if (!bi->r.set(mr) || !bi->g.set(mg) || !bi->b.set(mb) || !bi->a.set(ma)) return fail(err, "bad BMP masks");
		if (bi->r.bits > 8 || bi->g.bits > 8 || bi->b.bits > 8 || bi->a.bits > 8) return fail(err, "bad BMP masks");
		bi->plain_bgra = bi->bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0xff0000u && ma == 0xff000000u;
	}
	bi->has_alpha_mask = ma != 0;
	bi->default_masks32 = bi->bpp == 32 && !masks_from_file;
	bi->bottom_up = h > 0;
	bi->width = w;
	bi->height = h < 0 ? -h : h;
	if (bi->width <= 0 || bi->height <= 0 || bi->width > (1 << 24) || bi->height > (1 << 24) ||
	    (int64_t)bi->width * bi->height > ((int64_t)1 << 28))
		return fail(err, "bad BMP size");
	return true;
}

} // namespace

bool looks_like_bmp(const uint8_t *bytes, size_t len) {
	if (len < 18 || bytes[0] != 'B' || bytes[1] != 'M') return false;
	const uint32_t hs = bytes[14] | (bytes[15] << 8) | (bytes[16] << 16) | ((uint32_t)bytes[17] << 24);
	return hs == 12 || hs == 40 || hs == 56 || hs == 108 || hs == 124;
}

bool decode_bmp(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Cursor c(bytes, len);
	BmpInfo bi;
	if (!bmp_header(c, &bi, err)) return false;
	const int w = bi.width, h = bi.height;
	const int file_comp = bi.has_alpha_mask ? 4 : 3;
	// decode straight into 3 or 4 channels: 4 when the caller wants 4 or the file has alpha and the
	// caller did not ask for exactly 3
	const int have = (req_comp >= 3) ? req_comp : file_comp;
	std::vector<uint8_t> px((size_t)w * h * have);
	const int64_t after_headers = 14 + (int64_t)bi.header;
	bool any_alpha = !bi.default_masks32;

	if (bi.bpp == 1 || bi.bpp == 4 || bi.bpp == 8) {
		const int entry = bi.header == 12 ? 3 : 4;
		const int64_t table_bytes = (int64_t)bi.data_offset - after_headers;
		const int64_t entries = table_bytes / entry;
		if (entries <= 0 || entries > 256) return fail(err, "bad BMP palette");
		uint8_t pal[256][3];
		for (int i = 0; i < (int)entries; ++i) {
			pal[i][2] = (uint8_t)c.u8();
			pal[i][1] = (uint8_t)c.u8();
			pal[i][0] = (uint8_t)c.u8();
			if (entry == 4) c.u8();
		}
		c.skip(table_bytes - entries * entry);
		const int row_data = bi.bpp == 8 ? w : (bi.bpp == 4 ? (w + 1) / 2 : (w + 7) / 8);
		const int pad = (4 - (row_data & 3)) & 3;
		if (!c.ok() || (int64_t)c.left() < (int64_t)(row_data + pad) * (h - 1) + row_data) return fail(err, "truncated BMP");
		for (int y = 0; y < h; ++y) {
			const uint8_t *row = c.here();
			uint8_t *o = &px[(size_t)y * w * have];
			for (int x = 0; x < w; ++x, o += have) {
				int idx;
				if (bi.bpp == 8) idx = row[x];
				else if (bi.bpp == 4) idx = (x & 1) ? (row[x >> 1] & 15) : (row[x >> 1] >> 4);
				else idx = (row[x >> 3] >> (7 - (x & 7))) & 1;
				if (idx >= (int)entries) idx = 0; // (an index beyond the table: colour 0)
				o[0] = pal[idx][0];
				o[1] = pal[idx][1];
				o[2] = pal[idx][2];
				if (have == 4) o[3] = 255;
			}
			c.skip(y + 1 < h ? row_data + pad : row_data);
		}
	} else if (bi.bpp == 16 || bi.bpp == 24 || bi.bpp == 32) {
		c.skip((int64_t)bi.data_offset - (int64_t)c.pos()); // (bit-field masks may follow a 40/56-byte header)
		const int bytes_pp = bi.bpp / 8;
		const int64_t row_data = (int64_t)w * bytes_pp;
		const int pad = (int)((4 - (row_data & 3)) & 3);
		if (!c.ok() || (int64_t)c.left() < (row_data + pad) * (h - 1) + row_data) return fail(err, "truncated BMP");
		for (int y = 0; y < h; ++y) {
			const uint8_t *s = c.here();
			uint8_t *o = &px[(size_t)y * w * have];
			for (int x = 0; x < w; ++x, o += have, s += bytes_pp) {
				uint8_t a = 255;
				if (bi.bpp == 24 || bi.plain_bgra) {
					o[0] = s[2];
					o[1] = s[1];
					o[2] = s[0];
					if (bi.bpp == 32) a = s[3];
				} else {
					const uint32_t v = bi.bpp == 16 ? (uint32_t)(s[0] | (s[1] << 8))
					                                : (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
					o[0] = bi.r.get(v);
					o[1] = bi.g.get(v);
					o[2] = bi.b.get(v);
					if (bi.a.mask) a = bi.a.get(v);
				}
				any_alpha = any_alpha || a != 0;
				if (have == 4) o[3] = a;
			}
			c.skip(y + 1 < h ? row_data + pad : row_data);
		}
		// pixel contract: default-mask 32-bit files whose alpha bytes are zero everywhere are opaque
		if (have == 4 && !any_alpha)
			for (size_t i = 3; i < px.size(); i += 4) px[i] = 255;
	} else {
		return fail(err, "unsupported BMP bit depth");
	}
	if (bi.bottom_up) flip_rows(px, h, (size_t)w * have);
	return finish(out, px, w, h, have, file_comp, req_comp);
}

// ------------------------------------------------------------------- TGA ----
namespace {

struct TgaInfo {
	int id_len = 0, map_type = 0, image_type = 0;
	int map_first = 0, map_len = 0, map_bits = 0;
	int width = 0, height = 0, bpp = 0, descriptor = 0;
	bool rle = false;
};

// channels of one pixel (or palette entry) of `bits` bits; 15/16-bit colour is 5-5-5 -> 3 channels
int tga_channels(int bits, bool grey, bool *packed555) {
	*packed555 = false;
	switch (bits) {
	case 8: return 1;
	case 16:
		if (grey) return 2; // grey + alpha
		// fall through
	case 15: *packed555 = true; return 3;
	case 24: return 3;
	case 32: return 4;
	default: return 0;
	}
}

bool tga_header(Cursor &c, TgaInfo *t) {
	t->id_len = (int)c.u8();
	t->map_type = (int)c.u8();
	t->image_type = (int)c.u8();
	t->map_first = (int)c.u16();
	t->map_len = (int)c.u16();
	t->map_bits = (int)c.u8();
	c.u16(); // x origin
	c.u16(); // y origin
	t->width = (int)c.u16();
	t->height = (int)c.u16();
	t->bpp = (int)c.u8();
	t->descriptor = (int)c.u8();
	t->rle = t->image_type >= 8;
	if (t->rle) t->image_type -= 8;
	return c.ok();
}

bool tga_plausible(const TgaInfo &t) {
	if (t.map_type > 1) return false;
	if (t.map_type == 1) {
		if (t.image_type != 1) return false;
		if (t.map_bits != 8 && t.map_bits != 15 && t.map_bits != 16 && t.map_bits != 24 && t.map_bits != 32) return false;
		if (t.bpp != 8 && t.bpp != 16) return false;
	} else {
		if (t.image_type != 2 && t.image_type != 3) return false;
		if (t.bpp != 8 && t.bpp != 15 && t.bpp != 16 && t.bpp != 24 && t.bpp != 32) return false;
	}
	return t.width >= 1 && t.height >= 1;
}

// pixel contract: 5 bits -> 8 bits as v * 255 / 31
inline void unpack555(uint32_t v, uint8_t *rgb) {
	rgb[0] = (uint8_t)((((v >> 10) & 31u) * 255u) / 31u);
	rgb[1] = (uint8_t)((((v >> 5) & 31u) * 255u) / 31u);
	rgb[2] = (uint8_t)(((v & 31u) * 255u) / 31u);
}

} // namespace

bool looks_like_tga(const uint8_t *bytes, size_t len) {
	Cursor c(bytes, len);
	TgaInfo t;
	return tga_header(c, &t) && tga_plausible(t);
}

bool decode_tga(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Cursor c(bytes, len);
	TgaInfo t;
	if (!tga_header(c, &t) || !tga_plausible(t)) return fail(err, "bad TGA header");
	const bool indexed = t.map_type == 1;
	bool packed = false;
	const int comp = tga_channels(indexed ? t.map_bits : t.bpp, t.image_type == 3, &packed);
	if (comp == 0) return fail(err, "bad TGA format");
	const int w = t.width, h = t.height;
	if ((int64_t)w * h > ((int64_t)1 << 28)) return fail(err, "TGA too large");
	c.skip(t.id_len);

	std::vector<uint8_t> palette;
	if (indexed) {
		if (t.map_len == 0) return fail(err, "bad TGA palette");
		c.skip(t.map_first); // pixel contract: the first-entry field is skipped as a byte count
		palette.resize((size_t)t.map_len * comp);
		for (int i = 0; i < t.map_len; ++i) {
			uint8_t *e = &palette[(size_t)i * comp];
			if (packed) unpack555(c.u16(), e);
			else
				for (int k = 0; k < comp; ++k) e[k] = (uint8_t)c.u8();
		}
		if (!c.ok()) return fail(err, "truncated TGA palette");
	}

	// one stored pixel -> comp bytes, still in file channel order (B,G,R[,A] for true colour)
	auto fetch = [&](uint8_t *dst) {
		if (indexed) {
			uint32_t idx = t.bpp == 8 ? c.u8() : c.u16();
			if (idx >= (uint32_t)t.map_len) idx = 0;
			std::memcpy(dst, &palette[(size_t)idx * comp], (size_t)comp);
		} else if (packed) {
			unpack555(c.u16(), dst);
		} else {
			for (int k = 0; k < comp; ++k) dst[k] = (uint8_t)c.u8();
		}
	};

	const size_t npix = (size_t)w * h;
	std::vector<uint8_t> px(npix * comp);
	if (!t.rle) {
		const int stored = indexed ? t.bpp / 8 : (packed ? 2 : comp);
		if (c.left() < npix * (size_t)stored) return fail(err, "truncated TGA");
		for (size_t i = 0; i < npix; ++i) fetch(&px[i * comp]);
	} else {
		size_t i = 0;
		uint8_t run_px[4] = {0, 0, 0, 0};
		while (i < npix) {
			const uint32_t packet = c.u8();
			size_t count = (packet & 127u) + 1;
			if (!c.ok()) return fail(err, "truncated TGA");
			if (count > npix - i) count = npix - i; // (a packet may not run past the image)
			if (packet & 128u) {
				fetch(run_px);
				for (size_t k = 0; k < count; ++k) std::memcpy(&px[(i + k) * comp], run_px, (size_t)comp);
			} else {
				for (size_t k = 0; k < count; ++k) fetch(&px[(i + k) * comp]);
			}
			if (!c.ok()) return fail(err, "truncated TGA");
			i += count;
		}
	}
	if (!c.ok()) return fail(err, "truncated TGA");
	// bit 5 of the descriptor set = first row is the top row
	if (!((t.descriptor >> 5) & 1)) flip_rows(px, h, (size_t)w * comp);
	// true colour is stored blue first (5-5-5 pixels were unpacked as R,G,B already)
	if (comp >= 3 && !packed)
		for (size_t i = 0; i < npix; ++i) {
			const uint8_t b = px[i * comp];
			px[i * comp] = px[i * comp + 2];
			px[i * comp + 2] = b;
		}
	return finish(out, px, w, h, comp, comp, req_comp);
}

} // namespace hmrm
