// legacy_formats.cpp -- GIF, Photoshop PSD, Softimage PIC and Radiance HDR maps.
//
// The reference loads its height and colour maps with stb_image v2.27 (main/hmap.cpp:320-321, 341-342) and its
// README lists these four formats among the usable ones, so a drop-in has to read them -- and read them to the
// same 8-bit pixels, since the heights are computed from those pixels.  Written for this project from the format
// definitions (GIF89a specification incl. its variable-length-code LZW; Adobe's PSD file format, sections header /
// colour mode / image resources / layer info / image data with PackBits rows; the Softimage PIC channel-packet
// layout; Greg Ward's RGBE scanlines), each decoder structured as: parse the container into a small description,
// then fill an RGBA canvas.  What the formats leave open, or what a conforming reader would do differently, follows
// the reference's loader and is marked "pixel contract":
//   * all four read through a stream whose bytes past the end are zero (a truncated file decodes to whatever its
//     bytes give rather than failing);
//   * GIF: only the first image is composed; pixels it does not draw are transparent black, or -- when the
//     background index is not zero -- the background colour with its red and blue swapped; a transparent index
//     is simply not drawn; the file reports 4 channels;
//   * PSD: RGB mode only, 8 or 16 bits (raw 16-bit keeps the high byte; PackBits rows are taken as bytes whatever
//     the depth), at most the first four channels, a white matte is removed from colours under partial alpha with
//     single-precision arithmetic; reports 4 channels;
//   * PIC: canvas starts opaque white, 3 channels unless a packet carries alpha;
//   * HDR: new-style RLE scanlines or flat RGBE, tone-mapped to 8 bits by v^(1/2.2) * 255 + 0.5 in single
//     precision around a double-precision pow; 3 channels; grey = (r + g + b) / 3.
// tests/golden/legacy_decode.npz holds what the reference's own build of stb produces for the fixture files;
// tests/test_image_io.py compares against it, and live against oracle/_ref where that exists.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "image_io.hpp"

namespace hmrm {
namespace {

constexpr int kMaxSide = 1 << 24; // the reference loader's limit on either dimension

// Forgiving big/little-endian byte stream (pixel contract: zeros past the end).
class Stream {
public:
	Stream(const uint8_t *p, size_t n) : p_(p), n_(n) {}
	uint32_t u8() {
		const uint32_t v = at_ < n_ ? p_[at_] : 0u;
		if (at_ < n_) ++at_;
		return v;
	}
	uint32_t be16() { const uint32_t hi = u8(); return (hi << 8) | u8(); }
	uint32_t le16() { const uint32_t lo = u8(); return lo | (u8() << 8); }
	uint32_t be32() { const uint32_t hi = be16(); return (hi << 16) | be16(); }
	// a negative count (a 32-bit length with its top bit set) runs to the end of the data
	void skip(int64_t k) {
		if (k < 0 || (uint64_t)k > n_ - at_) at_ = n_;
		else at_ += (size_t)k;
	}
	bool drained() const { return at_ >= n_; } // every byte consumed (true right after the last one was read)
	// k bytes at once, or nothing at all when fewer are left (dst then keeps what it held)
	bool take(uint8_t *dst, size_t k) {
		if (k > n_ - at_) return false;
		std::memcpy(dst, p_ + at_, k);
		at_ += k;
		return true;
	}

private:
	const uint8_t *p_;
	size_t n_, at_ = 0;
};

bool refuse(std::string *err, const char *why) {
	*err = why;
	return false;
}

// Hand over an RGBA (or `have`-channel) canvas in the caller's channel count.
bool deliver(Image *img, std::vector<uint8_t> &px, int w, int h, int have, int file_comp, int req_comp) {
	img->w = w;
	img->h = h;
	img->comp_in_file = file_comp;
	const int want = req_comp ? req_comp : have;
	img->px = want == have ? std::move(px) : convert_channels8(px, have, want, (size_t)w * h);
	img->comp = want;
	return true;
}

bool size_fits(int w, int h, int comps) {
	return w > 0 && h > 0 && w <= kMaxSide && h <= kMaxSide && (uint64_t)w * (uint64_t)h * (uint64_t)comps < ((uint64_t)1 << 31);
}

// ================================================================== GIF ====
struct GifPalette {
	uint8_t rgb[256][3];
	bool opaque[256];
	void read(Stream &in, int entries, int transparent) {
		for (int i = 0; i < entries; ++i) {
			rgb[i][0] = (uint8_t)in.u8();
			rgb[i][1] = (uint8_t)in.u8();
			rgb[i][2] = (uint8_t)in.u8();
			opaque[i] = i != transparent;
		}
	}
};

// Where the next decoded index of an image lands: left to right, then down by the row step; interlaced images make
// four passes (rows 0, 8, 16..; 4, 12..; 2, 6..; 1, 3..).
struct GifRaster {
	int left, top, width, height; // the image's rectangle on the canvas
	bool interlaced;
	int col = 0, row = 0, pass = 0;
	bool exhausted() const { return width == 0 || row >= height; }
	void advance() {
		if (++col < width) return;
		col = 0;
		if (!interlaced) {
			++row;
			return;
		}
		static const int first[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2};
		row += step[pass];
		while (row >= height && pass < 3) {
			++pass;
			row = first[pass];
		}
	}
};

// Variable-width LZW as GIF uses it: codes of min_bits + 1 .. 12 bits, least significant bit first, packed in
// sub-blocks of up to 255 bytes; dictionary entries are (prefix code, last byte) pairs expanded through a stack.
class GifLzw {
public:
	explicit GifLzw(int min_bits) : min_bits_(min_bits), clear_(1 << min_bits) { reset(); }
	int clear_code() const { return clear_; }
	int end_code() const { return clear_ + 1; }
	int width() const { return width_; }
	void reset() {
		width_ = min_bits_ + 1;
		next_ = clear_ + 2;
		prev_ = -1;
	}
	// Takes one data code; appends the bytes it stands for to `out` (first byte first).  false = not decodable.
	bool take(int code, std::vector<uint8_t> &out, const char **why) {
		if (code > next_) { *why = "illegal code in raster"; return false; }
		if (prev_ >= 0) {
			if (next_ >= 8192) { *why = "too many codes"; return false; }
			prefix_[next_] = (int16_t)prev_;
			head_[next_] = head_of(prev_);
			// (for code == next_, the entry being made, this is its own head: the "KwKwK" case)
			tail_[next_] = code == next_ ? head_[next_] : head_of(code);
			++next_;
		} else if (code == next_) {
			*why = "illegal code in raster";
			return false;
		}
		// expand: follow the prefixes to a root, emit in forward order
		size_t at = out.size();
		int c = code, n = 0;
		for (; c >= clear_; c = prefix_[c]) ++n;
		out.resize(at + (size_t)n + 1);
		c = code;
		for (int k = n; k > 0; --k, c = prefix_[c]) out[at + (size_t)k] = tail_[c];
		out[at] = (uint8_t)c;
		if ((next_ & ((1 << width_) - 1)) == 0 && next_ <= 0x0fff) ++width_;
		prev_ = code;
		return true;
	}

private:
	uint8_t head_of(int code) const { return code < clear_ ? (uint8_t)code : head_[code]; }
	int min_bits_, clear_, width_ = 0, next_ = 0, prev_ = -1;
	int16_t prefix_[8192];
	uint8_t head_[8192], tail_[8192];
};

bool gif_first_image(Stream &in, int req_comp, Image *out, std::string *err) {
	if (in.u8() != 'G' || in.u8() != 'I' || in.u8() != 'F' || in.u8() != '8') return refuse(err, "not GIF");
	const uint32_t version = in.u8();
	if ((version != '7' && version != '9') || in.u8() != 'a') return refuse(err, "not GIF");
	const int W = (int)in.le16(), H = (int)in.le16();
	const uint32_t screen_flags = in.u8();
	const int background = (int)in.u8();
	(void)in.u8(); // pixel aspect ratio
	if (!size_fits(W, H, 4)) return refuse(err, W > 0 && H > 0 ? "too large" : "bad GIF size");
	GifPalette global{}, local{};
	const bool have_global = (screen_flags & 0x80) != 0;
	if (have_global) global.read(in, 2 << (screen_flags & 7), -1);

	int transparent = -1;
	bool use_transparent = false;
	for (;;) {
		const uint32_t tag = in.u8();
		if (tag == 0x21) { // extension
			const uint32_t label = in.u8();
			uint32_t len;
			if (label == 0xF9) { // graphic control: which index (if any) is transparent
				len = in.u8();
				if (len == 4) {
					const uint32_t gflags = in.u8();
					(void)in.le16(); // frame delay
					if (transparent >= 0) global.opaque[transparent] = true;
					use_transparent = (gflags & 1) != 0;
					if (use_transparent) {
						transparent = (int)in.u8();
						global.opaque[transparent] = false; // (pixel contract: the flag sticks to the GLOBAL table)
					} else {
						(void)in.u8();
						transparent = -1;
					}
				} else {
					in.skip(len);
					continue;
				}
			}
			while ((len = in.u8()) != 0) in.skip(len);
			continue;
		}
		if (tag == 0x3B) return refuse(err, "GIF holds no image");
		if (tag != 0x2C) return refuse(err, "unknown code");
		break;
	}

	GifRaster r{};
	r.left = (int)in.le16();
	r.top = (int)in.le16();
	r.width = (int)in.le16();
	r.height = (int)in.le16();
	if (r.left + r.width > W || r.top + r.height > H) return refuse(err, "bad Image Descriptor");
	const uint32_t image_flags = in.u8();
	r.interlaced = (image_flags & 0x40) != 0;
	const GifPalette *pal = nullptr;
	if (image_flags & 0x80) {
		local.read(in, 2 << (image_flags & 7), use_transparent ? transparent : -1);
		pal = &local;
	} else if (have_global) {
		pal = &global;
	} else {
		return refuse(err, "missing color table");
	}

	std::vector<uint8_t> canvas((size_t)W * H * 4, 0);
	std::vector<uint8_t> drawn((size_t)W * H, 0);
	const uint32_t min_bits = in.u8();
	if (min_bits > 12) return refuse(err, "bad GIF code size");
	GifLzw lzw((int)min_bits);
	std::vector<uint8_t> run;
	bool cleared = false, finished = false;
	uint32_t acc = 0, block_left = 0;
	int have_bits = 0;
	while (!finished) {
		if (have_bits < lzw.width()) {
			if (block_left == 0) {
				block_left = in.u8();
				if (block_left == 0) break; // block terminator (or the end of a truncated file)
			}
			--block_left;
			acc |= in.u8() << have_bits;
			have_bits += 8;
			continue;
		}
		const int code = (int)(acc & ((1u << lzw.width()) - 1u));
		acc >>= lzw.width();
		have_bits -= lzw.width();
		if (code == lzw.clear_code()) {
			lzw.reset();
			cleared = true;
		} else if (code == lzw.end_code()) {
			finished = true; // (what follows in the file does not matter: only this image is read)
		} else {
			if (!cleared) return refuse(err, "no clear code");
			const char *why = "";
			run.clear();
			if (!lzw.take(code, run, &why)) return refuse(err, why);
			for (uint8_t index : run) {
				if (r.exhausted()) break;
				const size_t at = (size_t)(r.top + r.row) * W + (size_t)(r.left + r.col);
				drawn[at] = 1;
				if (pal->opaque[index]) {
					uint8_t *px = &canvas[at * 4];
					px[0] = pal->rgb[index][0];
					px[1] = pal->rgb[index][1];
					px[2] = pal->rgb[index][2];
					px[3] = 255;
				}
				r.advance();
			}
		}
	}
	if (background > 0) { // pixel contract: untouched pixels take the background entry, blue first
		const uint8_t fill[4] = {global.rgb[background][2], global.rgb[background][1], global.rgb[background][0], 255};
		for (size_t i = 0; i < drawn.size(); ++i)
			if (!drawn[i]) std::memcpy(&canvas[i * 4], fill, 4);
	}
	return deliver(out, canvas, W, H, 4, 4, req_comp);
}

// ================================================================== PSD ====
// PackBits, one byte per pixel into every 4th byte of `dst`: n < 128 copies n + 1 bytes, n > 128 repeats the next
// byte 257 - n times, 128 does nothing.  false when a run overshoots the plane.
bool psd_unpack_plane(Stream &in, uint8_t *dst, int64_t pixels) {
	int64_t done = 0;
	while (done < pixels) {
		const int n = (int)in.u8();
		if (n == 128) continue;
		const int run = n < 128 ? n + 1 : 257 - n;
		if (run > pixels - done) return false;
		if (n < 128) {
			for (int k = 0; k < run; ++k) dst[(done + k) * 4] = (uint8_t)in.u8();
		} else {
			const uint8_t v = (uint8_t)in.u8();
			for (int k = 0; k < run; ++k) dst[(done + k) * 4] = v;
		}
		done += run;
	}
	return true;
}

// (uint8_t) of a float the way the reference's x86 build converts it: truncate to int32, keep the low byte.
inline uint8_t low_byte_of(float v) {
	int32_t i;
	if (!(v > -2147483904.0f && v < 2147483648.0f)) i = INT32_MIN; // (NaN / out of range: cvttss2si's indefinite value)
	else i = (int32_t)v;
	return (uint8_t)(uint32_t)i;
}

bool psd_composite(Stream &in, int req_comp, Image *out, std::string *err) {
	if (in.be32() != 0x38425053u) return refuse(err, "not PSD");
	if (in.be16() != 1) return refuse(err, "wrong version");
	in.skip(6);
	const int channels = (int)in.be16();
	if (channels > 16) return refuse(err, "wrong channel count");
	const uint32_t h32 = in.be32(), w32 = in.be32();
	if (h32 > (uint32_t)kMaxSide || w32 > (uint32_t)kMaxSide) return refuse(err, "too large");
	const int H = (int)h32, W = (int)w32;
	const int depth = (int)in.be16();
	if (depth != 8 && depth != 16) return refuse(err, "unsupported bit depth");
	if (in.be16() != 3) return refuse(err, "wrong color format"); // RGB mode only
	in.skip((int32_t)in.be32()); // colour mode data
	in.skip((int32_t)in.be32()); // image resources
	in.skip((int32_t)in.be32()); // layer and mask information
	const int compression = (int)in.be16();
	if (compression > 1) return refuse(err, "bad compression");
	if (!size_fits(W, H, 4)) return refuse(err, W > 0 && H > 0 ? "too large" : "bad PSD size");
	const int64_t pixels = (int64_t)W * H;
	std::vector<uint8_t> canvas((size_t)pixels * 4);
	if (compression) in.skip((int64_t)H * channels * 2); // the per-row byte counts are not needed
	for (int ch = 0; ch < 4; ++ch) {
		uint8_t *plane = canvas.data() + ch;
		if (ch >= channels) {
			for (int64_t i = 0; i < pixels; ++i) plane[i * 4] = ch == 3 ? 255 : 0;
		} else if (compression) {
			if (!psd_unpack_plane(in, plane, pixels)) return refuse(err, "bad RLE data");
		} else if (depth == 16) {
			for (int64_t i = 0; i < pixels; ++i) plane[i * 4] = (uint8_t)(in.be16() >> 8);
		} else {
			for (int64_t i = 0; i < pixels; ++i) plane[i * 4] = (uint8_t)in.u8();
		}
	}
	if (channels >= 4) { // colours were stored blended onto white: undo that under partial alpha
		for (int64_t i = 0; i < pixels; ++i) {
			uint8_t *px = &canvas[(size_t)i * 4];
			if (px[3] == 0 || px[3] == 255) continue;
			const float a = (float)px[3] / 255.0f;
			const float ra = 1.0f / a;
			const float shift = 255.0f * (1.0f - ra);
			for (int k = 0; k < 3; ++k) px[k] = low_byte_of((float)px[k] * ra + shift);
		}
	}
	return deliver(out, canvas, W, H, 4, 4, req_comp);
}

// ================================================================== PIC ====
struct PicPacket {
	uint8_t type = 0, channels = 0; // 0 raw, 1 pure run-length, 2 mixed; channel mask 0x80 R, 0x40 G, 0x20 B, 0x10 A
};

// One pixel's worth of the packet's channels, straight from the stream.  false at the end of the data.
bool pic_fetch(Stream &in, uint8_t mask, uint8_t *px) {
	for (int k = 0; k < 4; ++k)
		if (mask & (0x80 >> k)) {
			if (in.drained()) return false;
			px[k] = (uint8_t)in.u8();
		}
	return true;
}
inline void pic_store(uint8_t mask, uint8_t *px, const uint8_t *v) {
	for (int k = 0; k < 4; ++k)
		if (mask & (0x80 >> k)) px[k] = v[k];
}

bool pic_picture(Stream &in, int req_comp, Image *out, std::string *err) {
	in.skip(92); // magic, version, comment, "PICT"
	const int W = (int)in.be16(), H = (int)in.be16();
	if (in.drained()) return refuse(err, "PIC file too short");
	if (!size_fits(W, H, 4)) return refuse(err, W > 0 && H > 0 ? "too large" : "bad PIC size");
	in.skip(8); // aspect ratio, fields, padding
	std::vector<PicPacket> packets;
	uint32_t seen = 0;
	for (bool more = true; more;) {
		if (packets.size() == 10) return refuse(err, "too many packets");
		more = in.u8() != 0;
		const uint32_t bits = in.u8();
		PicPacket p;
		p.type = (uint8_t)in.u8();
		p.channels = (uint8_t)in.u8();
		seen |= p.channels;
		if (in.drained()) return refuse(err, "PIC file too short");
		if (bits != 8) return refuse(err, "packet isn't 8bpp");
		packets.push_back(p);
	}
	const int file_comp = (seen & 0x10) ? 4 : 3;
	std::vector<uint8_t> canvas((size_t)W * H * 4, 0xff);
	for (int y = 0; y < H; ++y) {
		for (const PicPacket &p : packets) {
			uint8_t *px = &canvas[(size_t)y * W * 4];
			int left = W;
			if (p.type == 0) {
				for (; left > 0; --left, px += 4)
					if (!pic_fetch(in, p.channels, px)) return refuse(err, "PIC file too short");
			} else if (p.type == 1) {
				while (left > 0) {
					int count = (int)in.u8();
					if (in.drained()) return refuse(err, "PIC file too short");
					if (count > left) count = left;
					uint8_t v[4];
					if (!pic_fetch(in, p.channels, v)) return refuse(err, "PIC file too short");
					for (int k = 0; k < count; ++k, px += 4) pic_store(p.channels, px, v);
					left -= count;
				}
			} else if (p.type == 2) {
				while (left > 0) {
					int count = (int)in.u8();
					if (in.drained()) return refuse(err, "PIC file too short");
					if (count >= 128) { // a run: 129..255 -> 2..128 pixels, 128 -> a 16-bit count follows
						count = count == 128 ? (int)in.be16() : count - 127;
						if (count > left) return refuse(err, "scanline overrun");
						uint8_t v[4];
						if (!pic_fetch(in, p.channels, v)) return refuse(err, "PIC file too short");
						for (int k = 0; k < count; ++k, px += 4) pic_store(p.channels, px, v);
					} else { // count + 1 literal pixels
						++count;
						if (count > left) return refuse(err, "scanline overrun");
						for (int k = 0; k < count; ++k, px += 4)
							if (!pic_fetch(in, p.channels, px)) return refuse(err, "PIC file too short");
					}
					left -= count;
				}
			} else {
				return refuse(err, "packet has bad compression type");
			}
		}
	}
	// (the canvas is RGBA; a file without an alpha packet reports 3 channels and, unless the caller asks
	// otherwise, is handed over as 3)
	out->w = W;
	out->h = H;
	out->comp_in_file = file_comp;
	const int want = req_comp ? req_comp : file_comp;
	out->px = want == 4 ? std::move(canvas) : convert_channels8(canvas, 4, want, (size_t)W * H);
	out->comp = want;
	return true;
}

// ================================================================== HDR ====
// A header line (pixel contract: at most 1023 characters are kept; a final character that ends the data without
// a newline is dropped).
std::string hdr_line(Stream &in) {
	std::string s;
	char c = (char)in.u8();
	while (!in.drained() && c != '\n') {
		s.push_back(c);
		if (s.size() == 1023) {
			while (!in.drained() && in.u8() != '\n') {
			}
			break;
		}
		c = (char)in.u8();
	}
	return s;
}

// One RGBE pixel to `want` linear floats: 1 grey, 2 grey + 1, 3 rgb, 4 rgb + 1 (single precision throughout).
void hdr_expand(const uint8_t rgbe[4], int want, float *dst) {
	float r = 0.0f, g = 0.0f, b = 0.0f, grey = 0.0f;
	if (rgbe[3] != 0) {
		const float scale = (float)std::ldexp(1.0f, (int)rgbe[3] - 136);
		if (want <= 2) {
			grey = (float)((int)rgbe[0] + (int)rgbe[1] + (int)rgbe[2]) * scale / 3.0f;
		} else {
			r = (float)rgbe[0] * scale;
			g = (float)rgbe[1] * scale;
			b = (float)rgbe[2] * scale;
		}
	}
	if (want <= 2) {
		dst[0] = grey;
		if (want == 2) dst[1] = 1.0f;
	} else {
		dst[0] = r; dst[1] = g; dst[2] = b;
		if (want == 4) dst[3] = 1.0f;
	}
}

inline uint8_t hdr_byte(float z) {
	if (z < 0.0f) z = 0.0f;
	if (z > 255.0f) z = 255.0f;
	return low_byte_of(z);
}

bool hdr_picture(Stream &in, int req_comp, Image *out, std::string *err) {
	const std::string magic = hdr_line(in);
	if (magic != "#?RADIANCE" && magic != "#?RGBE") return refuse(err, "not HDR");
	bool rle_rgbe = false;
	for (;;) {
		const std::string line = hdr_line(in);
		if (line.empty()) break;
		if (line == "FORMAT=32-bit_rle_rgbe") rle_rgbe = true;
	}
	if (!rle_rgbe) return refuse(err, "unsupported format");
	const std::string dims = hdr_line(in);
	if (dims.compare(0, 3, "-Y ") != 0) return refuse(err, "unsupported data layout");
	char *rest = nullptr;
	// (pixel contract: the loader this replaces narrows strtol's long to int BEFORE its range check, so a size of 2^32 + 5
	// reads as 5 there; the same here -- the unsigned detour keeps the narrowing defined)
	const int H = (int)(unsigned int)(unsigned long)std::strtol(dims.c_str() + 3, &rest, 10);
	while (*rest == ' ') ++rest;
	if (std::strncmp(rest, "+X ", 3) != 0) return refuse(err, "unsupported data layout");
	const int W = (int)(unsigned int)(unsigned long)std::strtol(rest + 3, nullptr, 10);
	if (H > kMaxSide || W > kMaxSide) return refuse(err, "too large");
	const int want = req_comp ? req_comp : 3;
	if (W <= 0 || H <= 0) return refuse(err, "bad HDR size");
	if ((uint64_t)W * (uint64_t)H * (uint64_t)want * sizeof(float) >= ((uint64_t)1 << 31)) return refuse(err, "too large");

	std::vector<float> lin((size_t)W * H * want);
	auto flat_from = [&](size_t first_pixel) { // the rest of the image as 4 bytes per pixel
		// (pixel contract: a pixel is read whole or not at all; past the end of a truncated file the last
		// complete pixel repeats)
		uint8_t rgbe[4] = {0, 0, 0, 0};
		for (size_t i = first_pixel; i < (size_t)W * H; ++i) {
			(void)in.take(rgbe, 4);
			hdr_expand(rgbe, want, &lin[i * want]);
		}
	};
	if (W < 8 || W >= 32768) {
		flat_from(0);
	} else {
		std::vector<uint8_t> row((size_t)W * 4);
		for (int y = 0; y < H; ++y) {
			const uint32_t c1 = in.u8(), c2 = in.u8(), hi = in.u8();
			if (c1 != 2 || c2 != 2 || (hi & 0x80)) {
				// Not a run-length scanline.  Pixel contract: these four bytes become the image's FIRST pixel and
				// everything from the second pixel on is read flat from here, wherever this happened.
				const uint8_t rgbe[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)hi, (uint8_t)in.u8()};
				hdr_expand(rgbe, want, &lin[0]);
				flat_from(1);
				break;
			}
			if ((int)((hi << 8) | in.u8()) != W) return refuse(err, "invalid decoded scanline length");
			for (int plane = 0; plane < 4; ++plane) { // R, G, B and E of the row, each run-length coded
				int x = 0;
				while (x < W) {
					uint32_t count = in.u8();
					const bool run = count > 128;
					if (run) count -= 128;
					if ((int)count > W - x) return refuse(err, "bad RLE data in HDR");
					// (a count of zero makes no progress: at the end of a truncated file that would never end)
					if (count == 0 && in.drained()) return refuse(err, "bad RLE data in HDR");
					if (run) {
						const uint8_t v = (uint8_t)in.u8();
						for (uint32_t k = 0; k < count; ++k) row[(size_t)(x++) * 4 + plane] = v;
					} else {
						for (uint32_t k = 0; k < count; ++k) row[(size_t)(x++) * 4 + plane] = (uint8_t)in.u8();
					}
				}
			}
			for (int x = 0; x < W; ++x) hdr_expand(&row[(size_t)x * 4], want, &lin[((size_t)y * W + x) * want]);
		}
	}
	// tone mapping to 8 bits: gamma 2.2 on the colour channels, the constant alpha stays linear
	const int colour = (want & 1) ? want : want - 1;
	const float inv_gamma = 1.0f / 2.2f, unit = 1.0f;
	std::vector<uint8_t> px((size_t)W * H * want);
	for (size_t i = 0; i < (size_t)W * H; ++i) {
		for (int k = 0; k < colour; ++k) {
			const float v = lin[i * want + k] * unit;
			px[i * want + k] = hdr_byte((float)std::pow((double)v, (double)inv_gamma) * 255.0f + 0.5f);
		}
		if (colour < want) px[i * want + colour] = hdr_byte(lin[i * want + colour] * 255.0f + 0.5f);
	}
	out->w = W;
	out->h = H;
	out->comp_in_file = 3;
	out->comp = want;
	out->px = std::move(px);
	return true;
}

} // namespace

bool looks_like_gif(const uint8_t *b, size_t n) {
	return n >= 6 && std::memcmp(b, "GIF8", 4) == 0 && (b[4] == '7' || b[4] == '9') && b[5] == 'a';
}
bool looks_like_psd(const uint8_t *b, size_t n) { return n >= 4 && std::memcmp(b, "8BPS", 4) == 0; }
bool looks_like_pic(const uint8_t *b, size_t n) {
	return n >= 92 && b[0] == 0x53 && b[1] == 0x80 && b[2] == 0xF6 && b[3] == 0x34 && std::memcmp(b + 88, "PICT", 4) == 0;
}
bool looks_like_hdr(const uint8_t *b, size_t n) {
	return (n >= 11 && std::memcmp(b, "#?RADIANCE\n", 11) == 0) || (n >= 7 && std::memcmp(b, "#?RGBE\n", 7) == 0);
}

bool decode_gif(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Stream in(bytes, len);
	return gif_first_image(in, req_comp, out, err);
}
bool decode_psd(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Stream in(bytes, len);
	return psd_composite(in, req_comp, out, err);
}
bool decode_pic(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Stream in(bytes, len);
	return pic_picture(in, req_comp, out, err);
}
bool decode_hdr(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Stream in(bytes, len);
	return hdr_picture(in, req_comp, out, err);
}

} // namespace hmrm
