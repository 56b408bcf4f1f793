// more_decoders.cpp -- GIF (first frame), Photoshop PSD, Softimage PIC and Radiance HDR,
// written for this project: the remaining formats the reference accepts for its maps
// through stbi_load (main/hmap.cpp:320-321, 341-342; README "Options").  Conventions
// follow stb_image v2.27 so that the decoded 8-bit pixels are the same (checked against the
// reference's own stb build in tests/test_image_io.py):
//   GIF (vendor/stb_image.h:6476-7000): first frame only; 4 channels; pixels the frame does
//     not draw stay (0,0,0,0) unless the background index is > 0, in which case they take the
//     background palette entry in its STORED byte order (B,G,R) with alpha 255 -- stb's
//     quirk, kept; a transparent index (graphic control extension) is not drawn; interlace;
//     LZW with at most 8192 codes; a stream without a clear code first is refused;
//   PSD (:5998-6250): version 1, RGB mode, 8 or 16 bits (high byte kept), raw or PackBits
//     planes, missing channels read 0 (alpha 255), "white matte" removed where 0 < a < 255
//     with stb's float arithmetic;
//   PIC (:6252-6470): 8-bit packets, uncompressed / pure RLE / mixed RLE, channels by mask,
//     untouched channels stay 255;
//   HDR (:7002-7190 + :1864-1890): "#?RADIANCE" / "#?RGBE", FORMAT=32-bit_rle_rgbe, "-Y h +X w",
//     flat or new-style RLE scanlines, RGBE -> float -> 8 bit with gamma 1/2.2f through the
//     host's double-precision pow, + 0.5f, clamped, truncated.
// A read past the end of the data gives zeros, as in stb's reader.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "image_io.hpp"

namespace hmrm {
namespace {

constexpr int kMaxDim = 1 << 24; // STBI_MAX_DIMENSIONS

struct Reader {
	const uint8_t *base, *p, *end;
	bool at_eof() const { return p >= end; }
	int get8() { return p < end ? *p++ : 0; }
	int get16le() { int a = get8(); return a | (get8() << 8); }
	int get16be() { int a = get8(); return (a << 8) | get8(); }
	uint32_t get32be() { uint32_t a = (uint32_t)get16be(); return (a << 16) | (uint32_t)get16be(); }
	void skip(int n) {
		if (n < 0) { p = end; return; }
		p = (end - p) < n ? end : p + n;
	}
};

// a*b*c fits an int (stb's stbi__mad3sizes_valid with add = 0), all factors >= 0
bool sizes_ok(int64_t a, int64_t b, int64_t c) {
	if (a < 0 || b < 0 || c < 0) return false;
	if (a == 0 || b == 0 || c == 0) return true;
	if (a > 0x7fffffffLL / b) return false;
	return a * b <= 0x7fffffffLL / c;
}

// Not in stb: a header that claims millions of pixels over a few bytes of data (no run-length
// scheme here packs better than ~1400:1) is refused before anything of that size is allocated.
bool plausible(const Reader &s, int64_t w, int64_t h) {
	const int64_t pixels = w * h, left = (int64_t)(s.end - s.p);
	return pixels <= (1 << 20) || left >= pixels / 4096;
}

bool finish(std::vector<uint8_t> &&px, int w, int h, int have, int file_comp, int req_comp, Image *out, std::string *err) {
	if (w <= 0 || h <= 0) { *err = "empty image"; return false; }
	out->w = w;
	out->h = h;
	out->comp_in_file = file_comp;
	const int want = req_comp ? req_comp : have;
	out->comp = want;
	out->px = want == have ? std::move(px) : convert_channels8(px, have, want, (size_t)w * (size_t)h);
	return true;
}

// ------------------------------------------------------------------------ GIF --
struct GifCode {
	int16_t prefix;
	uint8_t first, suffix;
};

struct Gif {
	int w = 0, h = 0, flags = 0, bgindex = 0, transparent = -1, eflags = 0, lflags = 0;
	uint8_t pal[256][4], lpal[256][4]; // stored B,G,R,A
	const uint8_t *color_table = nullptr;
	std::vector<uint8_t> out, history;
	GifCode codes[8192];
	int parse = 0, step = 0, start_x = 0, start_y = 0, max_x = 0, max_y = 0, cur_x = 0, cur_y = 0, line_size = 0;
};

void gif_palette(Reader &s, uint8_t pal[256][4], int n, int transparent) {
	for (int i = 0; i < n; ++i) {
		pal[i][2] = (uint8_t)s.get8();
		pal[i][1] = (uint8_t)s.get8();
		pal[i][0] = (uint8_t)s.get8();
		pal[i][3] = transparent == i ? 0 : 255;
	}
}

// one decoded string, first symbol first (the code table links backwards)
void gif_emit(Gif &g, int code) {
	uint8_t stack[8192];
	int n = 0;
	for (int c = code; c >= 0 && n < 8192; c = g.codes[c].prefix) stack[n++] = g.codes[c].suffix;
	while (n > 0) {
		const uint8_t sym = stack[--n];
		if (g.cur_y >= g.max_y) continue; // (stb keeps walking the string; nothing more is drawn)
		const int idx = g.cur_x + g.cur_y;
		g.history[(size_t)idx / 4] = 1;
		const uint8_t *c = g.color_table + (size_t)sym * 4;
		if (c[3] > 128) {
			uint8_t *p = &g.out[(size_t)idx];
			p[0] = c[2];
			p[1] = c[1];
			p[2] = c[0];
			p[3] = c[3];
		}
		g.cur_x += 4;
		if (g.cur_x >= g.max_x) {
			g.cur_x = g.start_x;
			g.cur_y += g.step;
			while (g.cur_y >= g.max_y && g.parse > 0) { // next interlace pass
				g.step = (1 << g.parse) * g.line_size;
				g.cur_y = g.start_y + (g.step >> 1);
				--g.parse;
			}
		}
	}
}

bool gif_raster(Reader &s, Gif &g, std::string *err) {
	const int lzw_cs = s.get8();
	if (lzw_cs > 12) { *err = ""; return false; } // (stb fails here without a reason of its own)
	const int clear = 1 << lzw_cs;
	bool first = true;
	int codesize = lzw_cs + 1, codemask = (1 << codesize) - 1;
	int32_t bits = 0;
	int valid_bits = 0;
	for (int c = 0; c < clear; ++c) {
		g.codes[c].prefix = -1;
		g.codes[c].first = (uint8_t)c;
		g.codes[c].suffix = (uint8_t)c;
	}
	int avail = clear + 2, oldcode = -1, len = 0;
	for (;;) {
		if (valid_bits < codesize) {
			if (len == 0) {
				len = s.get8(); // next data sub-block
				if (len == 0) return true;
			}
			--len;
			bits |= (int32_t)((uint32_t)s.get8() << valid_bits);
			valid_bits += 8;
			continue;
		}
		const int code = bits & codemask;
		bits >>= codesize;
		valid_bits -= codesize;
		if (code == clear) {
			codesize = lzw_cs + 1;
			codemask = (1 << codesize) - 1;
			avail = clear + 2;
			oldcode = -1;
			first = false;
		} else if (code == clear + 1) { // end of information: swallow the rest of the sub-blocks
			s.skip(len);
			while ((len = s.get8()) > 0) s.skip(len);
			return true;
		} else if (code <= avail) {
			if (first) { *err = "no clear code"; return false; }
			if (oldcode >= 0) {
				GifCode *p = &g.codes[avail++];
				if (avail > 8192) { *err = "too many codes"; return false; }
				p->prefix = (int16_t)oldcode;
				p->first = g.codes[oldcode].first;
				p->suffix = (code == avail) ? p->first : g.codes[code].first;
			} else if (code == avail) {
				*err = "illegal code in raster";
				return false;
			}
			gif_emit(g, code);
			if ((avail & codemask) == 0 && avail <= 0x0fff) {
				++codesize;
				codemask = (1 << codesize) - 1;
			}
			oldcode = code;
		} else {
			*err = "illegal code in raster";
			return false;
		}
	}
}

} // namespace

bool decode_gif(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Reader s{bytes, bytes, bytes + len};
	if (s.get8() != 'G' || s.get8() != 'I' || s.get8() != 'F' || s.get8() != '8') { *err = "not GIF"; return false; }
	const int version = s.get8();
	if (version != '7' && version != '9') { *err = "not GIF"; return false; }
	if (s.get8() != 'a') { *err = "not GIF"; return false; }
	std::vector<Gif> holder(1); // (large: keep it off the stack)
	Gif &g = holder[0];
	g.w = s.get16le();
	g.h = s.get16le();
	g.flags = s.get8();
	g.bgindex = s.get8();
	(void)s.get8(); // aspect ratio
	memset(g.pal, 0, sizeof g.pal); // entries a short table does not define read as zeros
	memset(g.lpal, 0, sizeof g.lpal);
	if (g.flags & 0x80) gif_palette(s, g.pal, 2 << (g.flags & 7), -1);
	if (!sizes_ok(4, g.w, g.h) || !plausible(s, g.w, g.h)) { *err = "too large"; return false; }
	const size_t pcount = (size_t)g.w * (size_t)g.h;
	g.out.assign(pcount * 4, 0);
	g.history.assign(pcount, 0);

	for (;;) {
		const int tag = s.get8();
		if (tag == 0x2c) { // image descriptor
			const int x = s.get16le(), y = s.get16le(), w = s.get16le(), h = s.get16le();
			if (x + w > g.w || y + h > g.h) { *err = "bad Image Descriptor"; return false; }
			g.line_size = g.w * 4;
			g.start_x = x * 4;
			g.start_y = y * g.line_size;
			g.max_x = g.start_x + w * 4;
			g.max_y = g.start_y + h * g.line_size;
			g.cur_x = g.start_x;
			g.cur_y = w == 0 ? g.max_y : g.start_y; // an empty rectangle draws nothing
			g.lflags = s.get8();
			if (g.lflags & 0x40) {
				g.step = 8 * g.line_size; // first interlace pass
				g.parse = 3;
			} else {
				g.step = g.line_size;
				g.parse = 0;
			}
			if (g.lflags & 0x80) {
				gif_palette(s, g.lpal, 2 << (g.lflags & 7), (g.eflags & 0x01) ? g.transparent : -1);
				g.color_table = &g.lpal[0][0];
			} else if (g.flags & 0x80) {
				g.color_table = &g.pal[0][0];
			} else {
				*err = "missing color table";
				return false;
			}
			if (!gif_raster(s, g, err)) return false;
			if (g.bgindex > 0) { // pixels the first frame did not draw take the background entry as stored
				g.pal[g.bgindex][3] = 255;
				for (size_t pi = 0; pi < pcount; ++pi)
					if (g.history[pi] == 0) memcpy(&g.out[pi * 4], g.pal[g.bgindex], 4);
			}
			return finish(std::move(g.out), g.w, g.h, 4, 4, req_comp, out, err);
		} else if (tag == 0x21) { // extension
			const int ext = s.get8();
			if (ext == 0xf9) { // graphic control
				const int n = s.get8();
				if (n == 4) {
					g.eflags = s.get8();
					(void)s.get16le(); // delay
					if (g.transparent >= 0) g.pal[g.transparent][3] = 255;
					if (g.eflags & 0x01) {
						g.transparent = s.get8();
						g.pal[g.transparent][3] = 0;
					} else {
						s.skip(1);
						g.transparent = -1;
					}
				} else {
					s.skip(n);
					continue; // (stb leaves the following sub-blocks to the tag loop)
				}
			}
			int n;
			while ((n = s.get8()) != 0) s.skip(n);
		} else if (tag == 0x3b) {
			*err = "no image in GIF";
			return false;
		} else {
			*err = "unknown code";
			return false;
		}
	}
}

// ------------------------------------------------------------------------ PSD --
namespace {
// PackBits into every fourth byte of p
bool psd_rle(Reader &s, uint8_t *p, int pixel_count) {
	int count = 0, nleft;
	while ((nleft = pixel_count - count) > 0) {
		int n = s.get8();
		if (n == 128) continue;
		if (n < 128) {
			++n;
			if (n > nleft) return false;
			count += n;
			for (; n; --n, p += 4) *p = (uint8_t)s.get8();
		} else {
			n = 257 - n;
			if (n > nleft) return false;
			const uint8_t v = (uint8_t)s.get8();
			count += n;
			for (; n; --n, p += 4) *p = v;
		}
	}
	return true;
}
} // namespace

bool decode_psd(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Reader s{bytes, bytes, bytes + len};
	if (s.get32be() != 0x38425053u) { *err = "not PSD"; return false; }
	if (s.get16be() != 1) { *err = "wrong version"; return false; }
	s.skip(6);
	const int channel_count = s.get16be();
	if (channel_count < 0 || channel_count > 16) { *err = "wrong channel count"; return false; }
	const int h = (int)s.get32be(), w = (int)s.get32be();
	if (h > kMaxDim || w > kMaxDim) { *err = "too large"; return false; }
	const int bitdepth = s.get16be();
	if (bitdepth != 8 && bitdepth != 16) { *err = "unsupported bit depth"; return false; }
	if (s.get16be() != 3) { *err = "wrong color format"; return false; }
	s.skip((int)s.get32be()); // mode data
	s.skip((int)s.get32be()); // image resources
	s.skip((int)s.get32be()); // layer and mask information
	const int compression = s.get16be();
	if (compression > 1) { *err = "bad compression"; return false; }
	if (!sizes_ok(4, w, h) || !plausible(s, w, h)) { *err = "too large"; return false; }
	if (w <= 0 || h <= 0) { *err = "empty image"; return false; }
	const int pixel_count = w * h;
	std::vector<uint8_t> px((size_t)pixel_count * 4);
	if (compression) {
		s.skip(h * channel_count * 2); // per-row byte counts
		for (int ch = 0; ch < 4; ++ch) {
			uint8_t *p = px.data() + ch;
			if (ch >= channel_count) {
				for (int i = 0; i < pixel_count; ++i, p += 4) *p = ch == 3 ? 255 : 0;
			} else if (!psd_rle(s, p, pixel_count)) {
				*err = "corrupt";
				return false;
			}
		}
	} else {
		for (int ch = 0; ch < 4; ++ch) {
			uint8_t *p = px.data() + ch;
			if (ch >= channel_count) {
				for (int i = 0; i < pixel_count; ++i, p += 4) *p = ch == 3 ? 255 : 0;
			} else if (bitdepth == 16) {
				for (int i = 0; i < pixel_count; ++i, p += 4) *p = (uint8_t)(s.get16be() >> 8);
			} else {
				for (int i = 0; i < pixel_count; ++i, p += 4) *p = (uint8_t)s.get8();
			}
		}
	}
	if (channel_count >= 4) { // remove the white matte; float arithmetic and int conversion as in stb's x86 build
		for (int i = 0; i < pixel_count; ++i) {
			uint8_t *pixel = px.data() + 4 * (size_t)i;
			if (pixel[3] != 0 && pixel[3] != 255) {
				const float a = pixel[3] / 255.0f;
				const float ra = 1.0f / a;
				const float inv_a = 255.0f * (1 - ra);
				pixel[0] = (uint8_t)(int)(pixel[0] * ra + inv_a);
				pixel[1] = (uint8_t)(int)(pixel[1] * ra + inv_a);
				pixel[2] = (uint8_t)(int)(pixel[2] * ra + inv_a);
			}
		}
	}
	return finish(std::move(px), w, h, 4, 4, req_comp, out, err);
}

// ------------------------------------------------------------------------ PIC --
namespace {
bool pic_is4(Reader &s, const char *str) {
	for (int i = 0; i < 4; ++i)
		if (s.get8() != (uint8_t)str[i]) return false;
	return true;
}
bool pic_readval(Reader &s, int channel, uint8_t *dest, std::string *err) {
	for (int i = 0, mask = 0x80; i < 4; ++i, mask >>= 1)
		if (channel & mask) {
			if (s.at_eof()) { *err = "bad file"; return false; }
			dest[i] = (uint8_t)s.get8();
		}
	return true;
}
void pic_copyval(int channel, uint8_t *dest, const uint8_t *src) {
	for (int i = 0, mask = 0x80; i < 4; ++i, mask >>= 1)
		if (channel & mask) dest[i] = src[i];
}
} // namespace

bool looks_like_pic(const uint8_t *bytes, size_t len) {
	Reader s{bytes, bytes, bytes + len};
	if (!pic_is4(s, "\x53\x80\xF6\x34")) return false;
	s.skip(84);
	return pic_is4(s, "PICT");
}

bool decode_pic(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Reader s{bytes, bytes, bytes + len};
	s.skip(92);
	const int w = s.get16be(), h = s.get16be();
	if (s.at_eof()) { *err = "bad file"; return false; }
	if (!sizes_ok(w, h, 4) || !plausible(s, w, h)) { *err = "too large"; return false; }
	(void)s.get32be(); // ratio
	(void)s.get16be(); // fields
	(void)s.get16be(); // pad
	if (w <= 0 || h <= 0) { *err = "empty image"; return false; }
	std::vector<uint8_t> px((size_t)w * (size_t)h * 4, 0xff);

	struct Packet { uint8_t size, type, channel; } packets[10];
	int num_packets = 0, act_comp = 0, chained;
	do {
		if (num_packets == 10) { *err = "bad format"; return false; }
		Packet &pk = packets[num_packets++];
		chained = s.get8();
		pk.size = (uint8_t)s.get8();
		pk.type = (uint8_t)s.get8();
		pk.channel = (uint8_t)s.get8();
		act_comp |= pk.channel;
		if (s.at_eof()) { *err = "bad file"; return false; }
		if (pk.size != 8) { *err = "bad format"; return false; }
	} while (chained);
	const int file_comp = (act_comp & 0x10) ? 4 : 3;

	for (int y = 0; y < h; ++y) {
		for (int k = 0; k < num_packets; ++k) {
			const Packet &pk = packets[k];
			uint8_t *dest = px.data() + (size_t)y * (size_t)w * 4;
			if (pk.type == 0) { // uncompressed
				for (int x = 0; x < w; ++x, dest += 4)
					if (!pic_readval(s, pk.channel, dest, err)) return false;
			} else if (pk.type == 1) { // pure RLE
				int left = w;
				while (left > 0) {
					uint8_t value[4];
					int count = s.get8();
					if (s.at_eof()) { *err = "bad file"; return false; }
					if (count > left) count = (uint8_t)left;
					if (!pic_readval(s, pk.channel, value, err)) return false;
					for (int i = 0; i < count; ++i, dest += 4) pic_copyval(pk.channel, dest, value);
					left -= count;
				}
			} else if (pk.type == 2) { // mixed RLE
				int left = w;
				while (left > 0) {
					int count = s.get8();
					if (s.at_eof()) { *err = "bad file"; return false; }
					if (count >= 128) { // repeated
						uint8_t value[4];
						if (count == 128) count = s.get16be();
						else count -= 127;
						if (count > left) { *err = "bad file"; return false; }
						if (!pic_readval(s, pk.channel, value, err)) return false;
						for (int i = 0; i < count; ++i, dest += 4) pic_copyval(pk.channel, dest, value);
					} else { // raw
						++count;
						if (count > left) { *err = "bad file"; return false; }
						for (int i = 0; i < count; ++i, dest += 4)
							if (!pic_readval(s, pk.channel, dest, err)) return false;
					}
					left -= count;
				}
			} else {
				*err = "bad format";
				return false;
			}
		}
	}
	return finish(std::move(px), w, h, 4, file_comp, req_comp ? req_comp : file_comp, out, err);
}

// ------------------------------------------------------------------------ HDR --
namespace {
std::string hdr_token(Reader &s) {
	std::string tok;
	char c = (char)s.get8();
	while (!s.at_eof() && c != '\n') {
		tok.push_back(c);
		if (tok.size() == 1023) { // over-long line: drop the rest of it
			while (!s.at_eof() && s.get8() != '\n') {}
			break;
		}
		c = (char)s.get8();
	}
	return tok;
}
void hdr_convert(float *output, const uint8_t *input, int req_comp) {
	if (input[3] != 0) {
		const float f1 = (float)std::ldexp(1.0f, input[3] - (int)(128 + 8));
		if (req_comp <= 2) {
			output[0] = (input[0] + input[1] + input[2]) * f1 / 3;
		} else {
			output[0] = input[0] * f1;
			output[1] = input[1] * f1;
			output[2] = input[2] * f1;
		}
		if (req_comp == 2) output[1] = 1;
		if (req_comp == 4) output[3] = 1;
	} else {
		switch (req_comp) {
		case 4: output[3] = 1; /* fallthrough */
		case 3: output[0] = output[1] = output[2] = 0; break;
		case 2: output[1] = 1; /* fallthrough */
		case 1: output[0] = 0; break;
		default: break;
		}
	}
}
} // namespace

bool looks_like_hdr(const uint8_t *bytes, size_t len) {
	return (len >= 11 && memcmp(bytes, "#?RADIANCE\n", 11) == 0) || (len >= 7 && memcmp(bytes, "#?RGBE\n", 7) == 0);
}

bool decode_hdr(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err) {
	Reader s{bytes, bytes, bytes + len};
	const std::string magic = hdr_token(s);
	if (magic != "#?RADIANCE" && magic != "#?RGBE") { *err = "not HDR"; return false; }
	bool valid = false;
	for (;;) {
		const std::string tok = hdr_token(s);
		if (tok.empty()) break;
		if (tok == "FORMAT=32-bit_rle_rgbe") valid = true;
	}
	if (!valid) { *err = "unsupported format"; return false; }
	const std::string dims = hdr_token(s);
	const char *t = dims.c_str();
	if (strncmp(t, "-Y ", 3)) { *err = "unsupported data layout"; return false; }
	t += 3;
	char *endp = nullptr;
	const int height = (int)strtol(t, &endp, 10);
	t = endp;
	while (*t == ' ') ++t;
	if (strncmp(t, "+X ", 3)) { *err = "unsupported data layout"; return false; }
	t += 3;
	const int width = (int)strtol(t, nullptr, 10);
	if (height > kMaxDim || width > kMaxDim) { *err = "too large"; return false; }
	const int comp = req_comp ? req_comp : 3;
	if (width <= 0 || height <= 0) { *err = "empty image"; return false; }
	if (!sizes_ok(width, height, (int64_t)comp * 4) || !plausible(s, width, height)) { *err = "too large"; return false; }
	std::vector<float> hdr((size_t)width * (size_t)height * (size_t)comp, 0.0f);

	auto flat_from = [&](int j0, int i0) { // flat RGBE pixels from (row j0, column i0) on
		for (int j = j0; j < height; ++j)
			for (int i = (j == j0 ? i0 : 0); i < width; ++i) {
				uint8_t rgbe[4] = {0, 0, 0, 0};
				if (s.end - s.p >= 4) { memcpy(rgbe, s.p, 4); s.p += 4; } // (stb leaves rgbe unset on a short read)
				hdr_convert(&hdr[((size_t)j * width + i) * comp], rgbe, comp);
			}
	};
	if (width < 8 || width >= 32768) {
		flat_from(0, 0);
	} else {
		std::vector<uint8_t> scanline((size_t)width * 4);
		for (int j = 0; j < height; ++j) {
			const int c1 = s.get8(), c2 = s.get8();
			int n = s.get8();
			if (c1 != 2 || c2 != 2 || (n & 0x80)) {
				// not run-length encoded: these four bytes are the first pixel, and (as in stb) the
				// decode restarts in flat mode from the top of the image
				uint8_t rgbe[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)n, (uint8_t)s.get8()};
				hdr_convert(&hdr[0], rgbe, comp);
				flat_from(0, 1);
				break;
			}
			n = (n << 8) | s.get8();
			if (n != width) { *err = "invalid decoded scanline length"; return false; }
			for (int k = 0; k < 4; ++k) {
				int i = 0, nleft;
				while ((nleft = width - i) > 0) {
					int count = s.get8();
					if (count == 0) { *err = "corrupt"; return false; } // (stb v2.27 spins forever on a zero count)
					if (count > 128) { // run
						const uint8_t value = (uint8_t)s.get8();
						count -= 128;
						if (count > nleft) { *err = "corrupt"; return false; }
						for (int z = 0; z < count; ++z) scanline[(size_t)(i++) * 4 + k] = value;
					} else { // literal bytes
						if (count > nleft) { *err = "corrupt"; return false; }
						for (int z = 0; z < count; ++z) scanline[(size_t)(i++) * 4 + k] = (uint8_t)s.get8();
					}
				}
			}
			for (int i = 0; i < width; ++i)
				hdr_convert(&hdr[((size_t)j * width + i) * comp], &scanline[(size_t)i * 4], comp);
		}
	}

	// stbi__hdr_to_ldr with the default gamma 2.2 and scale 1
	const float gamma_i = 1.0f / 2.2f, scale_i = 1.0f;
	std::vector<uint8_t> px((size_t)width * (size_t)height * (size_t)comp);
	const int n = (comp & 1) ? comp : comp - 1; // colour channels; an alpha channel is linear
	for (size_t i = 0; i < (size_t)width * (size_t)height; ++i) {
		int k = 0;
		for (; k < n; ++k) {
			float z = (float)std::pow((double)(hdr[i * comp + k] * scale_i), (double)gamma_i) * 255 + 0.5f; // double pow, as in C
			if (z < 0) z = 0;
			if (z > 255) z = 255;
			px[i * comp + k] = (uint8_t)(int)z;
		}
		if (k < comp) {
			float z = hdr[i * comp + k] * 255 + 0.5f;
			if (z < 0) z = 0;
			if (z > 255) z = 255;
			px[i * comp + k] = (uint8_t)(int)z;
		}
	}
	out->w = width;
	out->h = height;
	out->comp_in_file = 3;
	out->comp = comp;
	out->px = std::move(px);
	return true;
}

} // namespace hmrm
