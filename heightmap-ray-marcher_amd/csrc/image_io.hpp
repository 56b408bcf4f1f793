// image_io.hpp -- native image decode/encode for the data formats either side
// of the hot path (L1 data prep in, L0 PNG/PPM out).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace hmrm {

struct Image {
	int32_t w = 0, h = 0;
	int32_t comp_in_file = 0; // channels the file itself holds (stb's *n)
	int32_t comp = 0;         // channels in `px`
	std::vector<uint8_t> px;  // row-major, top-left origin, 8 bits per channel
};

// Decode PNG, JPEG, BMP, GIF, PSD, PIC, Radiance HDR, TGA or binary PNM from memory into 8-bit channels, converted to
// req_comp (0 = keep) the way stb_image v2.27's stbi_load does.  On failure
// returns false and sets err to a short reason.
bool decode_image(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err);
bool load_image_file(const char *path, int req_comp, Image *out, std::string *err);
// Baseline / progressive JPEG (jpeg_decode.cpp), same conventions.
bool decode_jpeg(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err);
// Windows BMP and Truevision TGA (bmp_tga_decode.cpp), same conventions.
bool looks_like_bmp(const uint8_t *bytes, size_t len);
bool decode_bmp(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err);
bool looks_like_tga(const uint8_t *bytes, size_t len);
bool decode_tga(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err);
// GIF (first image), Photoshop PSD (composite, RGB), Softimage PIC, Radiance HDR tone-mapped to 8 bits
// (legacy_formats.cpp), same conventions.
bool looks_like_gif(const uint8_t *bytes, size_t len);
bool decode_gif(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err);
bool looks_like_psd(const uint8_t *bytes, size_t len);
bool decode_psd(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err);
bool looks_like_pic(const uint8_t *bytes, size_t len);
bool decode_pic(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err);
bool looks_like_hdr(const uint8_t *bytes, size_t len);
bool decode_hdr(const uint8_t *bytes, size_t len, int req_comp, Image *out, std::string *err);
// stb's channel conversion table for 8-bit data (grey<->RGB, alpha add/drop, luma).
std::vector<uint8_t> convert_channels8(const std::vector<uint8_t> &src, int from, int to, size_t npix);

// PNG encoder producing the same bytes as stb_image_write v1.16's
// stbi_write_png_to_mem at its defaults (compression level 8, filter chosen per
// row by minimum sum of absolute values).
bool encode_png(int32_t w, int32_t h, int32_t comp, const uint8_t *data, size_t stride_bytes,
                std::vector<uint8_t> *out);
bool write_file(const char *path, const uint8_t *data, size_t len);
// P6, drops alpha when comp == 4; comp 1 writes P5.
bool encode_pnm(int32_t w, int32_t h, int32_t comp, const uint8_t *data, size_t stride_bytes,
                std::vector<uint8_t> *out);

// zlib stream helpers (exposed for tests).
bool zlib_inflate(const uint8_t *src, size_t len, std::vector<uint8_t> *out, std::string *err);
void zlib_deflate_stb(const uint8_t *data, int data_len, int quality, std::vector<uint8_t> *out);

} // namespace hmrm
