// device_common.hpp -- device helpers shared by the render kernels: ray
// generation (src/{Perspective,Spherical,Orthographic}.cpp GetRay), the slab test
// (src/AABB.cpp:49-77), pixel packing (main/hmap.cpp:139-154) and the sky shade
// (main/hmap.cpp:1041-1057).  Include only from .hip files compiled with
// -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "frame.hpp"

#pragma clang fp contract(off)

namespace hmrm {

// ----------------------------------------------------------------- render ----
struct DevRay {
	double px, py, pz;
	double dx, dy, dz;
};

template <int PROJ>
__device__ __forceinline__ DevRay make_ray(const DevFrame &f, int px, int py) {
	DevRay r;
	if (PROJ == 2) {
		// Spherical.cpp:23-25; sin/cos come from the host tables
		const double sva = f.row_sin_va[py], cva = f.row_cos_va[py];
		const double cha = f.col_cos_ha[px], sha = f.col_sin_ha[px];
		r.px = f.cam[0]; r.py = f.cam[1]; r.pz = f.cam[2];
		r.dx = sva * cha;
		r.dy = sva * sha;
		r.dz = cva;
	} else {
		// hmap.cpp:985-988
		const double w = (double)px / (double)(f.screen_w - 1);
		const double h = (double)py / (double)(f.screen_h - 1);
		// upper_left + w*plane_right + h*plane_down  (Perspective.cpp:27, Orthographic.cpp:20)
		const double ox = (f.upper_left[0] + w * f.plane_right[0]) + h * f.plane_down[0];
		const double oy = (f.upper_left[1] + w * f.plane_right[1]) + h * f.plane_down[1];
		const double oz = (f.upper_left[2] + w * f.plane_right[2]) + h * f.plane_down[2];
		if (PROJ == 1) {
			const double vx = ox - f.cam[0], vy = oy - f.cam[1], vz = oz - f.cam[2];
			// glm::normalize: v * (1 / sqrt(dot(v,v))), dot = (x*x + y*y) + z*z
			const double tx = vx * vx, ty = vy * vy, tz = vz * vz;
			const double inv = 1.0 / __builtin_sqrt((tx + ty) + tz);
			r.px = f.cam[0]; r.py = f.cam[1]; r.pz = f.cam[2];
			r.dx = vx * inv;
			r.dy = vy * inv;
			r.dz = vz * inv;
		} else {
			r.px = ox; r.py = oy; r.pz = oz;
			r.dx = f.look[0]; r.dy = f.look[1]; r.dz = f.look[2];
		}
	}
	return r;
}

// AABB.cpp:49-77, axis order x,y,z, same comparisons (NaN => every test false).
__device__ __forceinline__ double slab_distance(const DevRay &r, const DevFrame &f) {
	const double inf = __builtin_huge_val();
	double lo = -inf, hi = inf;
	const double ro[3] = {r.px, r.py, r.pz};
	const double rd[3] = {r.dx, r.dy, r.dz};
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		double dim_lo = (f.c0[i] - ro[i]) / rd[i];
		double dim_hi = (f.c1[i] - ro[i]) / rd[i];
		if (dim_lo > dim_hi) {
			const double t = dim_lo;
			dim_lo = dim_hi;
			dim_hi = t;
		}
		if (dim_hi < lo || dim_lo > hi) return inf;
		if (dim_lo > lo) lo = dim_lo;
		if (dim_hi < hi) hi = dim_hi;
	}
	return (lo > hi) ? inf : lo;
}

// Relative error of the hardware reciprocal.  MEASURED on gfx950 (tools/rcp_accuracy.py over 1.6e11 inputs: the
// leading 32 mantissa bits exhaustively in six binades, hashed mantissas / signs / exponents 2^-1000..2^1000, and the
// product n * rcp(d) against n / d; profiles/r03_rcp_accuracy.txt): |rcp(x) * x - 1| <= 2^-24.36 everywhere, the same
// in every binade -- a 24-bit result, as AMD's ISA guides say ("(2**29) ULP").  tests/test_parity_gpu.py
// re-measures a slice in every GPU run and fails if the device at hand exceeds kRcpRelErr.
constexpr double kRcpRelErr = 0x1p-24;
// ... and the margin every ordering decision below keeps between two approximate slab parameters, as a fraction of
// the largest of them: 16 x kRcpRelErr.  (Needed: 2 x (kRcpRelErr + 2^-53), one error bar per operand.)
constexpr double kSlabMargin = 16.0 * kRcpRelErr; // 2^-20

// Cheap classification of a ray against the box from approximate reciprocals (one v_rcp_f64 and two multiplies per
// axis).  With every direction component and every product of moderate magnitude (2^-500..2^500: no overflow, no
// underflow, no zero) an approximate slab parameter is t' = t (1 + e), |e| <= kRcpRelErr + 2^-53, of the exact real
// quotient t: same sign, and two of them that differ by more than kSlabMargin * max|t'| are ordered like the exact
// quotients -- hence, rounding being monotonic, their correctly rounded values (the reference's) are ordered the same
// way or equal.  So a MISS is proven when
//   (A) min_i hi'_i < 0          => the exact hi is negative => d <= hi < 0 or d = inf;
//   (B) max_i lo'_i exceeds min_i hi'_i by more than the error bars (2^-12 of the largest: far more) => the exact
//       intervals do not overlap => distance() returns inf (early or at its last line).
// Either way the pixel is a miss (AABB.cpp:33-40); nothing else about d is used for a miss.  "Don't know" (take the
// exact path) whenever a direction component or a parameter is zero, tiny, huge or not finite.
// The same approximate parameters also settle most HITS with one division instead of six.  For finite
// quotients distance() returns lo = max_i min(t0_i, t1_i) when lo <= hi = min_i max(t0_i, t1_i), else inf
// (its early returns are that comparison on a prefix of the axes).  If the approximate values show, by more than
// the margin, (a) that the intervals overlap, (b) which axis holds the maximum of the lower ends and (c) which of its
// two quotients is the lower end, then the exact result is that ONE quotient, correctly rounded (where two rounded
// lower ends coincide the value is the same whichever axis supplied it).  Rays through an edge or a corner of the
// box (lower ends within 2^-20 of each other) stay undecided and take the reference's six divisions.
// Returns 0: undecided, take slab_distance(); 1: surely a miss (only when `may_report_miss`; d is not set);
// 2: *d is distance()'s value, bit for bit (it may be negative: the caller's d < 0 test applies as usual).
__device__ __forceinline__ int slab_classify(const DevRay &r, const DevFrame &f, bool may_report_miss, double *d) {
	const double ro[3] = {r.px, r.py, r.pz};
	const double rd[3] = {r.dx, r.dy, r.dz};
	const double inf = __builtin_huge_val();
	double LO = -inf, LO2 = -inf, HI = inf, mag = 0.0, least = inf;
	double num = 0.0, den = 1.0, span = 0.0; // of the axis that holds LO: entry numerator, direction, |t0' - t1'|
	bool fine = true;
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		fine = fine & (__builtin_fabs(rd[i]) > 0x1p-500) & (__builtin_fabs(rd[i]) < 0x1p500);
		const double inv = __builtin_amdgcn_rcp(rd[i]);
		const double n0 = f.c0[i] - ro[i], n1 = f.c1[i] - ro[i];
		const double t0 = n0 * inv, t1 = n1 * inv;
		const bool first = t0 < t1;
		const double lo_i = first ? t0 : t1, hi_i = first ? t1 : t0;
		const bool bigger = lo_i > LO;
		LO2 = bigger ? LO : __builtin_fmax(LO2, lo_i);
		num = bigger ? (first ? n0 : n1) : num;
		den = bigger ? rd[i] : den;
		span = bigger ? hi_i - lo_i : span;
		LO = bigger ? lo_i : LO;
		HI = __builtin_fmin(HI, hi_i);
		const double a0 = __builtin_fabs(t0), a1 = __builtin_fabs(t1);
		mag = __builtin_fmax(mag, __builtin_fmax(a0, a1));
		least = __builtin_fmin(least, __builtin_fmin(a0, a1)); // (NaN operands drop out here and are caught by mag / fine below)
		fine = fine & !__builtin_isunordered(t0, t1);
	}
	// every parameter finite and of moderate magnitude: the relative-error model above holds for each of them
	if (!(fine & (mag < 0x1p500) & (least > 0x1p-500))) return 0;
	if (may_report_miss && (HI < 0.0 || (LO - HI) > mag * 0x1p-12)) return 1;
	const double margin = mag * kSlabMargin;
	if ((HI - LO) > margin && (LO - LO2) > margin && span > margin) {
		*d = num / den;
		return 2;
	}
	return 0;
}

// Cheaper still, from sign and exponent bits alone: on some axis the whole box lies on one
// side of the origin (c0-o and c1-o have the same sign) and the ray points the other way.
// With all three numbers finite, non-zero and of moderate exponent (2^-500 .. 2^499) both
// exact quotients are finite, non-zero and negative, so that axis' dim_hi < 0 and
// distance() (AABB.cpp:49-77) ends in inf or in an entry distance <= hi < 0: a miss.
// PROJ 1 / 2: the origin is the camera for every ray, so the two box-side tests are done once per frame on the
// host (camera.cpp, DevFrame::box_side) and a ray only adds the sign and the exponent of its direction.
template <int PROJ>
__device__ __forceinline__ bool slab_points_away(const DevRay &r, const DevFrame &f) {
	const double ro[3] = {r.px, r.py, r.pz};
	const double rd[3] = {r.dx, r.dy, r.dz};
	bool away = false;
	if (PROJ != 3) {
#pragma unroll
		for (int i = 0; i < 3; ++i) {
			if (!f.box_side_known[i]) continue; // (uniform)
			const uint32_t hd = (uint32_t)((unsigned long long)__double_as_longlong(rd[i]) >> 32);
			away = away | ((((hd >> 20) & 0x7ffu) - 523u) < 1000u & ((f.box_side[i] ^ hd) >> 31) != 0u);
		}
		return away;
	}
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const uint32_t h0 = (uint32_t)((unsigned long long)__double_as_longlong(f.c0[i] - ro[i]) >> 32);
		const uint32_t h1 = (uint32_t)((unsigned long long)__double_as_longlong(f.c1[i] - ro[i]) >> 32);
		const uint32_t hd = (uint32_t)((unsigned long long)__double_as_longlong(rd[i]) >> 32);
		const bool moderate = (((h0 >> 20) & 0x7ffu) - 523u) < 1000u && (((h1 >> 20) & 0x7ffu) - 523u) < 1000u &&
		                      (((hd >> 20) & 0x7ffu) - 523u) < 1000u;
		away = away || (moderate && ((h0 ^ h1) >> 31) == 0u && ((h0 ^ hd) >> 31) != 0u);
	}
	return away;
}

__device__ __forceinline__ uint32_t pack_rgba(uint32_t r, uint32_t g, uint32_t b) {
	return r | (g << 8) | (b << 16) | 0xff000000u; // bytes R,G,B,A=255 (hmap.cpp:150-153)
}

// Clamp<double>(v,0,255) then floor then (Uint8), hmap.cpp:1049-1051, as written there (the literal kernel's)
__device__ __forceinline__ uint32_t sky_channel_literal(double v) {
	if (v < 0.0) v = 0.0;
	else if (v > 255.0) v = 255.0;
	return (uint32_t)(int)__builtin_floor(v);
}
// The same for the values shade_miss produces: v is
// a sum of non-negative terms (a weight times z or z*z with z > 0, plus a colour byte), never NaN and never
// negative, possibly +inf.  So the lower clamp never acts, min(v, 255) is the upper one, and the truncating
// cast equals the floor.
__device__ __forceinline__ uint32_t sky_channel(double v) {
	return (uint32_t)__builtin_fmin(v, 255.0);
}

struct StatsOut {
	unsigned long long *counters; // [0] steps [1] hits [2] capped
	uint32_t *steps_per_pixel;    // screen_w*screen_h or null
	double *entry_d;              // screen_w*screen_h or null
};


// Miss shade: hmap.cpp:1041-1057.
__device__ __forceinline__ uint32_t shade_miss(const DevFrame &f, double dz) {
	if (dz > 0.0) {
		const double zz = dz * dz; // std::pow(z,2) == z*z under -std=c++98 (__builtin_powi)
		const double r_ = 220.0 * zz + (double)f.bg[0];
		const double g_ = 240.0 * zz + (double)f.bg[1];
		const double b_ = 255.0 * dz + (double)f.bg[2];
		return pack_rgba(sky_channel(r_), sky_channel(g_), sky_channel(b_));
	}
	return pack_rgba(f.bg[0], f.bg[1], f.bg[2]);
}

// Hit shade: hmap.cpp:1018-1031 (alpha 0 -> background colour).
__device__ __forceinline__ uint32_t shade_hit(const DevFrame &f, uint32_t texel) {
	return ((texel >> 24) == 0) ? pack_rgba(f.bg[0], f.bg[1], f.bg[2]) : (texel | 0xff000000u);
}

// Which pixel a lane owns.  A wave covers kWaveW x kWaveH pixels (kWaveW * kWaveH = 64), a
// workgroup a kWavesX x kWavesY arrangement of waves (default 2 x 2 = 256 threads).
#ifndef HMRM_WAVE_W
#define HMRM_WAVE_W 8
#endif
#ifndef HMRM_WAVES_X
#define HMRM_WAVES_X 1
#endif
#ifndef HMRM_WAVES_Y
#define HMRM_WAVES_Y 2
#endif
constexpr int kWaveW = HMRM_WAVE_W, kWaveH = 64 / HMRM_WAVE_W;
constexpr int kWavesX = HMRM_WAVES_X, kWavesY = HMRM_WAVES_Y;          // waves per workgroup
constexpr int kBlockThreads = 64 * kWavesX * kWavesY;
constexpr int kTileW = kWavesX * kWaveW, kTileH = kWavesY * kWaveH;
struct PixelId {
	int px, py, lrow;
	int tile_y; // tile row of the launch this workgroup renders (wave-uniform)
	bool live;
};
// Frame row of local row `lrow` of the launch's output (contiguous strip or cyclic bands).
__device__ __forceinline__ int frame_row_of(const RowMap &rows, int lrow) {
	if (rows.band_rows > 0) {
		const int band = lrow / rows.band_rows, within = lrow - band * rows.band_rows;
		return (rows.band_index + band * rows.band_count) * rows.band_rows + within;
	}
	return rows.row_begin + lrow;
}

// Pixel of lane `lane` of wave `wave` of the workgroup-sized tile (tile_x, grid row gy).
__device__ __forceinline__ PixelId pixel_of_tile_lane(const DevFrame &f, const RowMap &rows, int tiles_y, int tile_x, unsigned gy,
                                                      int wave, int lane) {
	// Grid of tiles: x = tile column (fastest, so workgroups still start row by row), y (+ z for
	// frames taller than 32768 tile rows) = grid row; grid rows are handed to tile rows piece by piece so that the
	// costliest tile rows start first (RowMap::seg_first / seg_delta).  Scalar work, no integer division per wave.
	const bool row_exists = gy < (unsigned)tiles_y; // (the last z-slab may be partly empty)
	int delta = rows.seg_delta[0];
#pragma unroll
	for (int k = 1; k < kOrderSegs; ++k) delta = gy >= (unsigned)rows.seg_first[k - 1] ? rows.seg_delta[k] : delta;
	unsigned ty = gy + (unsigned)delta;
	if (ty >= (unsigned)tiles_y) ty -= (unsigned)tiles_y;
	const int tile_y = (int)ty;
	PixelId p;
	p.tile_y = row_exists ? tile_y : -1;
	p.px = tile_x * kTileW + (wave % kWavesX) * kWaveW + (lane % kWaveW);
	p.lrow = tile_y * kTileH + (wave / kWavesX) * kWaveH + (lane / kWaveW);
	p.py = frame_row_of(rows, p.lrow);
	p.live = row_exists && p.px < f.screen_w && p.lrow < rows.local_rows && p.py < f.screen_h;
	return p;
}
// ... of the launch's own workgroup (one workgroup per tile: blockIdx is the tile)
__device__ __forceinline__ PixelId pixel_of_lane(const DevFrame &f, const RowMap &rows, int tiles_y) {
	return pixel_of_tile_lane(f, rows, tiles_y, (int)blockIdx.x, blockIdx.z * 32768u + blockIdx.y, (int)(threadIdx.x >> 6),
	                          (int)(threadIdx.x & 63));
}

// Wave-reduce and publish the per-launch counters {steps, hits, capped}.
template <bool STATS>
__device__ __forceinline__ void publish_counters(const StatsOut &st, unsigned long long steps,
                                                 uint32_t hit, uint32_t cap) {
	if (STATS) {
		unsigned long long s = steps, h = hit, c = cap;
		for (int off = 32; off > 0; off >>= 1) {
			s += __shfl_xor(s, off);
			h += __shfl_xor(h, off);
			c += __shfl_xor(c, off);
		}
		if ((threadIdx.x & 63) == 0) {
			if (s) atomicAdd(&st.counters[0], s);
			if (h) atomicAdd(&st.counters[1], h);
			if (c) atomicAdd(&st.counters[2], c);
		}
	} else if (cap) {
		atomicAdd(&st.counters[2], 1ull);
	}
}

} // namespace hmrm
