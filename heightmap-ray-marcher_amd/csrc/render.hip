// render.hip -- gfx950 kernels for the reference's hot path and their launchers.
//
//   k_prepare_heights  UpdateHeightmap            main/hmap.cpp:171-191
//   k_render           the pixel loop             main/hmap.cpp:978-1058
//                        ImagePlane::GetRay       src/{Perspective,Spherical,Orthographic}.cpp
//                        intersection/distance    src/AABB.cpp:30-77
//                        SetPixel                 main/hmap.cpp:139-154
//
// Numerical contract: every fp64 operation of the reference is performed once, in
// the reference's order, with IEEE round-to-nearest and NO fused multiply-add
// (this file must be compiled with -ffp-contract=off; hipcc contracts by
// default).  Division and sqrt are the correctly rounded LLVM expansions.  No
// transcendental is evaluated on the device (the host supplies them, frame.hpp).
//
// Bit-preserving departures from the literal loop, each argued where it is made:
//   * heightmap_buf[i] + hmap_c0.z (hmap.cpp:1016) is loop invariant per cell and
//     precomputed into the `thr` table by k_prepare_heights;
//   * step_dist * ray.dir (hmap.cpp:1037) is loop invariant per ray and hoisted;
//   * x / grid_width is evaluated as x * (1/grid_width) only when grid_width is a
//     power of two (exactly representable reciprocal => identical rounding);
//   * the (int) casts + integer range test (hmap.cpp:1001-1011) are evaluated as
//     an fp range test followed by the cast (same predicate, no UB, NaN breaks as
//     on x86 where cvttsd2si gives INT_MIN);
//   * the unbounded while(true) gets a step cap that is reported, never silent.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "device_common.hpp"
#include "render.hpp"

#pragma clang fp contract(off)

namespace hmrm {

// --------------------------------------------------------------- heights ----
// thr[i] = heightmap_buf[i] + c0.z  with heightmap_buf[i] exactly as
// UpdateHeightmap computes it.  PLAIN = true writes heightmap_buf[i] itself
// (test hook).  Also reduces max(thr) with an order-preserving integer key.
__device__ __forceinline__ unsigned long long f64_order_key(double v) {
	unsigned long long b = (unsigned long long)__double_as_longlong(v);
	return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

template <bool PLAIN>
__global__ __launch_bounds__(256) void k_prepare_heights(const uint8_t *__restrict__ rgb,
                                                         double *__restrict__ out, int64_t n,
                                                         double lum_r, double lum_g, double lum_b,
                                                         double min_h, double max_h,
                                                         unsigned long long *__restrict__ max_key) {
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	unsigned long long local = 0; // smaller than the key of every double
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const double r = (double)rgb[3 * i + 0];
		const double g = (double)rgb[3 * i + 1];
		const double b = (double)rgb[3 * i + 2];
		double value = ((lum_r * r) + (lum_g * g)) + (lum_b * b);
		// Clamp<double>, hmap.cpp:118-124 (NaN falls through both tests)
		if (value < 0.0) value = 0.0;
		else if (value > 255.0) value = 255.0;
		const double hz = (value / 255.0) * (max_h - min_h) + min_h;
		const double v = PLAIN ? hz : hz + min_h; // hmap.cpp:1016: heightmap_z + hmap_c0.z
		out[i] = v;
		if (!PLAIN) {
			const unsigned long long k = f64_order_key(v);
			if (v == v && k > local) local = k;
		}
	}
	if (!PLAIN) {
		for (int off = 32; off > 0; off >>= 1) {
			unsigned long long o = __shfl_xor(local, off);
			if (o > local) local = o;
		}
		if ((threadIdx.x & 63) == 0 && local) atomicMax(max_key, local);
	}
}

template <int PROJ, bool STATS>
__global__ __launch_bounds__(kBlockThreads) void k_render(const DevFrame f, const RowMap rows,
                                                const double *__restrict__ thr,
                                                const uint32_t *__restrict__ cmap,
                                                uint32_t *__restrict__ out, int64_t out_stride_px,
                                                int tiles_y, StatsOut st) {
	// one small pixel tile per wave (device_common.hpp): neighbouring rays walk neighbouring
	// ground tracks, so a wave's height loads share cache lines and its lanes leave the loop
	// at similar times.
	const PixelId pid = pixel_of_lane(f, rows, tiles_y);
	const int px = pid.px, py = pid.py, lrow = pid.lrow;
	const bool live = pid.live;

	unsigned long long my_steps = 0;
	uint32_t my_hit = 0, my_cap = 0;

	if (live) {
		const DevRay ray = make_ray<PROJ>(f, px, py);
		const double d = slab_distance(ray, f);
		if (STATS && st.entry_d) st.entry_d[(int64_t)py * f.screen_w + px] = d;

		uint32_t rgba = 0;
		bool real_hit = false;

		// intersection(): AABB.cpp:33-44
		if (!(d == __builtin_huge_val()) && !(d < 0.0)) {
			double x = ray.px + d * ray.dx;
			double y = ray.py + d * ray.dy;
			double z = ray.pz + d * ray.dz;
			// hmap.cpp:998  int_point += (grid_width*0.01) * dir
			x = x + f.nudge * ray.dx;
			y = y + f.nudge * ray.dy;
			z = z + f.nudge * ray.dz;
			// hmap.cpp:1037  step_dist * dir is the same three products every iteration
			const double sx = f.step_dist * ray.dx;
			const double sy = f.step_dist * ray.dy;
			const double sz = f.step_dist * ray.dz;
			const double wlim = (double)f.map_w, hlim = (double)f.map_h;
			const double c0x = f.c0[0], c0y = f.c0[1];
			const bool pow2 = f.grid_pow2 != 0;
			int64_t budget = f.step_cap;

			for (;;) {
				// hmap.cpp:1001-1004
				double qx, qy;
				if (pow2) {
					qx = (x - c0x) * f.inv_grid_width;
					qy = -(y - c0y) * f.inv_grid_width;
				} else {
					qx = (x - c0x) / f.grid_width;
					qy = -(y - c0y) / f.grid_width;
				}
				// hmap.cpp:1006-1011: (int)q >= 0  <=>  q > -1 ;  (int)q < W  <=>  q < W
				if (!(qx > -1.0 && qx < wlim && qy > -1.0 && qy < hlim)) break;
				if (budget-- <= 0) { my_cap = 1; break; }
				const int gridx = (int)qx, gridy = (int)qy;
				const int64_t cell = (int64_t)gridy * f.map_w + gridx;
				const double t = thr[cell]; // hmap.cpp:1013-1014 (+ c0.z folded in)
				if (STATS) my_steps += 1;
				if (z < t) { // hmap.cpp:1016
					const uint32_t c = cmap[cell];
					rgba = ((c >> 24) == 0) ? pack_rgba(f.bg[0], f.bg[1], f.bg[2]) : (c | 0xff000000u);
					real_hit = true;
					break;
				}
				x = x + sx;
				y = y + sy;
				z = z + sz;
			}
		}

		if (!real_hit) {
			// hmap.cpp:1041-1057
			if (ray.dz > 0.0) {
				const double zz = ray.dz * ray.dz; // std::pow(z,2) == z*z under -std=c++98
				const double r_ = 220.0 * zz + (double)f.bg[0];
				const double g_ = 240.0 * zz + (double)f.bg[1];
				const double b_ = 255.0 * ray.dz + (double)f.bg[2];
				rgba = pack_rgba(sky_channel_literal(r_), sky_channel_literal(g_), sky_channel_literal(b_));
			} else {
				rgba = pack_rgba(f.bg[0], f.bg[1], f.bg[2]);
			}
		} else {
			my_hit = 1;
		}
		out[(int64_t)lrow * out_stride_px + px] = rgba;
		if (STATS && st.steps_per_pixel)
			st.steps_per_pixel[(int64_t)py * f.screen_w + px] =
			    my_steps > 0xffffffffull ? 0xffffffffu : (uint32_t)my_steps;
	}

	publish_counters<STATS>(st, my_steps, my_hit, my_cap);
}

// Per-ray parity hook: GetRay + distance() of one pixel -> out[0..2] pos, [3..5] dir, [6] d.
template <int PROJ>
__global__ void k_probe(const DevFrame f, int px, int py, double *__restrict__ out) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const DevRay r = make_ray<PROJ>(f, px, py);
	out[0] = r.px; out[1] = r.py; out[2] = r.pz;
	out[3] = r.dx; out[4] = r.dy; out[5] = r.dz;
	out[6] = slab_distance(r, f);
}

// Accuracy probe for v_rcp_f64 (test hook; device_common.hpp slab_classify rests on the bound it reports).
// For every sample x: r = rcp(x) and e = |fma(r, x, -1)|, which is |r - 1/x| / |1/x| up to a factor 1 + 2^-53
// (r*x - 1 is formed exactly inside the fma and rounded once).  mode 0 walks the leading 32 mantissa bits
// exhaustively (sample i has them = i; the 20 trailing bits are 0, all ones, or hashed, by `seed & 3`; exponent
// exp_lo, positive); mode 1 hashes mantissa, sign and an exponent in [exp_lo, exp_hi]; mode 2 measures what the
// shortcut really forms, t' = n * rcp(d) against the correctly rounded n / d (|t' - q| / |q|: the difference of two
// neighbours is exact), with hashed n of exponent in [exp_lo, exp_hi] and direction-like d (exponent -40..0).
// out[0] = max e as fp64 bits (positive doubles order like integers), hist[k] = samples with e in [2^-k, 2^-(k-1)).
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
	z += 0x9e3779b97f4a7c15ull;
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_rcp_error(int mode, uint64_t count, uint64_t seed, int exp_lo, int exp_hi,
                                                   unsigned long long *__restrict__ out_max,
                                                   unsigned long long *__restrict__ hist) {
	__shared__ unsigned int lh[64];
	if (threadIdx.x < 64) lh[threadIdx.x] = 0u;
	__syncthreads();
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	double worst = 0.0;
	const unsigned span = (unsigned)(exp_hi - exp_lo + 1);
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
		const uint64_t h = mix64(i ^ (seed * 0x9e3779b97f4a7c15ull));
		double e;
		if (mode == 2) {
			const uint64_t h2 = mix64(h);
			const uint64_t nb = (h & 0x800fffffffffffffull) | ((uint64_t)(1023 + exp_lo + (int)((h >> 52) % span)) << 52);
			const uint64_t db = (h2 & 0x800fffffffffffffull) | ((uint64_t)(1023 - (int)((h2 >> 52) % 41u)) << 52);
			const double n = __longlong_as_double((long long)nb), d = __longlong_as_double((long long)db);
			const double t = n * __builtin_amdgcn_rcp(d), q = n / d;
			e = __builtin_fabs(t - q) / __builtin_fabs(q);
		} else {
			uint64_t b;
			if (mode == 0) {
				const uint64_t low = (seed & 3) == 0 ? 0ull : ((seed & 3) == 1 ? 0xfffffull : (h & 0xfffffull));
				b = ((uint64_t)(1023 + exp_lo) << 52) | ((i & 0xffffffffull) << 20) | low;
			} else {
				b = (h & 0x800fffffffffffffull) | ((uint64_t)(1023 + exp_lo + (int)((h >> 52) % span)) << 52);
			}
			const double x = __longlong_as_double((long long)b);
			const double r = __builtin_amdgcn_rcp(x);
			e = __builtin_fabs(__builtin_fma(r, x, -1.0));
		}
		worst = __builtin_fmax(worst, e);
		// bin = -exponent of e, clamped to 0..63 (e == 0 -> 63)
		int k = 1023 - (int)((((unsigned long long)__double_as_longlong(e)) >> 52) & 0x7ffu);
		k = e == 0.0 ? 63 : (k < 0 ? 0 : (k > 63 ? 63 : k));
		atomicAdd(&lh[k], 1u);
	}
	for (int off = 32; off > 0; off >>= 1) worst = __builtin_fmax(worst, __shfl_xor(worst, off));
	if ((threadIdx.x & 63) == 0) atomicMax(out_max, (unsigned long long)__double_as_longlong(worst));
	__syncthreads();
	if (threadIdx.x < 64 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

hipError_t launch_rcp_error(int mode, uint64_t count, uint64_t seed, int exp_lo, int exp_hi,
                            unsigned long long *d_out65, hipStream_t stream) {
	// (a block's LDS histogram counts in 32 bits: at most 2^32 / 4096 blocks... keep every block below 2^31 samples)
	hipLaunchKernelGGL(k_rcp_error, dim3(256 * 16), dim3(256), 0, stream, mode, count, seed, exp_lo, exp_hi, d_out65,
	                   d_out65 + 1);
	return hipGetLastError();
}

// Spherical tables, pinned host staging -> device, by a kernel of the launch stream instead of a copy command: a
// DMA copy between two kernels of one stream costs two cross-engine hand-overs (~25 us per frame measured on a
// moving camera); this is one more small dispatch on the same queue, reading 96 KB (4K frame) over PCIe.
__global__ __launch_bounds__(256) void k_upload_tables(const double *__restrict__ host_src, double *__restrict__ dst, int n) {
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i < n) dst[i] = host_src[i];
}

hipError_t launch_upload_tables(const double *h_pinned, double *d_dst, size_t n, hipStream_t stream) {
	if (n == 0 || n > 0x7fffffffu) return n ? hipErrorInvalidValue : hipSuccess;
	hipLaunchKernelGGL(k_upload_tables, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, h_pinned, d_dst, (int)n);
	return hipGetLastError();
}

// ------------------------------------------------------------- launchers ----
hipError_t launch_probe(const DevFrame &f, int px, int py, double *d_out7, hipStream_t stream) {
	switch (f.projection) {
	case 1: hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(64), 0, stream, f, px, py, d_out7); break;
	case 2: hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(64), 0, stream, f, px, py, d_out7); break;
	default: hipLaunchKernelGGL(k_probe<3>, dim3(1), dim3(64), 0, stream, f, px, py, d_out7); break;
	}
	return hipGetLastError();
}

hipError_t launch_prepare_heights(const uint8_t *d_rgb, double *d_out, int64_t n, double lum_r,
                                  double lum_g, double lum_b, double min_h, double max_h, bool plain,
                                  unsigned long long *d_max_key, hipStream_t stream) {
	int64_t blocks = (n + 255) / 256;
	if (blocks > 256 * 8) blocks = 256 * 8; // grid-stride the rest
	if (blocks < 1) blocks = 1;
	if (plain)
		hipLaunchKernelGGL(k_prepare_heights<true>, dim3((unsigned)blocks), dim3(256), 0, stream, d_rgb,
		                   d_out, n, lum_r, lum_g, lum_b, min_h, max_h, d_max_key);
	else
		hipLaunchKernelGGL(k_prepare_heights<false>, dim3((unsigned)blocks), dim3(256), 0, stream, d_rgb,
		                   d_out, n, lum_r, lum_g, lum_b, min_h, max_h, d_max_key);
	return hipGetLastError();
}

double max_key_to_double(unsigned long long key) {
	unsigned long long b = (key & 0x8000000000000000ull) ? (key & 0x7fffffffffffffffull) : ~key;
	double v;
	__builtin_memcpy(&v, &b, sizeof v);
	return v;
}

template <bool STATS>
static hipError_t launch_render_t(const DevFrame &f, const RowMap &rows, const double *d_thr,
                                  const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px,
                                  StatsOut st, hipStream_t stream) {
	const int tiles_x = (f.screen_w + kTileW - 1) / kTileW;
	const int tiles_y = (rows.local_rows + kTileH - 1) / kTileH;
	if (tiles_x <= 0 || tiles_y <= 0) return hipSuccess;
	const dim3 grid((unsigned)tiles_x, (unsigned)(tiles_y < 32768 ? tiles_y : 32768), (unsigned)((tiles_y + 32767) / 32768)), block(kBlockThreads);
	switch (f.projection) {
	case 1:
		hipLaunchKernelGGL((k_render<1, STATS>), grid, block, 0, stream, f, rows, d_thr, d_cmap, d_out,
		                   out_stride_px, tiles_y, st);
		break;
	case 2:
		hipLaunchKernelGGL((k_render<2, STATS>), grid, block, 0, stream, f, rows, d_thr, d_cmap, d_out,
		                   out_stride_px, tiles_y, st);
		break;
	default:
		hipLaunchKernelGGL((k_render<3, STATS>), grid, block, 0, stream, f, rows, d_thr, d_cmap, d_out,
		                   out_stride_px, tiles_y, st);
		break;
	}
	return hipGetLastError();
}

hipError_t launch_render(const DevFrame &f, const RowMap &rows, const double *d_thr,
                         const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px,
                         unsigned long long *d_counters, uint32_t *d_steps, double *d_entry,
                         bool stats, hipStream_t stream) {
	StatsOut st{d_counters, d_steps, d_entry};
	return stats ? launch_render_t<true>(f, rows, d_thr, d_cmap, d_out, out_stride_px, st, stream)
	             : launch_render_t<false>(f, rows, d_thr, d_cmap, d_out, out_stride_px, st, stream);
}

} // namespace hmrm
