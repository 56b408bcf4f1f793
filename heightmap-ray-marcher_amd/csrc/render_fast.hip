// render_fast.hip -- the production march kernel for gfx950.
//
// Same pixels, same per-ray step counts as k_render (render.hip) and therefore as
// the reference loop main/hmap.cpp:978-1058; two bit-preserving restructurings:
//
// (1) SPECULATIVE GROUPS.  The positions a ray visits do not depend on the heights
//     it loads -- only the decision to stop does.  So U consecutive positions are
//     produced with the reference's sequential adds (hmap.cpp:1037), their U height
//     loads are issued together, and the tests (range :1006, hit :1016) are then
//     resolved in order.  One memory latency per U steps instead of per step.
//
// (2) EXACT LEAPS.  While a coordinate p stays inside one binade [2^E, 2^(E+1)) every
//     value is a multiple of u = 2^(E-52), and fl(p + s) = p + delta with the SAME
//     delta = round_u(s) for every p of that binade (round-to-nearest; exact ties,
//     where the result depends on the parity of p, are excluded).  Hence the
//     reference's sequential accumulation satisfies p_k = p_0 + k*delta EXACTLY, and
//     both the product k*delta and the sum are exact in fp64.  A max pyramid over the
//     hit thresholds then lets a ray jump over n steps at once when all n skipped
//     positions provably (a) stay in the block whose maximum was looked up and
//     (b) stay at or above that maximum (no hit possible: hmap.cpp:1016 needs
//     z < threshold), with (c) all three coordinates inside their binades.  The jump
//     length is only ESTIMATED (approximate reciprocals); the landing point is then
//     VERIFIED with exact tests (cell -> block id, z >= max, exponent/sign/mantissa of
//     each coordinate), and monotonicity of each coordinate in k extends the
//     verification from the landing point to every skipped position.  A failed
//     verification just means "no jump".  Skipped positions are counted as steps:
//     each was inside the grid, so the reference executed its height load there.
#include "device_common.hpp"
#include "render.hpp"

#pragma clang fp contract(off)

namespace hmrm {

namespace {

constexpr int kGroup = 4;       // U: positions per speculative group
constexpr int kMinLeap = 8;     // a jump shorter than this is not worth its bookkeeping

__device__ __forceinline__ uint32_t hi32(double v) { return (uint32_t)((unsigned long long)__double_as_longlong(v) >> 32); }
__device__ __forceinline__ uint32_t lo32(double v) { return (uint32_t)(unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ double f64_from_hi(uint32_t hi) { return __longlong_as_double((long long)((unsigned long long)hi << 32)); }

// Cell coordinate q with trunc(q) == trunc(v / grid_width) and identical range
// predicates (q > -1, q < N); v is x - c0.x or -(y - c0.y) with c0.x = c0.y = 0.0
// (hmap.cpp:968: v - 0.0 == v for every v, so the subtraction is elided).
//   GWM 0: grid_width == 1.0          -> q = v
//   GWM 1: grid_width = 2^k           -> q = v * 2^-k          (exact reciprocal)
//   GWM 2: any grid_width             -> q' = v * fl(1/gw) differs from v/gw by < 2^-50
//          relative; unless q' is within 2^-20 of an integer (or huge / NaN) both
//          truncate alike and compare alike with integers; otherwise divide for real.
template <int GWM>
__device__ __forceinline__ double cell_coord(double v, const DevFrame &f) {
	if (GWM == 0) return v;
	if (GWM == 1) return v * f.inv_grid_width;
	double q = v * f.inv_grid_width;
	const double fr = q - __builtin_rint(q);
	if (!(__builtin_fabs(fr) > 0x1p-20) || !(__builtin_fabs(q) < 0x1p28)) q = v / f.grid_width;
	return q;
}

// Per-step increment of coordinate p inside p's binade.  Returns false when p + s
// leaves the binade / changes sign, p is tiny or non-finite, or s lands on an exact
// rounding tie (then the increment depends on the parity of p).
__device__ __forceinline__ bool binade_delta(double p, double s, double &delta) {
	const double p1 = p + s;
	const uint32_t hp = hi32(p), hp1 = hi32(p1);
	const uint32_t e = (hp >> 20) & 0x7ffu;
	const double d = p1 - p; // exact: both are multiples of u and |d| < 2^53 u
	delta = d;
	if (((hp ^ hp1) >> 20) != 0) return false; // sign or exponent changed
	if (e < 128u || e > 1900u) return false;    // subnormal/tiny, huge, inf, NaN
	const double err = s - d;                   // exact low part of s, |err| <= u/2
	const double half_u = f64_from_hi((e - 53u) << 20);
	return __builtin_fabs(err) != half_u;
}

// p_n = p + n*delta is trustworthy iff it is still in p's binade with the same sign, and
// -- when moving towards zero -- not exactly on the binade's lower boundary (the
// step that produced it could have rounded on the finer grid below 2^E).
__device__ __forceinline__ bool binade_landing_ok(double p, double pn, double delta) {
	if (((hi32(p) ^ hi32(pn)) >> 20) != 0) return false;
	return delta == 0.0 || (hi32(pn) & 0xfffffu) != 0u || lo32(pn) != 0u;
}

// Upper estimate of how many steps keep |p| inside its binade (not exact: verified later).
__device__ __forceinline__ double binade_room(double p, double delta, double rcp_abs_step) {
	if (delta == 0.0) return 0x1p40;
	const uint32_t hp = hi32(p);
	const double ap = __builtin_fabs(p);
	const double lo = f64_from_hi(hp & 0x7ff00000u);
	const bool away = ((hi32(delta) ^ hp) >> 31) == 0; // same sign: |p| grows
	const double dist = away ? (lo + lo) - ap : ap - lo;
	return dist * rcp_abs_step;
}

} // namespace

template <int PROJ, bool STATS, int GWM, bool LEAP>
__global__ __launch_bounds__(256) void k_render_fast(const DevFrame f, const RowMap rows,
                                                     const double *__restrict__ thr,
                                                     const uint32_t *__restrict__ cmap,
                                                     uint32_t *__restrict__ out, int64_t out_stride_px,
                                                     int tiles_x, StatsOut st) {
	const PixelId pid = pixel_of_lane(f, rows, tiles_x);
	unsigned long long my_steps = 0;
	uint32_t my_hit = 0, my_cap = 0;
	uint32_t dg_attempts = 0, dg_leaps = 0, dg_groups = 0; // STATS-only diagnostics
	unsigned long long dg_leaped = 0;

	if (pid.live) {
		const DevRay ray = make_ray<PROJ>(f, pid.px, pid.py);
		const double d = slab_distance(ray, f);
		if (STATS && st.entry_d) st.entry_d[(int64_t)pid.py * f.screen_w + pid.px] = d;

		uint32_t rgba = 0;
		bool real_hit = false;

		if (!(d == __builtin_huge_val()) && !(d < 0.0)) { // intersection(), AABB.cpp:33-44
			double x = ray.px + d * ray.dx;
			double y = ray.py + d * ray.dy;
			double z = ray.pz + d * ray.dz;
			x = x + f.nudge * ray.dx; // hmap.cpp:998
			y = y + f.nudge * ray.dy;
			z = z + f.nudge * ray.dz;
			const double sx = f.step_dist * ray.dx; // hmap.cpp:1037, loop invariant
			const double sy = f.step_dist * ray.dy;
			const double sz = f.step_dist * ray.dz;
			const double wlim = (double)f.map_w, hlim = (double)f.map_h;
			int64_t budget = f.step_cap;

			// leap state
			int lev = kMipLevels - 1; // start coarse: rays enter the box high above the terrain
			int cooldown = 0, penalty = 1;
			double rsx = 0, rsy = 0, rsz = 0; // ~1/|step| in cell units (x,y) / world units (z)
			if (LEAP) {
				const double inv = (GWM == 0) ? 1.0 : f.inv_grid_width;
				rsx = __builtin_amdgcn_rcp(__builtin_fabs(sx * inv));
				rsy = __builtin_amdgcn_rcp(__builtin_fabs(sy * inv));
				rsz = __builtin_amdgcn_rcp(__builtin_fabs(sz));
			}

			bool done = false;
			while (!done) {
				// ---------------------------------------------------------- leap
				if (LEAP) {
					bool tried = false, leaped = false;
					if (cooldown > 0) {
						--cooldown;
					} else {
						tried = true;
						if (STATS) ++dg_attempts;
						const double qx = cell_coord<GWM>(x, f), qy = cell_coord<GWM>(-y, f);
						double dx_, dy_, dz_;
						const bool okx = binade_delta(x, sx, dx_);
						const bool oky = binade_delta(y, sy, dy_);
						const bool okz = binade_delta(z, sz, dz_);
						if (qx > -1.0 && qx < wlim && qy > -1.0 && qy < hlim && okx && oky && okz) {
							const int gx = (int)qx, gy = (int)qy;
							const int sh = lev == 0 ? kMipShift[0] : (lev == 1 ? kMipShift[1] : kMipShift[2]);
							const int bx = gx >> sh, by = gy >> sh;
							const double *mp = lev == 0 ? f.mip[0] : (lev == 1 ? f.mip[1] : f.mip[2]);
							const int mw = lev == 0 ? f.mip_w[0] : (lev == 1 ? f.mip_w[1] : f.mip_w[2]);
							const double m = mp[(int64_t)by * mw + bx];
							if (z >= m) {
								// estimates (cell units laterally); every one is an over-estimate at
								// worst by rounding -- the landing point is verified below
								const double bsz = (double)(1 << sh);
								const double bx0 = (double)(bx << sh), by0 = (double)(by << sh);
								// x grows with sx; the y cell index grows when y decreases (qy = -y/gw)
								double room = 0x1p30;
								if (sx != 0.0) room = __builtin_fmin(room, (sx > 0.0 ? (bx0 + bsz) - qx : qx - bx0) * rsx);
								if (sy != 0.0) room = __builtin_fmin(room, (sy < 0.0 ? (by0 + bsz) - qy : qy - by0) * rsy);
								if (sz < 0.0) room = __builtin_fmin(room, (z - m) * rsz);
								room = __builtin_fmin(room, binade_room(x, dx_, rsx * ((GWM == 0) ? 1.0 : f.inv_grid_width)));
								room = __builtin_fmin(room, binade_room(y, dy_, rsy * ((GWM == 0) ? 1.0 : f.inv_grid_width)));
								room = __builtin_fmin(room, binade_room(z, dz_, rsz));
								room = __builtin_fmin(room, (double)budget);
								const int n = (int)(room * 0.998) - 1;
								if (n >= kMinLeap) {
									const double nn = (double)n;
									const double xn = x + nn * dx_, yn = y + nn * dy_, zn = z + nn * dz_;
									const double qxn = cell_coord<GWM>(xn, f), qyn = cell_coord<GWM>(-yn, f);
									bool ok = qxn > -1.0 && qxn < wlim && qyn > -1.0 && qyn < hlim;
									ok = ok && (((int)qxn) >> sh) == bx && (((int)qyn) >> sh) == by;
									ok = ok && zn >= m;
									ok = ok && binade_landing_ok(x, xn, dx_) && binade_landing_ok(y, yn, dy_) &&
									     binade_landing_ok(z, zn, dz_);
									if (ok) {
										x = xn; y = yn; z = zn;
										budget -= n;
										if (STATS) { my_steps += (unsigned)n; dg_leaped += (unsigned)n; ++dg_leaps; }
										leaped = true;
									}
								}
							}
						}
					}
					if (tried) {
						if (leaped) {
							penalty = 1;
							if (lev < kMipLevels - 1) ++lev;
						} else if (lev > 0) {
							--lev;
						} else {
							cooldown = penalty;
							if (penalty < 8) penalty <<= 1;
						}
					}
					if (leaped) continue; // try the next block straight away
				}

				// --------------------------------------------- speculative group
				if (STATS) ++dg_groups;
				double X[kGroup], Y[kGroup], Z[kGroup], T[kGroup];
				int cell[kGroup];
				bool inb[kGroup];
				X[0] = x; Y[0] = y; Z[0] = z;
#pragma unroll
				for (int j = 1; j < kGroup; ++j) {
					X[j] = X[j - 1] + sx;
					Y[j] = Y[j - 1] + sy;
					Z[j] = Z[j - 1] + sz;
				}
#pragma unroll
				for (int j = 0; j < kGroup; ++j) {
					const double qx = cell_coord<GWM>(X[j], f), qy = cell_coord<GWM>(-Y[j], f);
					// hmap.cpp:1006-1011: (int)q >= 0 <=> q > -1 ; (int)q < N <=> q < N ; NaN -> break
					inb[j] = qx > -1.0 && qx < wlim && qy > -1.0 && qy < hlim;
					const int gx = (int)(inb[j] ? qx : 0.0), gy = (int)(inb[j] ? qy : 0.0);
					cell[j] = gy * f.map_w + gx;
				}
#pragma unroll
				for (int j = 0; j < kGroup; ++j) T[j] = thr[cell[j]]; // hmap.cpp:1013-1014 (+ c0.z)
#pragma unroll
				for (int j = 0; j < kGroup; ++j) {
					if (done) break;
					if (!inb[j]) { done = true; break; }
					if (budget <= 0) { my_cap = 1; done = true; break; }
					--budget;
					if (STATS) my_steps += 1;
					if (Z[j] < T[j]) { // hmap.cpp:1016
						rgba = shade_hit(f, cmap[cell[j]]);
						real_hit = true;
						done = true;
						break;
					}
				}
				if (!done) {
					x = X[kGroup - 1] + sx;
					y = Y[kGroup - 1] + sy;
					z = Z[kGroup - 1] + sz;
				}
			}
		}

		if (!real_hit) rgba = shade_miss(f, ray.dz);
		else my_hit = 1;
		out[(int64_t)pid.lrow * out_stride_px + pid.px] = rgba;
		if (STATS && st.steps_per_pixel)
			st.steps_per_pixel[(int64_t)pid.py * f.screen_w + pid.px] =
			    my_steps > 0xffffffffull ? 0xffffffffu : (uint32_t)my_steps;
	}
	publish_counters<STATS>(st, my_steps, my_hit, my_cap);
	if (STATS) {
		unsigned long long a = dg_attempts, l = dg_leaps, g = dg_groups, s = dg_leaped;
		for (int off = 32; off > 0; off >>= 1) {
			a += __shfl_xor(a, off);
			l += __shfl_xor(l, off);
			g += __shfl_xor(g, off);
			s += __shfl_xor(s, off);
		}
		if ((threadIdx.x & 63) == 0) {
			if (a) atomicAdd(&st.counters[4], a);
			if (l) atomicAdd(&st.counters[5], l);
			if (g) atomicAdd(&st.counters[6], g);
			if (s) atomicAdd(&st.counters[7], s);
		}
	}
}

// Max pyramid: dst(bx,by) = max over the factor x factor block of src (NaN ignored).
__global__ __launch_bounds__(256) void k_build_mip(const double *__restrict__ src, int src_w, int src_h,
                                                   double *__restrict__ dst, int dst_w, int dst_h, int factor) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (int64_t)dst_w * dst_h) return;
	const int bx = (int)(i % dst_w), by = (int)(i / dst_w);
	double m = -__builtin_huge_val();
	for (int yy = by * factor; yy < (by + 1) * factor && yy < src_h; ++yy)
		for (int xx = bx * factor; xx < (bx + 1) * factor && xx < src_w; ++xx) {
			const double v = src[(int64_t)yy * src_w + xx];
			if (v > m) m = v;
		}
	dst[i] = m;
}

hipError_t launch_build_mip(const double *d_src, int src_w, int src_h, double *d_dst, int dst_w, int dst_h,
                            int factor, hipStream_t stream) {
	const int64_t n = (int64_t)dst_w * dst_h;
	hipLaunchKernelGGL(k_build_mip, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_src, src_w,
	                   src_h, d_dst, dst_w, dst_h, factor);
	return hipGetLastError();
}

template <int PROJ, bool STATS, int GWM, bool LEAP>
static void launch_one(const DevFrame &f, const RowMap &rows, const double *d_thr, const uint32_t *d_cmap,
                       uint32_t *d_out, int64_t out_stride_px, StatsOut st, dim3 grid, int tiles_x,
                       hipStream_t stream) {
	hipLaunchKernelGGL((k_render_fast<PROJ, STATS, GWM, LEAP>), grid, dim3(256), 0, stream, f, rows, d_thr,
	                   d_cmap, d_out, out_stride_px, tiles_x, st);
}

template <int PROJ, bool STATS, int GWM>
static void launch_leap(bool leap, const DevFrame &f, const RowMap &rows, const double *d_thr,
                        const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px, StatsOut st, dim3 grid,
                        int tiles_x, hipStream_t stream) {
	if (leap) launch_one<PROJ, STATS, GWM, true>(f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream);
	else launch_one<PROJ, STATS, GWM, false>(f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream);
}

template <int PROJ, bool STATS>
static void launch_gwm(bool leap, const DevFrame &f, const RowMap &rows, const double *d_thr,
                       const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px, StatsOut st, dim3 grid,
                       int tiles_x, hipStream_t stream) {
	switch (f.grid_mode) {
	case 0: launch_leap<PROJ, STATS, 0>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream); break;
	case 1: launch_leap<PROJ, STATS, 1>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream); break;
	default: launch_leap<PROJ, STATS, 2>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream); break;
	}
}

template <bool STATS>
static void launch_proj(bool leap, const DevFrame &f, const RowMap &rows, const double *d_thr,
                        const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px, StatsOut st, dim3 grid,
                        int tiles_x, hipStream_t stream) {
	switch (f.projection) {
	case 1: launch_gwm<1, STATS>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream); break;
	case 2: launch_gwm<2, STATS>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream); break;
	default: launch_gwm<3, STATS>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream); break;
	}
}

hipError_t launch_render_fast(const DevFrame &f, const RowMap &rows, const double *d_thr,
                              const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px,
                              unsigned long long *d_counters, uint32_t *d_steps, double *d_entry, bool stats,
                              bool leap, hipStream_t stream) {
	const int tiles_x = (f.screen_w + 15) / 16;
	const int tiles_y = (rows.local_rows + 15) / 16;
	if (tiles_x <= 0 || tiles_y <= 0) return hipSuccess;
	const dim3 grid((unsigned)((int64_t)tiles_x * tiles_y));
	StatsOut st{d_counters, d_steps, d_entry};
	if (stats) launch_proj<true>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream);
	else launch_proj<false>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_x, stream);
	return hipGetLastError();
}

} // namespace hmrm
