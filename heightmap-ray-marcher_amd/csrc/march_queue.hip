// march_queue.hip -- second pass of the two-pass march: finishes the rays the tile kernel
// (render_fast.hip) handed over, with SEVERAL LANES PER RAY.
//
// Why.  In the tile kernel a lane owns a pixel, so the time of a wave is that of its longest ray:
// a ray that skims the terrain near the horizon takes ~100 dependent loop trips of ~350 instructions
// while its neighbours are long done (lane slots wasted), and the few such waves end the launch alone
// on their SIMDs (critical path).  Both come from the same fact -- one ray's march is a serial chain
// in that kernel -- and it need not be one:
//
// Inside one binade the reference's sequentially accumulated positions (main/hmap.cpp:1037) are
// p_k = p_0 + k*delta EXACTLY (render_fast.hip header, leap_common.hpp), so position k can be computed
// directly, by any lane, and the conditions that end a ray at step k -- cell outside the grid
// (hmap.cpp:1006) or z < threshold (hmap.cpp:1016) -- can be evaluated for many k at once.  Here a
// group of kCoopLanes lanes works on one ray.  In a round lane j looks at the n consecutive steps
// [j*n, (j+1)*n):
//   * level >= 0 (window maxima, as in the tile kernel): it computes the first and the last
//     position of its segment in closed form, verifies both exactly (same binade, off the boundary:
//     axis_landing_ok), takes the pyramid window around the first cell and proves the segment
//     harmless: both cells inside the grid and the window, min(z_first, z_last) >= window maximum,
//     budget left.  Every coordinate is monotone in k, so the proof covers all n positions.
//   * level -1 (n = 1): it tests its one position like the reference does -- range test, step cap,
//     threshold of its cell.
// A ballot finds the first lane e whose segment is NOT proven harmless.  Steps [0, e*n) are then
// known to be plain in-grid misses: the ray advances by e*n steps -- to position e*n - 1 in closed
// form (verified by lane e-1) plus ONE real step, exactly what the reference computes next -- and
// they count as steps, since the reference loaded a height at each of them.  If lane e was at level
// -1 its verdict is final (out of the grid: miss; cap; hit) unless its position could not be trusted
// (left the binade), in which case the round only advances.  Otherwise the level goes down (or up
// after a round with no objection).  When the coordinates are not inside steady binades (crossing a
// power of two, a rounding tie on the wrong parity, tiny values) lanes 0..3 test four real sequential
// steps instead.  Lane 0's first position is always the ray's real position, so every round
// either advances, ends the ray or lowers the level: the loop terminates, by the step cap at the latest.
//
// Results are bit-identical to the tile kernel finishing the same ray: same pixel, same step count.
#include "device_common.hpp"
#include "leap_common.hpp"
#include "render.hpp"

#pragma clang fp contract(off)

namespace hmrm {

namespace {

#ifndef HMRM_COOP_LANES
#define HMRM_COOP_LANES 16
#endif
constexpr int kCoopLanes = HMRM_COOP_LANES;           // lanes per ray: 8, 16, 32 or 64
constexpr int kCoopGroups = 64 / kCoopLanes;           // rays per wave
constexpr int kQueueBlockThreads = 256;
static_assert(kCoopLanes == 8 || kCoopLanes == 16 || kCoopLanes == 32 || kCoopLanes == 64, "lanes per ray");

__device__ __forceinline__ DevRay make_ray_any(const DevFrame &f, int px, int py) {
	switch (f.projection) {
	case 1: return make_ray<1>(f, px, py);
	case 2: return make_ray<2>(f, px, py);
	default: return make_ray<3>(f, px, py);
	}
}

// the ballot bits of this lane's group, moved down to bit 0
__device__ __forceinline__ unsigned long long group_bits(unsigned long long ballot, int gshift) {
	const unsigned long long b = ballot >> gshift;
	return kCoopLanes == 64 ? b : (b & ((1ull << (kCoopLanes & 63)) - 1ull));
}

} // namespace

template <bool STATS, int GWM>
__global__ __launch_bounds__(kQueueBlockThreads) void k_march_queue(const DevFrame f, const RowMap rows,
                                                                    const double *__restrict__ thr,
                                                                    const uint32_t *__restrict__ cmap,
                                                                    uint32_t *__restrict__ out, int64_t out_stride_px,
                                                                    const RayQueue q, uint32_t *__restrict__ next_count,
                                                                    StatsOut st) {
	constexpr int L = kCoopLanes, G = kCoopGroups;
	const int lane = (int)(threadIdx.x & 63u);
	const int j = lane & (L - 1);       // which segment of the ray's round this lane looks at
	const int g = lane / L;             // which of the wave's rays
	const int gshift = g * L;
	unsigned count = *q.count;
	if (count > q.capacity) count = q.capacity;
	if (blockIdx.x == 0 && threadIdx.x == 0 && next_count) *next_count = 0u;

	const unsigned waves_per_block = kQueueBlockThreads / 64;
	const unsigned wave = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
	const unsigned nwaves = gridDim.x * waves_per_block;

	unsigned long long my_steps = 0;
	uint32_t my_hit = 0, my_cap = 0;
	unsigned long long dg_rounds = 0, dg_level_rounds = 0, dg_cell_rounds = 0, dg_leaped = 0;

	const unsigned wlim = (unsigned)f.map_w, hlim = (unsigned)f.map_h;
	const int budget0 = f.step_cap > 0x7fffffff ? 0x7fffffff : (int)f.step_cap;
	const double gwid = (GWM == 0) ? 1.0 : f.grid_width;

	for (unsigned base = wave * G; base < count; base += nwaves * G) {
		const unsigned entry = base + (unsigned)g;
		bool gdone = !(entry < count);
		const unsigned ei = gdone ? 0u : entry;
		double x = q.x[ei], y = q.y[ei], z = q.z[ei];
		const int px = q.px[ei], lrow = q.lrow[ei];
		int budget = q.budget[ei], lev = q.lev[ei];
		const int py = frame_row_of(rows, lrow);
		const DevRay ray = make_ray_any(f, px, py);
		const double sx = f.step_dist * ray.dx; // hmap.cpp:1037, as in the tile kernel
		const double sy = f.step_dist * ray.dy;
		const double sz = f.step_dist * ray.dz;
		const int offx = sx < 0.0 ? 1 : 0, offy = sy > 0.0 ? 1 : 0; // window choice, as in the tile kernel
		Axis ax, ay, az;
		ax.key = ay.key = az.key = 0xfffffffeu; // never matches: forces the first refresh
		ax.delta = ay.delta = az.delta = 0.0;
		ax.lim = ay.lim = az.lim = 0.0;
		ax.rdel = ay.rdel = az.rdel = 0.0;
		int crawl = 0; // cell rounds still to do before windows are tried again (performance only)

		while (!gdone) {
			// ------------------------------------------------ A: per ray (all lanes of the group alike)
			if ((hi32(x) >> 20) != ax.key) axis_refresh(ax, x, sx);
			if ((hi32(y) >> 20) != ay.key) axis_refresh(ay, y, sy);
			if ((hi32(z) >> 20) != az.key) axis_refresh(az, z, sz);
			const bool exact = ax.key != 0xffffffffu && ay.key != 0xffffffffu && az.key != 0xffffffffu;
			int lv = -1; // level of this round: -1 single cells, 0..kMipLevels-1 windows, kTopLevel whole map
			int n = 1;   // steps per lane
			if (exact && lev >= 0) {
				// estimate (only that: every lane verifies its own segment) of the steps left before a
				// binade boundary, the edge of the map or the step cap
				double room = (ax.lim - x) * ax.rdel;
				room = __builtin_fmin(room, (ay.lim - y) * ay.rdel);
				room = __builtin_fmin(room, (az.lim - z) * az.rdel);
				const double ex = sx > 0.0 ? (double)f.map_w * gwid : 0.0;
				const double ey = sy < 0.0 ? -(double)f.map_h * gwid : 0.0;
				room = __builtin_fmin(room, sx != 0.0 ? (ex - x) * ax.rdel : 0x1p40);
				room = __builtin_fmin(room, sy != 0.0 ? (ey - y) * ay.rdel : 0x1p40);
				room = __builtin_fmin(room, (double)budget);
				room = __builtin_fmin(__builtin_fmax(room, 0.0), 0x1p28);
				const int k_est = (int)(room * 0.998) - 1;
				int per = (k_est + L - 1) / L;
				if (lev < kTopLevel) {
					// a segment that moves at most half a window sideways stays inside the window picked
					// at its first cell (windows are placed every half window)
					const double lat = __builtin_fmax(__builtin_fabs(ax.delta), __builtin_fabs(ay.delta));
					const double half = (double)(1 << (kLevelStep * lev + 1)) * gwid;
					const double nl = lat > 0.0 ? half * __builtin_amdgcn_rcp(lat) : 0x1p28;
					const int nli = (int)__builtin_fmin(nl, 0x1p24);
					per = per < nli ? per : nli;
				}
				if (per > (1 << 24)) per = 1 << 24;
				if (per >= 2) {
					lv = lev;
					n = per;
				}
			}
			if (STATS) {
				dg_rounds += (j == 0) ? 1u : 0u;
				dg_level_rounds += (j == 0 && lv >= 0) ? 1u : 0u;
				dg_cell_rounds += (j == 0 && lv < 0) ? 1u : 0u;
			}

			// ------------------------------------------------ B: per lane
			const int kf = j * n, kl = kf + n - 1;
			double xf, yf, zf, xl, yl, zl;
			double x3 = x, y3 = y, z3 = z; // (real-step rounds: position of step 3)
			bool valid;
			if (exact) {
				const double df = (double)kf, dl = (double)kl;
				xf = x + df * ax.delta;
				yf = y + df * ay.delta;
				zf = z + df * az.delta;
				xl = x + dl * ax.delta;
				yl = y + dl * ay.delta;
				zl = z + dl * az.delta;
				// (lane 0's first position is the ray's own: kf = 0 adds an exact zero)
				valid = j == 0 || (axis_landing_ok(ax, xf) && axis_landing_ok(ay, yf) && axis_landing_ok(az, zf));
				valid = valid && (n == 1 || (axis_landing_ok(ax, xl) && axis_landing_ok(ay, yl) && axis_landing_ok(az, zl)));
			} else {
				// four real steps, one per lane 0..3 (the reference's sequential adds)
				const double x1 = x + sx, y1 = y + sy, z1 = z + sz;
				const double x2 = x1 + sx, y2 = y1 + sy, z2 = z1 + sz;
				x3 = x2 + sx;
				y3 = y2 + sy;
				z3 = z2 + sz;
				xf = j == 0 ? x : (j == 1 ? x1 : (j == 2 ? x2 : x3));
				yf = j == 0 ? y : (j == 1 ? y1 : (j == 2 ? y2 : y3));
				zf = j == 0 ? z : (j == 1 ? z1 : (j == 2 ? z2 : z3));
				xl = xf;
				yl = yf;
				zl = zf;
				valid = j < 4;
			}
			bool near = false;
			double qxf = cell_coord_fast<GWM>(xf, f, near), qyf = cell_coord_fast<GWM>(-yf, f, near);
			double qxl = cell_coord_fast<GWM>(xl, f, near), qyl = cell_coord_fast<GWM>(-yl, f, near);
			if (GWM == 2 && near) {
				qxf = xf / f.grid_width;
				qyf = -yf / f.grid_width;
				qxl = xl / f.grid_width;
				qyl = -yl / f.grid_width;
			}
			const int gxf = cvt_i32_sat(qxf), gyf = cvt_i32_sat(qyf); // hmap.cpp:1001-1004
			const int gxl = cvt_i32_sat(qxl), gyl = cvt_i32_sat(qyl);
			const bool inbf = (unsigned)gxf < wlim && (unsigned)gyf < hlim; // hmap.cpp:1006-1011
			const bool inbl = (unsigned)gxl < wlim && (unsigned)gyl < hlim;
			const int cellf = inbf ? gyf * f.map_w + gxf : 0;

			bool clear;
			int cls; // why not: cell rounds 0 untrusted position, 1 out of the grid, 2 step cap, 3 hit;
			         // window rounds 0 the height bound, 1 anything else
			if (lv < 0) {
				const double t = thr[cellf];        // hmap.cpp:1013-1014 (+ c0.z)
				const bool capped = kf >= budget;   // the cap is checked after the range test, before the load
				const bool hit = zf < t;            // hmap.cpp:1016
				clear = valid && inbf && !capped && !hit;
				cls = !valid ? 0 : (!inbf ? 1 : (capped ? 2 : 3));
			} else {
				double m = f.thr_max;
				bool contained = true;
				if (lv < kTopLevel) {
					const int hs = kLevelStep * lv + 1;
					int ix = (gxf >> hs) - offx, iy = (gyf >> hs) - offy;
					ix = ix < 0 ? 0 : ix;
					iy = iy < 0 ? 0 : iy;
					const int mw = (f.map_w + (1 << hs) - 1) >> hs;
					int loff = 0;
#pragma unroll
					for (int l = 1; l < kMipLevels; ++l) loff = lv == l ? f.mip_off[l] : loff;
					m = (double)f.mipbuf[inbf ? loff + iy * mw + ix : 0];
					const int wx0 = ix << hs, wy0 = iy << hs;
					const int span_x = min(2 << hs, f.map_w - wx0), span_y = min(2 << hs, f.map_h - wy0);
					contained = (unsigned)(gxl - wx0) < (unsigned)span_x && (unsigned)(gyl - wy0) < (unsigned)span_y;
				}
				const bool high = __builtin_fmin(zf, zl) >= m; // (z is monotone along the segment; NaN never is)
				const bool rest = valid && inbf && inbl && contained && kl < budget;
				clear = rest && high;
				cls = rest ? 0 : 1;
			}

			// ------------------------------------------------ C: first objection of the group
			const unsigned long long objections = group_bits(__ballot(!clear), gshift);
			const unsigned long long c0 = group_bits(__ballot((cls & 1) != 0), gshift);
			const unsigned long long c1 = group_bits(__ballot((cls & 2) != 0), gshift);
			const int e = objections ? (int)__ffsll((long long)objections) - 1 : L;
			const int cls_e = e < L ? (int)((c0 >> e) & 1ull) | ((int)((c1 >> e) & 1ull) << 1) : 0;

			if (lv < 0) {
				if (e == L || cls_e == 0) {
					// steps 0..e-1 were plain in-grid misses; step e is one real step after step e-1
					// (e >= 1: lane 0 always trusts its position; without steady binades e == 4)
					const double dp = (double)(e - 1);
					x = (exact ? x + dp * ax.delta : x3) + sx;
					y = (exact ? y + dp * ay.delta : y3) + sy;
					z = (exact ? z + dp * az.delta : z3) + sz;
					budget -= e;
					if (e == L) {
						// nothing in L cells: try windows again, unless they keep failing right here
						if (crawl > 0) --crawl;
						else lev = lev < 0 ? 0 : lev;
					}
				} else {
					// the ray ends at step e: lane e holds the cell and writes the pixel
					const bool hit = cls_e == 3;
					budget -= e + (hit ? 1 : 0);
					if (j == e) {
						uint32_t rgba;
						if (hit) {
							rgba = shade_hit(f, cmap[cellf]); // hmap.cpp:1018-1031
							my_hit += 1;
						} else {
							rgba = shade_miss(f, ray.dz); // hmap.cpp:1041-1057
							my_cap += cls_e == 2 ? 1u : 0u;
						}
						out[(int64_t)lrow * out_stride_px + px] = rgba;
						const unsigned steps = (unsigned)(budget0 - budget);
						if (STATS) {
							my_steps += steps;
							if (st.steps_per_pixel) st.steps_per_pixel[(int64_t)py * f.screen_w + px] = steps;
						}
					}
					gdone = true;
				}
			} else {
				if (e > 0) {
					const int adv = e * n;
					const double dp = (double)(adv - 1);
					x = (x + dp * ax.delta) + sx;
					y = (y + dp * ay.delta) + sy;
					z = (z + dp * az.delta) + sz;
					budget -= adv;
					if (STATS && j == 0) dg_leaped += (unsigned)adv;
				}
				if (e == L) {
					lev = lv < kTopLevel ? lv + 1 : lv; // no objection at all: coarser windows next
					crawl = 0;
				} else if (cls_e == 0 || e == 0) {
					// the height bound of segment e objected (or nothing moved): finer windows; below the
					// finest, single cells -- for longer each time windows fail right after a cell round
					lev = lv - 1;
					if (lev < 0 && e == 0) crawl = crawl < 3 ? crawl + 1 : 3;
				}
			}
		}
	}

	if (STATS) {
		publish_counters<true>(st, my_steps, my_hit, my_cap);
		unsigned long long a = dg_level_rounds, l = dg_level_rounds, c = dg_cell_rounds, s = dg_leaped;
		(void)dg_rounds;
		for (int off = 32; off > 0; off >>= 1) {
			a += __shfl_xor(a, off);
			l += __shfl_xor(l, off);
			c += __shfl_xor(c, off);
			s += __shfl_xor(s, off);
		}
		if ((threadIdx.x & 63) == 0) {
			if (a) atomicAdd(&st.counters[4], a); // window rounds count as look-ups ...
			if (l) atomicAdd(&st.counters[5], l); // ... and as jumps
			if (c) atomicAdd(&st.counters[6], c); // cell rounds count as groups
			if (s) atomicAdd(&st.counters[7], s);
		}
	} else if (my_cap) {
		atomicAdd(&st.counters[2], (unsigned long long)my_cap);
	}
}

template <bool STATS>
static void launch_q(const DevFrame &f, const RowMap &rows, const double *d_thr, const uint32_t *d_cmap, uint32_t *d_out,
                     int64_t out_stride_px, const RayQueue &q, uint32_t *d_next, StatsOut st, dim3 grid,
                     hipStream_t stream) {
	const dim3 block(kQueueBlockThreads);
	switch (f.grid_mode) {
	case 0:
		hipLaunchKernelGGL((k_march_queue<STATS, 0>), grid, block, 0, stream, f, rows, d_thr, d_cmap, d_out,
		                   out_stride_px, q, d_next, st);
		break;
	case 1:
		hipLaunchKernelGGL((k_march_queue<STATS, 1>), grid, block, 0, stream, f, rows, d_thr, d_cmap, d_out,
		                   out_stride_px, q, d_next, st);
		break;
	default:
		hipLaunchKernelGGL((k_march_queue<STATS, 2>), grid, block, 0, stream, f, rows, d_thr, d_cmap, d_out,
		                   out_stride_px, q, d_next, st);
		break;
	}
}

hipError_t launch_march_queue(const DevFrame &f, const RowMap &rows, const double *d_thr, const uint32_t *d_cmap,
                              uint32_t *d_out, int64_t out_stride_px, unsigned long long *d_counters,
                              uint32_t *d_steps, bool stats, const RayQueue &q, uint32_t *d_next_count,
                              hipStream_t stream) {
	// A fixed grid that fills the chip; the waves stride over the queue (its length is only known
	// on the device) and leave at once when there is nothing for them.
	const dim3 grid(2048);
	StatsOut st{d_counters, d_steps, nullptr};
	if (stats) launch_q<true>(f, rows, d_thr, d_cmap, d_out, out_stride_px, q, d_next_count, st, grid, stream);
	else launch_q<false>(f, rows, d_thr, d_cmap, d_out, out_stride_px, q, d_next_count, st, grid, stream);
	return hipGetLastError();
}

} // namespace hmrm
