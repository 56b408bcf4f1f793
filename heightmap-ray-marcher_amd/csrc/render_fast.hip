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
//     delta = round_u(s) for every p of that binade (round-to-nearest; on an exact
//     tie the increment is constant too once p has the steady parity: axis_refresh).  Hence the
//     reference's sequential accumulation satisfies p_k = p_0 + k*delta EXACTLY, and
//     both the product k*delta and the sum are exact in fp64.  A pyramid of window
//     maxima over the hit thresholds then lets a ray jump over n steps at once when
//     all n skipped positions provably (a) stay inside the window whose maximum was
//     looked up and (b) stay at or above that maximum (no hit possible: hmap.cpp:1016
//     needs z < threshold), with (c) all three coordinates inside their binades.
//     (a) and (b): the jump length is only ESTIMATED (approximate reciprocals); the landing
//     point is then VERIFIED with exact tests (cell inside the window, z >= max), and
//     monotonicity of each coordinate in k extends the verification from the landing point to
//     every skipped position.  A failed verification just means "no jump".
//     (c): how many further steps stay strictly inside a coordinate's binade is established ONCE
//     per binade (axis_refresh: estimate, shortened, verified at its far end) and counted down
//     with every step leaped or marched (Axis::left): the multiplied part of a jump never exceeds the
//     three counts.  A jump of n steps is n - 1 multiplied steps and then ONE real step fl(p + s) from that
//     exact position (HMRM_CROSS): where a binade's end cut the jump short, that step is the one
//     that carries the coordinate into its next binade -- otherwise a whole group of real steps would
//     have to be marched at every binade boundary (half of all groups before this was done).  The
//     end point after the real step is what (a) and (b) are verified for.
//     Skipped positions are counted as steps: each was inside the grid, so the reference
//     executed its height load there.
//
// Pyramid layout (built by k_build_mip*): level l holds maxima of S x S-cell windows,
// S = 4, 8, 16, .. 256, placed every S/2 cells on the 4-cell level and every S/4 cells above it
// (overlapping), so that a ray can always pick a window in which it has at least S/2 (3S/4) cells of
// room ahead.  Values are floats rounded UP (a larger bound is always safe).  Above them: the whole
// map, the one window of a top plane.  A ray moves two levels at a time until it has made kAdaptAfter
// jumps, then one at a time (performance only: any level sequence gives the same pixels).
#ifdef HMRM_TIMELINE
#include <cstdio>
#include <cstdlib>
#endif
#include "device_common.hpp"
#include "leap_common.hpp"
#include "leap_diag.hpp"
#include "render.hpp"

#pragma clang fp contract(off)

namespace hmrm {

namespace {

#ifndef HMRM_GROUP
#define HMRM_GROUP 4
#endif
constexpr int kGroup = HMRM_GROUP; // U: positions per speculative group of the production kernel
// ... and of the plain-groups instantiation (LEAP = false: no leaps, every height load of the reference is executed --
// what the scene's probe picks on content that admits no jumps, and the kernel SURVEY 8(d)'s byte roofline is defined
// for).  That kernel waits for its gathers 70 % of the time at 17 % VALU busy (profiles/r05_C3_group_rocprof.txt): more
// loads in flight per lane pay until the registers cost resident waves -- C3 2.61 ms with 4, 2.37 with 6, 2.57 with 8
// (profiles/r05_raw/group_len_ab.txt).  Deciding the hit test from a float copy of the table, doubles only where that is
// unsafe, halves the bytes and changes little: the gathers are bound by lanes, not bytes (r05_experiments.txt section 4).
#ifndef HMRM_GROUP_PLAIN
#define HMRM_GROUP_PLAIN 6
#endif
constexpr int kGroupPlain = HMRM_GROUP_PLAIN;
// ... and of the record kernel; the most refusals in a row its back-off counts (attempts every 2^n-th trip at most)
#ifndef HMRM_GROUP_REC
#define HMRM_GROUP_REC 6
#endif
constexpr int kGroupRec = HMRM_GROUP_REC;
#ifndef HMRM_REC_BACKOFF
#define HMRM_REC_BACKOFF 6
#endif
constexpr int kRecBackoff = HMRM_REC_BACKOFF;
#ifndef HMRM_MIN_LEAP
#define HMRM_MIN_LEAP 2
#endif
constexpr int kMinLeap = HMRM_MIN_LEAP; // a jump shorter than this is not worth its bookkeeping
#ifndef HMRM_UP_RATIO
#define HMRM_UP_RATIO (kLevelStep == 2 ? 4.0 : 2.0)
#endif
constexpr double kUpRatio = HMRM_UP_RATIO; // see the level policy in k_render_fast
// Per-ray adaptive level spacing (pyramids with windows doubling per level, HMRM_LEVEL_STEP=1): a ray moves two
// levels at a time -- windows of 4, 16, 64, 256 cells, which is what ordinary rays want (fewer level changes) --
// until it has made more than HMRM_ADAPT_AFTER successful jumps; from then on one level at a time, so that the few
// long rays skimming the terrain (the launch's tail) can use the 8-, 32- and 128-cell windows in between: where a
// 16-cell window clears the ray and the 64-cell one does not, the 32-cell one often does and the jump doubles.
// 0 = off (every ray one level at a time).  Performance only: any level sequence gives the same pixels.
#ifndef HMRM_ADAPT_AFTER
#define HMRM_ADAPT_AFTER 8
#endif
constexpr int kAdaptAfter = (kLevelStep == 1) ? HMRM_ADAPT_AFTER : 0;
constexpr bool kAdaptive = kAdaptAfter > 0;

// HMRM_EARLY_LOAD (default 1): order of an attempt -- the pyramid look-up is issued first, refreshes and lateral estimates
// run while it is in flight (see the attempt block).  0 = refresh first and estimates after the load, for A/B runs.
#ifndef HMRM_EARLY_LOAD
#define HMRM_EARLY_LOAD 1
#endif
constexpr bool kEarlyLoad = HMRM_EARLY_LOAD != 0;

// ---- bilinear quality mode (HMRM_BILINEAR; a build-side addition, not in the reference) ----
// Same definition, operation for operation, as oracle/hmrm_oracle.c "bilinear quality mode":
// values sit at cell centres; u = q - 0.5, t = u - floor(u), neighbours clamp(floor(u)) and
// clamp(floor(u) + 1); f = a + ty*(b - a), a = f00 + tx*(f10 - f00), b = f01 + tx*(f11 - f01).
struct Bil {
	int c00, c10, c01, c11; // the four neighbour cells (linear indices)
	double tx, ty;
};
__device__ __forceinline__ Bil bil_setup(double qx, double qy, int w, int h) {
	const double u = qx - 0.5, v = qy - 0.5;
	const double fu = __builtin_floor(u), fv = __builtin_floor(v);
	Bil b;
	b.tx = u - fu;
	b.ty = v - fv;
	const int iu = cvt_i32_sat(fu), iv = cvt_i32_sat(fv);
	const int i0 = min(max(iu, 0), w - 1), i1 = min(max(iu + 1, 0), w - 1);
	const int j0 = min(max(iv, 0), h - 1), j1 = min(max(iv + 1, 0), h - 1);
	b.c00 = j0 * w + i0; b.c10 = j0 * w + i1;
	b.c01 = j1 * w + i0; b.c11 = j1 * w + i1;
	return b;
}
__device__ __forceinline__ double bil_mix(const Bil &b, double f00, double f10, double f01, double f11) {
	const double a = f00 + b.tx * (f10 - f00);
	const double c = f01 + b.tx * (f11 - f01);
	return a + b.ty * (c - a);
}
// Colour at a hit: R,G,B interpolated and rounded with floor(f + 0.5); the alpha-0 rule
// (hmap.cpp:1020) keeps looking at the nearest cell.
__device__ __forceinline__ uint32_t shade_hit_bilinear(const DevFrame &f, const uint32_t *__restrict__ cmap,
                                                       int nearest_cell, const Bil &b) {
	const uint32_t tn = cmap[nearest_cell];
	if ((tn >> 24) == 0) return pack_rgba(f.bg[0], f.bg[1], f.bg[2]);
	const uint32_t t00 = cmap[b.c00], t10 = cmap[b.c10], t01 = cmap[b.c01], t11 = cmap[b.c11];
	uint32_t ch[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const int sh = 8 * k;
		double v = bil_mix(b, (double)((t00 >> sh) & 255u), (double)((t10 >> sh) & 255u),
		                   (double)((t01 >> sh) & 255u), (double)((t11 >> sh) & 255u)) + 0.5;
		if (v < 0.0) v = 0.0;
		else if (v > 255.0) v = 255.0;
		ch[k] = (uint32_t)(int)__builtin_floor(v);
	}
	return pack_rgba(ch[0], ch[1], ch[2]);
}

} // namespace

// Tool build only (-DHMRM_TIMELINE, tools/timeline.py): every wave of the production (non-instrumented) kernel
// stores its start and end time (s_memrealtime) and the XCD it ran on into a device buffer (not pinned host memory:
// 129 600 records per 0.1 ms launch were 24 GB/s of PCIe writes, which disturbed what they measured); the library
// copies the last launch's records out to $HMRM_TIMELINE_FILE when the process ends.  Not compiled into the product.
#ifdef HMRM_TIMELINE
struct TimelineRec { unsigned long long t0, t1; unsigned int xcc, pad; };
__device__ TimelineRec *g_timeline = nullptr;
#endif

#ifndef HMRM_MIN_WAVES
#define HMRM_MIN_WAVES 1
#endif
#ifdef HMRM_WAVES_PER_EU
#define HMRM_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu(HMRM_WAVES_PER_EU, HMRM_WAVES_PER_EU), amdgpu_num_vgpr(512 / HMRM_WAVES_PER_EU / 8 * 8)))
#else
#define HMRM_OCCUPANCY_ATTR
#endif
// SAMP: 0 nearest cell (the reference), 1 bilinear quality mode, 2 nearest cell with float thresholds
// (`thr` then points at the float copy of the table).
// One wave's 8 x 8 pixels: wave `wave` of the workgroup-sized tile (tile_x, grid row gy).  Returns the tile row rendered
// (-1: the grid row does not exist).  k_render_fast calls it with the tile = blockIdx (one tile per workgroup); round 4's
// persistent-tile experiment (resident waves pulling tiles from queue heads) called it in a loop and was 1.4-1.9 x slower:
// profiles/r04_experiments.txt section 1, code at commit 80527e9.
template <int PROJ, bool STATS, int GWM, int LEAP, int SAMP>
__device__ __forceinline__ int render_wave_tile(const DevFrame &f, const RowMap &rows, const double *__restrict__ thr,
                                                const uint32_t *__restrict__ cmap, uint32_t *__restrict__ out,
                                                int64_t out_stride_px, int tiles_y, const StatsOut &st, int tile_x, unsigned gy,
                                                int wave, int lane) {
	constexpr bool BILINEAR = SAMP == 1, F32 = SAMP == 2;
	constexpr bool REC = LEAP == 2;                     // leaps over window records instead of the pyramid (frame.hpp WindowRecord)
	constexpr int U = LEAP == 1 ? kGroup : (REC ? kGroupRec : kGroupPlain); // positions per speculative group
	static_assert(!REC || SAMP == 0, "records bound the nearest cell's double thresholds only");
	const float *__restrict__ thr32 = reinterpret_cast<const float *>(thr);
	const float *__restrict__ mip = BILINEAR ? f.mipbuf_bil : f.mipbuf; // the pyramid this sampling mode leaps on
	const PixelId pid = pixel_of_tile_lane(f, rows, tiles_y, tile_x, gy, wave, lane);
	LoopDiag<STATS> diag; // (empty unless STATS: leap_diag.hpp)
	diag.start();
	unsigned long long my_steps = 0;
	uint32_t my_hit = 0, my_cap = 0;

	if (pid.live) {
		const DevRay ray = make_ray<PROJ>(f, pid.px, pid.py);
		// most rays of a frame never touch the box: prove the miss cheaply where possible (the instrumented
		// variant reports d, so it takes no shortcut for misses); most of the others get their entry distance
		// from one division instead of six (slab_classify: the instrumented variant uses that path too, so the
		// parity tests compare its d with the oracle's bit for bit)
		double d = __builtin_huge_val();
		if (STATS || !slab_points_away<PROJ>(ray, f)) {
			const int verdict = slab_classify(ray, f, !STATS, &d);
			if (verdict == 0) d = slab_distance(ray, f);
		}
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
			const unsigned wlim = (unsigned)f.map_w, hlim = (unsigned)f.map_h;
			const int budget0 = f.step_cap > 0x7fffffff ? 0x7fffffff : (int)f.step_cap;
			int budget = budget0; // every step taken or leaped comes off it: steps so far = budget0 - budget
			// (int)NaN is INT_MIN on the reference's CPU: the first range test fails, the ray misses
			const bool entry_nan = x != x || y != y;

			// leap state
			// First level to look at: a finer window has a lower maximum, so the finest level whose
			// windows still leave the ray lateral room for its whole descent to the box floor is the
			// best one (steep rays: the finest level at once, instead of walking down from the top);
			// oblique rays start with the whole-map bound.  Performance only.
			int lev = kTopLevel;
			if (sz < 0.0) {
				const double descent = (z - f.c0[2]) * __builtin_amdgcn_rcp(-sz); // steps down to min_height
				const double lateral = descent * __builtin_fmax(__builtin_fabs(sx), __builtin_fabs(sy)) * (GWM == 0 ? 1.0 : f.inv_grid_width);
#pragma unroll
				for (int l = kMipLevels - 1; l >= 0; l -= (kAdaptive ? 2 : 1)) // windows every 1 << hs cells: at least that much room ahead
					lev = (l >= f.min_level && lateral <= (double)((win_strides(l) - 1) << mip_stride_shift(l))) ? l : lev;
			}
			if (REC) lev = lev == kTopLevel ? kTopLevel : kRecLevel; // (the record kernel knows the whole-map bound and the record level)
			int cooldown = 0, fails = 0;
			unsigned trip_no = 0; // (REC)
			int jumps = 0; // successful jumps so far (kAdaptive)
			Axis ax, ay, az;
			ax.key = ay.key = az.key = 0xfffffffeu; // never matches: forces the first refresh
			ax.delta = ay.delta = az.delta = 0.0;
			ax.lim = ay.lim = az.lim = 0.0;
			ax.rdel = ay.rdel = az.rdel = 0.0;
			ax.left = ay.left = az.left = -1; // (kStepsLeft: forces the first refresh)
			// window choice: step back one half-window when the cell index decreases along the ray
			const int offx = sx < 0.0 ? 1 : 0, offy = sy > 0.0 ? 1 : 0; // gy = trunc(-y/gw) falls when y grows
			const double gwid = (GWM == 0) ? 1.0 : f.grid_width;

			// The loop body is written branch-light on purpose: a wave executes every divergent
			// branch any of its lanes takes, and exec-mask juggling per `if` costs as much as
			// the arithmetic it guards.  Values are computed for all lanes and selected.
			bool done = entry_nan;
			while (!done) {
				bool skip_group = false;
				diag.begin_trip();
				// ---------------------------------------------------------- leap
				if (LEAP) {
					// (REC: after `fails` refusals in a row a ray attempts on every 2^fails-th trip of the WAVE's count -- rays that
					// back off do so in step, and on a map that admits no leaps the block is issued once in 64 trips)
					const bool attempt = REC ? (trip_no & ((1u << fails) - 1u)) == 0u : cooldown == 0;
					++trip_no;
					cooldown -= attempt ? 0 : 1;
					if (attempt) {
						diag.on_attempt();
						// Order of the block (HMRM_EARLY_LOAD): the window look-up depends on the position and the level only, so
						// its load is issued FIRST; the refreshes of stale coordinates and the estimates of lateral room, which
						// do not need the loaded maximum, then run while it is in flight.  For a wave alone on its SIMD -- the
						// long waves at the end of a launch -- a trip is one chain of dependent instructions, and this takes
						// the refreshes and ~20 instructions of the estimate out of the part that waits for the load.
						auto refresh_stale = [&]() {
							const bool stale_x = kStepsLeft ? ax.left < 0 : (hi32(x) >> 20) != ax.key;
							const bool stale_y = kStepsLeft ? ay.left < 0 : (hi32(y) >> 20) != ay.key;
							const bool stale_z = kStepsLeft ? az.left < 0 : (hi32(z) >> 20) != az.key;
							diag.on_refresh_check(f, stale_x, stale_y, stale_z);
							if (stale_x) axis_refresh(ax, x, sx);
							if (stale_y) axis_refresh(ay, y, sy);
							if (stale_z) axis_refresh(az, z, sz);
						};
						if (!kEarlyLoad) refresh_stale();
						bool near0 = false;
						const double qx = cell_coord_fast<GWM>(x, f, near0), qy = cell_coord_fast<GWM>(-y, f, near0);
						// General grid widths: a start or landing point within 2^-20 of a cell boundary would need the true quotient
						// to name its cell.  The START divides for real then (a rare, wave-uniform branch: a coordinate that never
						// moves -- orthographic rays along an axis -- can sit that close to a boundary for a whole ray).  The LANDING
						// point does not: its cell is gxn or gxn - 1 (below), and the jump is accepted when BOTH lie inside the window
						// -- two inlined divisions and their temporaries less at the block's point of highest register pressure.
						// (Refusing such landings outright is wrong for speed: with 1 / grid_width an integer -- 0.05, 0.01 -- every
						// power of two is a cell boundary, the landing point of a binade-limited jump is the first position behind
						// one, always the same whatever the start, and a ray whose crossing step ends within 2^-20 of it was
						// refused again and again until it had MARCHED there: 175 groups instead of one, the launch's last wave.)
						double qx2 = qx, qy2 = qy;
						if (GWM == 2 && __builtin_amdgcn_ballot_w64(near0) != 0ull) {
							qx2 = near0 ? x / f.grid_width : qx;
							qy2 = near0 ? -y / f.grid_width : qy;
						}
						const int gx = cvt_i32_sat(qx2), gy = GWM == 0 ? cvt_i32_sat_neg(y) : cvt_i32_sat(qy2);
						const bool inb0 = (unsigned)gx < wlim && (unsigned)gy < hlim;
						const bool top = lev == kTopLevel;
						// window (ix,iy) of level lev: S = 4 << lev cells wide, one every 1<<hs cells.  The whole map is
						// the one window of the top plane: with hs = 28 every in-grid cell has ix = iy = 0 and the
						// spans below come out as the map's, so nothing else treats that level specially.
						// (levels below kDenseFrom: windows every half window, the others every quarter -- frame.hpp)
						const bool sparse = lev < kDenseFrom;
						const int hs = top ? 28 : kLevelStep * lev + (sparse ? 1 : 0);
						const int back = sparse ? 1 : 3; // strides to step back when the cell index falls along the ray
						int ix = (gx >> hs) - (offx ? back : 0), iy = (gy >> hs) - (offy ? back : 0);
						ix = ix < 0 ? 0 : ix;
						iy = iy < 0 ? 0 : iy;
						const unsigned widx = ((unsigned)lev << f.mip_plane_shift) + (unsigned)index_2d(iy, f.mip_row, ix); // (= mip_index)
						diag.load_begin(f, 17);
#ifndef HMRM_WIDE_MIP
#define HMRM_WIDE_MIP 1
#endif
						// (the element index fits 32 bits -- 8 planes of at most 2^28 floats, api.cpp's map limit -- the byte offset
						// need not: a 64-bit offset, one v_lshl_add_u64 where the 32-bit form had a shift.  Measured equal within the
						// run-to-run spread on C3 / C5 / C2 / C4, profiles/r05_raw/wide_mip_abn.txt; with HMRM_WIDE_MIP 0 -- round 4 --
						// very oblong maps near the 2^29-cell limit, 16385 x 32766, had to be rendered by the literal loop, 85 x slower.)
						float mf = HMRM_WIDE_MIP ? mip[(size_t)((REC ? inb0 && top : inb0) ? widx : 0u)]
						                         : *(const float *)((const char *)mip + (size_t)((inb0 ? widx : 0u) * 4u));
						uint32_t rec_xs0 = ~0u, rec_xs1 = ~0u, rec_ys0 = ~0u, rec_ys1 = ~0u;
						if constexpr (REC) { // the window's record: two 16-byte loads of one 32-byte line
							const WindowRecord *recs = reinterpret_cast<const WindowRecord *>(f.mipbuf_bil);
							const unsigned ridx = (inb0 && !top) ? (unsigned)index_2d(iy, rec_row(f.map_w), ix) : 0u;
							const uint4 r0 = *reinterpret_cast<const uint4 *>(recs + ridx);
							const uint2 r1 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(recs + ridx) + 16);
							mf = top ? mf : __uint_as_float(r0.x);
							rec_xs0 = r0.z; rec_xs1 = r0.w; rec_ys0 = r1.x; rec_ys1 = r1.y;
						}
						diag.load_end(f, 17, mf);
						if (kEarlyLoad) refresh_stale();
						const bool exact = kStepsLeft ? (ax.left | ay.left | az.left) >= 0
						                              : ax.key != 0xffffffffu && ay.key != 0xffffffffu && az.key != 0xffffffffu;
						const int left_min = min(ax.left, min(ay.left, az.left)); // (kStepsLeft)
						const int wx0 = ix << hs, wy0 = iy << hs;
						// (the last windows of a row / column hang over the map's edge: the usable span ends at the edge)
						const int wcells = top ? (1 << 30) : (4 << (kLevelStep * lev)); // window size S in cells
						const int wspan_x = min(wcells, f.map_w - wx0), wspan_y = min(wcells, f.map_h - wy0);
						// estimates of the steps left before each lateral constraint bites; rdel is signed like
						// the motion, so every quotient is >= 0.  Only estimates: verified below.
						// (a coordinate that does not move -- s == 0 or absorbed -- has rdel = 2^40 and is strictly inside
						// of whichever edge it looks at: a huge quotient once its sign is dropped, which costs nothing -- an
						// operand modifier -- and changes nothing for a moving coordinate.  No case distinction needed.)
						const double ex = (double)(offx ? wx0 : wx0 + wspan_x) * gwid;  // x edge ahead
						const double ey = -(double)(offy ? wy0 : wy0 + wspan_y) * gwid; // y edge ahead
						double room_lat = 0.0;
						if (kStepsLeft) {
							room_lat = __builtin_fmin(__builtin_fabs((ex - x) * ax.rdel), __builtin_fabs((ey - y) * ay.rdel));
							// (computed here, not sunk behind the wait for the load: the empty statement reads the estimate
							// and stands between the load and the first use of its result)
							if (kEarlyLoad) asm volatile("" : "+v"(mf) : "v"(room_lat));
						}
						const double m = (double)mf; // (floats rounded up: also bounds every float / interpolated threshold)
						const bool above = z >= m;
						// Nothing below can succeed unless the ray is above this window's maximum: when no
						// lane of the wave is, skip the estimate and the verification (the usual case in
						// the last, nearly empty waves of a launch, which set its duration).
						double room = 0.0, room_b = 0x1p40, room_z = 0.0;
						bool z_bound = false, ok = false, can = false, binade_bound = false;
						int n = 0;
						const bool cand = inb0 && exact && above;
						if (__builtin_amdgcn_ballot_w64(cand) != 0ull) {
							if (kStepsLeft) {
								room = room_lat;
							} else {
								room = (ax.lim - x) * ax.rdel;
								room = __builtin_fmin(room, (ay.lim - y) * ay.rdel);
								room = __builtin_fmin(room, (az.lim - z) * az.rdel);
								room_b = room; // steps left inside the three binades
								room = __builtin_fmin(room, sx != 0.0 ? (ex - x) * ax.rdel : 0x1p40);
								room = __builtin_fmin(room, sy != 0.0 ? (ey - y) * ay.rdel : 0x1p40);
							}
							room_z = sz < 0.0 ? (m - z) * az.rdel : 0x1p40;
							z_bound = room_z < room;
							room = __builtin_fmin(room, room_z);
							// (the saturating cast takes care of huge and negative estimates; the step budget caps the
							// integer: a jump never takes more steps than the cap has left)
							n = min(cvt_i32_sat(room * 0.998), budget) - 1;
							if (kStepsLeft) { // (the binades' share is an exact count, not an estimate)
								// (kCross: the jump's last step is a real one and may leave the binade)
								const int left_lim = left_min + (kCross ? 1 : 0);
								binade_bound = left_lim <= n;
								z_bound = z_bound & !binade_bound;
								n = min(n, left_lim);
							} else {
								binade_bound = room_b <= room;
							}
							can = cand && n >= kMinLeap;
							// landing point and its exact verification
							// (kCross: n - 1 steps by multiplication, all inside the three binades by count, then one real step
							// from that exact position.  Coordinates move monotonically, so the tests of the end point below
							// hold for every position before it.)
							const double nn = (double)(kCross ? n - 1 : n);
							const double xm = x + nn * ax.delta, ym = y + nn * ay.delta, zm = z + nn * az.delta;
							const double xn = kCross ? xm + sx : xm, yn = kCross ? ym + sy : ym, zn = kCross ? zm + sz : zm;
							bool nearn = false;
							const double qxn = cell_coord_fast<GWM>(xn, f, nearn), qyn = cell_coord_fast<GWM>(-yn, f, nearn);
							const int gxn = cvt_i32_sat(qxn), gyn = GWM == 0 ? cvt_i32_sat_neg(yn) : cvt_i32_sat(qyn);
							const bool inbn = (unsigned)gxn < wlim && (unsigned)gyn < hlim; // (diagnostics only)
							// (inside the window implies inside the grid: the spans were cut at the map's edge)
							// (kStepsLeft: n <= left of every axis, so the landing point is inside the three binades by count)
							// (general grid widths, nearn: q' + 2^-20 lies in [k, k + 2^-19), so the landing cell is k = gxn or k - 1;
							// both are asked to be inside the window, on both axes -- one flag serves the two)
							const unsigned nm = (GWM == 2 && nearn) ? 1u : 0u;
							ok = can && (unsigned)(gxn - wx0) - nm < (unsigned)wspan_x - nm &&
							     (unsigned)(gyn - wy0) - nm < (unsigned)wspan_y - nm && zn >= m &&
							     (kStepsLeft || (axis_landing_ok(ax, xn) && axis_landing_ok(ay, yn) && axis_landing_ok(az, zn)));
							diag.on_landing_refused(f, can && !ok, inbn,
							                        (unsigned)(gxn - wx0) < (unsigned)wspan_x && (unsigned)(gyn - wy0) < (unsigned)wspan_y, zn >= m);
							if constexpr (REC) {
								// The positions leaped over are x + k delta, k = 0 .. n - 1 (kCross: up to (xm, ym); else up to the landing
								// point): all ON the segment between the first and the last, inside the window, at heights >= max2.  None
								// of them may lie in a recorded cell: segment against each cell's box, in cells relative to the window's
								// corner, float -- values below 32, conversion and product errors below 2^-13 -- with the box grown by
								// 2^-10 and the line test given 2^-9 of slack: conservative, never wrong.  (Separating axes: the
								// segment's extent in x, in y, and the line through it against the box's four corners.)
								const double bxd = kCross ? xm : xn, byd = kCross ? ym : yn;
								const double qbx = GWM == 0 ? bxd : bxd * f.inv_grid_width, qby = GWM == 0 ? -byd : -byd * f.inv_grid_width;
								const float pax = (float)(qx2 - (double)wx0), pay = (float)(qy2 - (double)wy0);
								const float pbx = (float)(qbx - (double)wx0), pby = (float)(qby - (double)wy0);
								constexpr float grow = 0x1p-10f;
								const float ddx = pbx - pax, ddy = pby - pay;
								const float x_lo = __builtin_fminf(pax, pbx) - (1.0f + grow), x_hi = __builtin_fmaxf(pax, pbx) + grow;
								const float y_lo = __builtin_fminf(pay, pby) - (1.0f + grow), y_hi = __builtin_fmaxf(pay, pby) + grow;
								const float reach = (0.5f + grow) * (__builtin_fabsf(ddx) + __builtin_fabsf(ddy)) + 0x1p-9f;
								const float cax = pax - 0.5f, cay = pay - 0.5f; // (cell corner - this = cell centre - start)
								bool touched = false;
#pragma unroll
								for (int k = 0; k < kRecCells; ++k) {
									const uint32_t wxs = k < 4 ? rec_xs0 : rec_xs1, wys = k < 4 ? rec_ys0 : rec_ys1;
									const float cx = (float)((wxs >> (8 * (k & 3))) & 0xffu), cy = (float)((wys >> (8 * (k & 3))) & 0xffu);
									const float ex = cx - cax, ey = cy - cay;
									const float cross = ex * ddy - ey * ddx;
									const bool apart = cx > x_hi || cx < x_lo || cy > y_hi || cy < y_lo || __builtin_fabsf(cross) > reach;
									touched = touched || !apart;
								}
								// (the reference truncates: a coordinate in (-1, 0) -- a ray on its way out through the map's low edge --
								// still names cell 0, which no box says.  Such paths are marched.)
								const bool floor_is_trunc = __builtin_fmin(qx2, qbx) >= 0.0 && __builtin_fmin(qy2, qby) >= 0.0;
								ok = ok && (top || (!touched && floor_is_trunc));
							}
							x = ok ? xn : x;
							y = ok ? yn : y;
							z = ok ? zn : z;
							budget -= ok ? n : 0;
							if (kStepsLeft) {
								const int took = ok ? n : 0;
								ax.left -= took;
								ay.left -= took;
								az.left -= took;
							}
						}
						diag.on_attempt_done(f, inb0, exact, above, n < kMinLeap, z_bound, can, ok, n, lev);
						diag.on_bounds(ok, binade_bound);
						// level policy (performance only; any policy gives the same pixels):
						//   window crossed                    -> coarser next time, if the height bound of this
						//                                        level left room for a window kUpRatio times
						//                                        longer (a coarser maximum is no lower)
						//   jump ended at a binade boundary   -> same level (without kCross: and march a group first)
						//   height bound was the limit        -> finer; without a jump retry at once (the
						//     (z < max, or z-room smallest)      level strictly decreases); at the finest
						//                                        level march 1 + finest_pause groups first
						//   no lateral/binade room, not exact -> coarser (a bigger window has more room),
						//                                        growing pause while attempts keep failing
						const bool height_limited = inb0 && exact && (!above || z_bound);
						// levels per move: two while the ray is young (kAdaptive), then one
						const bool young = kAdaptive && jumps <= kAdaptAfter;
						const int lstep = kAdaptive ? (young ? 2 : 1) : 1;
						if (kAdaptive) jumps += ok ? 1 : 0;
						const int coarser = lev + lstep > kMipLevels - 1 ? kMipLevels - 1 : lev + lstep;
						const int minlev = f.min_level;
#ifndef HMRM_DOWN
#define HMRM_DOWN 1
#endif
						// a failed height test drops HMRM_DOWN levels, a height-limited jump one
						const int drop = (ok ? 1 : HMRM_DOWN) * lstep;
						const int finer = top ? kMipLevels - 1 : (lev - drop > minlev ? lev - drop : minlev);
						const bool at_finest = lev == minlev;
						// (selects, not branches: the three cases are mutually exclusive)
						const bool crossed = ok && !z_bound;
						const bool hl = !crossed && height_limited;
						const bool other = !crossed && !hl;
						// (lane-mask logic: `a ? b : c` on booleans would be materialised in VGPRs)
						// (a window lstep levels up is 2^lstep times as long: the scaling rides on the exponent)
						const double room_up = kAdaptive ? __builtin_ldexp(room, lstep) : kUpRatio * room;
						const bool go_up = (crossed & (room_z >= room_up) & !binade_bound) | other;
						const int fails_before = fails;
						lev = hl ? finer : (go_up ? coarser : lev);
						fails = (crossed | (hl & ok)) ? 0 : fails + (other ? 1 : 0);
						cooldown = (hl & !ok & at_finest) ? f.finest_pause : (other ? (fails_before < 3 ? fails_before : 3) : 0);
						// retry one level down without marching; after a jump look at the next window straight
						// away -- unless the jump stopped at a binade boundary: only real steps cross it,
						// another attempt here would just fail
						// (kCross: the jump's last step has crossed it)
						skip_group = (hl & !ok & !at_finest) | (ok & (kCross | !binade_bound));
						if constexpr (REC) {
							// two levels only: the whole map while the ray is above everything, then the record level for good.
							// A refusal there is followed by a group; refusals in a row thin the attempts out (see `attempt`).
							lev = lev == kTopLevel ? kTopLevel : kRecLevel;
							fails = (ok | top) ? 0 : (fails_before < kRecBackoff ? fails_before + 1 : kRecBackoff);
							skip_group = (top & hl & !ok) | (ok & (kCross | !binade_bound));
						}
					}
				}
				diag.on_trip(f, LEAP, skip_group);
				if (skip_group) continue;

				// --------------------------------------------- speculative group
				diag.on_group();
				double X[U], Y[U], Z[U], T[U];
				unsigned cell[U]; // (unsigned: the 64-bit address needs no sign extension)
				bool inb[U];
				X[0] = x; Y[0] = y; Z[0] = z;
#pragma unroll
				for (int j = 1; j < U; ++j) {
					X[j] = X[j - 1] + sx;
					Y[j] = Y[j - 1] + sy;
					Z[j] = Z[j - 1] + sz;
				}
				// cells of the U positions.  Only the integers are kept: the general-grid-width quotient q' is needed for
				// nothing but its truncation (and the test whether it is too close to an integer to be trusted), and holding
				// U pairs of them cost the general instantiations 12 vector registers (76: 6 waves per SIMD).  The
				// bilinear mode needs the exact quotients themselves (its weights) and keeps them.
				double QX[BILINEAR ? U : 1], QY[BILINEAR ? U : 1];
				bool near = false;
#pragma unroll
				for (int j = 0; j < U; ++j) {
					int gx, gy;
					if constexpr (BILINEAR && GWM != 0) { // (the exact q: true division unless the reciprocal is exact)
						QX[j] = GWM == 2 ? X[j] / f.grid_width : X[j] * f.inv_grid_width;
						QY[j] = GWM == 2 ? -Y[j] / f.grid_width : -Y[j] * f.inv_grid_width;
						gx = cvt_i32_sat(QX[j]);
						gy = cvt_i32_sat(QY[j]);
					} else {
						const double qx = cell_coord_fast<GWM>(X[j], f, near), qy = cell_coord_fast<GWM>(-Y[j], f, near);
						if constexpr (BILINEAR) { QX[j] = qx; QY[j] = qy; }
						gx = cvt_i32_sat(qx);                                       // hmap.cpp:1001-1004
						gy = GWM == 0 ? cvt_i32_sat_neg(Y[j]) : cvt_i32_sat(qy);
					}
					inb[j] = (unsigned)gx < wlim && (unsigned)gy < hlim;      // hmap.cpp:1006-1011
					cell[j] = inb[j] ? (unsigned)index_2d(gy, f.map_w, gx) : 0u;
				}
				if (GWM == 2 && !BILINEAR && near) { // some position is on a cell boundary to within 2^-20: divide for real
#pragma unroll
					for (int j = 0; j < U; ++j) {
						// (one division at a time: interleaved, their temporaries set the kernel's register count)
						__builtin_amdgcn_sched_barrier(0);
						const int gx = cvt_i32_sat(X[j] / f.grid_width);
						__builtin_amdgcn_sched_barrier(0);
						const int gy = cvt_i32_sat(-Y[j] / f.grid_width);
						inb[j] = (unsigned)gx < wlim && (unsigned)gy < hlim;
						cell[j] = inb[j] ? (unsigned)index_2d(gy, f.map_w, gx) : 0u;
					}
					__builtin_amdgcn_sched_barrier(0);
				}
				diag.load_begin(f, 18);
				if constexpr (BILINEAR) {
#pragma unroll
					for (int j = 0; j < U; ++j) {
						const Bil b = bil_setup(inb[j] ? QX[j] : 0.0, inb[j] ? QY[j] : 0.0, f.map_w, f.map_h);
						T[j] = bil_mix(b, thr[b.c00], thr[b.c10], thr[b.c01], thr[b.c11]);
					}
				} else {
#pragma unroll
					// (32-bit byte offsets from the table's base -- api.cpp caps maps at 2^29 cells -- so that the loads can
					// take the base from scalar registers: no 64-bit address arithmetic per sample)
					for (int j = 0; j < U; ++j)
						T[j] = F32 ? (double)*(const float *)((const char *)thr32 + (size_t)(cell[j] * 4u))
						           : *(const double *)((const char *)thr + (size_t)(cell[j] * 8u)); // hmap.cpp:1013-1014 (+ c0.z)
				}
				if constexpr (U == 4) diag.load_end(f, 18, T[0], T[1], T[2], T[3]);
				if (budget >= U) {
					// in order: the first position that leaves the grid (:1006) or hits (:1016) ends the ray
					int first = U, hit_j = 0;
					unsigned hit_cell = 0u;
					bool hit = false;
#pragma unroll
					for (int j = U - 1; j >= 0; --j) { // (selects, last write = earliest position)
						const bool h = inb[j] && Z[j] < T[j];
						const bool s = !inb[j] || h;
						first = s ? j : first;
						hit = s ? h : hit;
						hit_cell = s ? cell[j] : hit_cell;
						hit_j = s ? j : hit_j;
					}
					const int taken = first + (hit ? 1 : 0); // loads the reference executed in this group
					budget -= taken;
					done = first < U;
					if (hit) {
						if constexpr (BILINEAR) {
							// (the weights are rebuilt for the one position that hit: cheaper than keeping
							// U sets of them alive)
							double qxh = QX[0], qyh = QY[0];
#pragma unroll
							for (int j = 1; j < U; ++j) {
								qxh = hit_j == j ? QX[j] : qxh;
								qyh = hit_j == j ? QY[j] : qyh;
							}
							rgba = shade_hit_bilinear(f, cmap, (int)hit_cell, bil_setup(qxh, qyh, f.map_w, f.map_h));
						} else {
							rgba = shade_hit(f, cmap[hit_cell]);
						}
						real_hit = true;
					}
				} else {
					// (almost never) close to the step cap: the literal loop, one position at a time, cap
					// checked per step; a real loop over scalars so that none of the group's arrays is
					// indexed dynamically
					double xs = x, ys = y, zs = z;
#pragma unroll 1
					for (int j = 0; j < U; ++j) {
						const double qx = (GWM == 0) ? xs : xs / f.grid_width, qy = (GWM == 0) ? -ys : -ys / f.grid_width;
						const int gx = cvt_i32_sat(qx), gy = cvt_i32_sat(qy);
						if (!((unsigned)gx < wlim && (unsigned)gy < hlim)) { done = true; break; }
						if (budget <= 0) { my_cap = 1; done = true; break; }
						--budget;
						const int c = gy * f.map_w + gx;
						Bil b{};
						double t;
						if (BILINEAR) {
							b = bil_setup(qx, qy, f.map_w, f.map_h);
							t = bil_mix(b, thr[b.c00], thr[b.c10], thr[b.c01], thr[b.c11]);
						} else {
							t = F32 ? (double)thr32[c] : thr[c];
						}
						if (zs < t) { // hmap.cpp:1016
							rgba = BILINEAR ? shade_hit_bilinear(f, cmap, c, b) : shade_hit(f, cmap[c]);
							real_hit = true;
							done = true;
							break;
						}
						xs += sx;
						ys += sy;
						zs += sz;
					}
				}
				x = X[U - 1] + sx;
				y = Y[U - 1] + sy;
				z = Z[U - 1] + sz;
				if (LEAP && kStepsLeft) { // U real steps further inside (or out of) the binades
					ax.left -= U;
					ay.left -= U;
					az.left -= U;
				}
			}
			if (STATS) my_steps = (unsigned long long)(unsigned)(budget0 - budget);
		}

		if (real_hit) my_hit = 1;
		else rgba = shade_miss(f, ray.dz);
		// (row and pitch are below 2^31, api.cpp: one 32 x 32 -> 64-bit multiply-add)
		out[(uint64_t)(uint32_t)pid.lrow * (uint32_t)out_stride_px + (uint32_t)pid.px] = rgba;
		if (STATS && st.steps_per_pixel)
			st.steps_per_pixel[(int64_t)pid.py * f.screen_w + pid.px] = diag.pixel_value(f, my_steps);
	}
	publish_counters<STATS>(st, my_steps, my_hit, my_cap);
	diag.publish(st, f);
	return pid.tile_y;
}

template <int PROJ, bool STATS, int GWM, int LEAP, int SAMP>
__global__ __launch_bounds__(kBlockThreads, HMRM_MIN_WAVES) HMRM_OCCUPANCY_ATTR void k_render_fast(const DevFrame f, const RowMap rows,
                                                     const double *__restrict__ thr,
                                                     const uint32_t *__restrict__ cmap,
                                                     uint32_t *__restrict__ out, int64_t out_stride_px,
                                                     int tiles_y, StatsOut st) {
#ifdef HMRM_TIMELINE
	const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();
#endif
	// calibration launches only (RowMap::measure): when did this wave start
	unsigned long long wave_t0 = 0;
	if (!STATS && rows.measure) wave_t0 = __builtin_amdgcn_s_memrealtime();
	const int tile_y = render_wave_tile<PROJ, STATS, GWM, LEAP, SAMP>(f, rows, thr, cmap, out, out_stride_px, tiles_y, st, (int)blockIdx.x,
	                                                                  blockIdx.z * 32768u + blockIdx.y, (int)(threadIdx.x >> 6),
	                                                                  (int)(threadIdx.x & 63));
	if (!STATS && rows.measure && tile_y >= 0 && (threadIdx.x & 63) == 0) {
		// record of a tile row: [0] start of its first workgroup (rows are handed out left to right), [1 + k] longest
		// wave among the tile columns = k mod 32 (32 addresses per row: the atomics of a row's 2 x 480 waves spread out)
		const unsigned long long took = __builtin_amdgcn_s_memrealtime() - wave_t0;
		unsigned long long *rec = rows.measure + (size_t)tile_y * kMeasureStride;
		if (blockIdx.x == 0 && threadIdx.x == 0) rec[0] = wave_t0;
		atomicMax(&rec[1 + (blockIdx.x & 31u)], took);
	}
#ifdef HMRM_TIMELINE
	if (!STATS && g_timeline && (threadIdx.x & 63) == 0) {
		unsigned xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		const size_t wave = ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (kBlockThreads / 64) + (threadIdx.x >> 6);
		g_timeline[wave] = TimelineRec{tl_t0, (unsigned long long)__builtin_amdgcn_s_memrealtime(), xcc, 0u};
	}
#endif
}

// ---------------------------------------------------------------- pyramid ----
__host__ __device__ __forceinline__ float round_up_to_float(double v) {
	float r = (float)v;
	if ((double)r < v) { // conversion rounded down (v finite): next float towards +inf
		uint32_t b;
		__builtin_memcpy(&b, &r, sizeof b);
		if (r == 0.0f) b = 1u;                 // smallest positive subnormal
		else if (b & 0x80000000u) b -= 1u;     // negative: magnitude shrinks
		else b += 1u;                          // positive: magnitude grows (max float -> +inf)
		__builtin_memcpy(&r, &b, sizeof r);
	}
	return r;
}
float round_up_to_float_host(double v) { return round_up_to_float(v); }

// Level 0: window (ix,iy) = max of thr over the 4 x 4 cells from (S0 ix, S0 iy), clipped; NaN ignored.
__global__ __launch_bounds__(256) void k_build_mip0(const double *__restrict__ thr, int map_w, int map_h,
                                                    float *__restrict__ dst, int dst_w, int dst_h, int pitch) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (int64_t)dst_w * dst_h) return;
	const int ix = (int)(i % dst_w), iy = (int)(i / dst_w);
	double m = -__builtin_huge_val();
	constexpr int S0 = 1 << mip_stride_shift(0); // level 0: 4-cell windows every S0 cells
	for (int yy = S0 * iy; yy < S0 * iy + 4 && yy < map_h; ++yy)
		for (int xx = S0 * ix; xx < S0 * ix + 4 && xx < map_w; ++xx) {
			const double v = thr[(int64_t)yy * map_w + xx];
			if (v > m) m = v;
		}
	dst[mip_index(ix, iy, pitch)] = round_up_to_float(m);
}

// Level l+1 from level l.  A level-l window is src_strides strides of level l wide; the window F = 2^kLevelStep times
// as large that starts at stride i of level l+1 (= stride_ratio * i of level l) is the union of the level-l windows with
// indices stride_ratio * i + {0, src_strides, .., src_strides * (F - 1)} per axis.
__global__ __launch_bounds__(256) void k_build_mip_up(const float *__restrict__ src, int src_w, int src_h,
                                                      float *__restrict__ dst, int dst_w, int dst_h, int pitch,
                                                      int stride_ratio, int src_strides) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (int64_t)dst_w * dst_h) return;
	const int ix = (int)(i % dst_w), iy = (int)(i / dst_w);
	constexpr int F = 1 << kLevelStep;
	float m = -__builtin_huge_valf();
	for (int b = 0; b < F; ++b) {
		const int yy = stride_ratio * iy + src_strides * b;
		if (yy >= src_h) break;
		for (int a = 0; a < F; ++a) {
			const int xx = stride_ratio * ix + src_strides * a;
			if (xx >= src_w) break;
			const float v = src[mip_index(xx, yy, pitch)];
			if (v > m) m = v;
		}
	}
	dst[mip_index(ix, iy, pitch)] = m;
}

// 3x3 maximum filter of the thr table (edges clamped, NaN ignored) plus a rounding margin:
// every bilinear interpolation that a position inside cell c can see mixes four cells of c's
// 3x3 neighbourhood, so in exact arithmetic it is at most their maximum m; the six roundings
// of bil_mix can push it above m by a few ulp of the largest magnitude A involved (differences
// reach 2A), hence the bound m + A * 2^-45.  A pyramid built from these values bounds the
// interpolated thresholds of all positions whose cell lies in a window.
__global__ __launch_bounds__(256) void k_dilate3x3(const double *__restrict__ thr, int w, int h,
                                                   double *__restrict__ dst) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (int64_t)w * h) return;
	const int x = (int)(i % w), y = (int)(i / w);
	double m = -__builtin_huge_val(), a = 0.0;
	for (int yy = max(y - 1, 0); yy <= min(y + 1, h - 1); ++yy)
		for (int xx = max(x - 1, 0); xx <= min(x + 1, w - 1); ++xx) {
			const double v = thr[(int64_t)yy * w + xx];
			if (v > m) m = v;
			if (__builtin_fabs(v) > a) a = __builtin_fabs(v);
		}
	dst[i] = m + a * 0x1p-45; // (inf stays inf; all-NaN neighbourhoods give -inf: never a hit there)
}

hipError_t launch_dilate3x3(const double *d_thr, int w, int h, double *d_dst, hipStream_t stream) {
	const int64_t n = (int64_t)w * h;
	hipLaunchKernelGGL(k_dilate3x3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_thr, w, h, d_dst);
	return hipGetLastError();
}

hipError_t launch_build_mip0(const double *d_thr, int map_w, int map_h, float *d_dst, int dst_w, int dst_h,
                             int pitch, hipStream_t stream) {
	const int64_t n = (int64_t)dst_w * dst_h;
	hipLaunchKernelGGL(k_build_mip0, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_thr, map_w,
	                   map_h, d_dst, dst_w, dst_h, pitch);
	return hipGetLastError();
}

hipError_t launch_build_mip_up(const float *d_src, int src_w, int src_h, float *d_dst, int dst_w, int dst_h,
                               int pitch, int src_level, hipStream_t stream) {
	const int64_t n = (int64_t)dst_w * dst_h;
	const int stride_ratio = 1 << (mip_stride_shift(src_level + 1) - mip_stride_shift(src_level));
	hipLaunchKernelGGL(k_build_mip_up, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_src, src_w,
	                   src_h, d_dst, dst_w, dst_h, pitch, stride_ratio, win_strides(src_level));
	return hipGetLastError();
}

// ---------------------------------------------------------------- window records ----
// One thread per window of the record level: the nine highest cells of its 16 x 16 (clipped at the map's edge, NaN ignored
// as in the pyramid), kept sorted by insertion.  The ninth is max2; the eight above it are recorded where they are
// strictly higher.
__global__ __launch_bounds__(256) void k_build_records(const double *__restrict__ thr, int map_w, int map_h,
                                                       WindowRecord *__restrict__ dst, int rw, int rh) {
	const int ix = (int)(blockIdx.x * 16u + (threadIdx.x & 15u)), iy = (int)(blockIdx.y * 16u + (threadIdx.x >> 4));
	if (ix >= rw || iy >= rh) return;
	constexpr int S = win_cells(kRecLevel), N = kRecCells + 1;
	const int wx0 = ix << mip_stride_shift(kRecLevel), wy0 = iy << mip_stride_shift(kRecLevel);
	double t[N];
	uint32_t at[N]; // row << 8 | column inside the window
#pragma unroll
	for (int j = 0; j < N; ++j) {
		t[j] = -__builtin_huge_val();
		at[j] = 0xffffu;
	}
	// A row at a time: its 16 loads are in flight together (one load per trip of a cell loop left the kernel waiting for
	// memory 256 times per window), and a row whose maximum does not reach the ninth-highest so far -- most rows -- costs
	// 16 maxima and one comparison.
	const int cols = min(S, map_w - wx0); // (>= 1)
	for (int r = 0; r < S && wy0 + r < map_h; ++r) {
		const double *row = thr + (size_t)(wy0 + r) * (size_t)map_w + wx0;
		double v[S];
#pragma unroll
		for (int c = 0; c < S; ++c) {
			const double x = row[c < cols ? c : cols - 1]; // (never past the row's end)
			v[c] = c < cols ? x : -__builtin_huge_val();
		}
		double m = v[0];
#pragma unroll
		for (int c = 1; c < S; ++c) m = __builtin_fmax(m, v[c]); // (fmax skips NaN, as the pyramid does)
		if (!(m > t[N - 1])) continue;
#pragma unroll
		for (int c = 0; c < S; ++c) {
			if (!(v[c] > t[N - 1])) continue; // (NaN too)
			t[N - 1] = v[c];
			at[N - 1] = (uint32_t)(r << 8 | c);
#pragma unroll
			for (int j = N - 1; j > 0; --j) {
				const bool up = t[j] > t[j - 1];
				const double tv = t[j];
				const uint32_t ta = at[j];
				t[j] = up ? t[j - 1] : t[j];
				at[j] = up ? at[j - 1] : at[j];
				t[j - 1] = up ? tv : t[j - 1];
				at[j - 1] = up ? ta : at[j - 1];
			}
		}
	}
	WindowRecord rec;
	rec.max2 = round_up_to_float(t[N - 1]);
	rec.spare0 = 0u;
	rec.spare1[0] = rec.spare1[1] = 0u;
	uint32_t xs[2] = {0u, 0u}, ys[2] = {0u, 0u};
#pragma unroll
	for (int j = 0; j < kRecCells; ++j) {
		const bool keep = t[j] > t[N - 1];
		xs[j >> 2] |= (keep ? (at[j] & 0xffu) : 0xffu) << (8 * (j & 3));
		ys[j >> 2] |= (keep ? (at[j] >> 8) : 0xffu) << (8 * (j & 3));
	}
	rec.xs[0] = xs[0]; rec.xs[1] = xs[1];
	rec.ys[0] = ys[0]; rec.ys[1] = ys[1];
	dst[(size_t)iy * (size_t)rw + ix] = rec;
}

hipError_t launch_build_records(const double *d_thr, int map_w, int map_h, WindowRecord *d_dst, hipStream_t stream) {
	const int rw = rec_row(map_w), rh = (map_h + 3) >> 2;
	const dim3 grid((unsigned)((rw + 15) / 16), (unsigned)((rh + 15) / 16));
	if (grid.y > 65535u) return hipErrorInvalidValue; // (api.cpp does not build records for such maps)
	hipLaunchKernelGGL(k_build_records, grid, dim3(256), 0, stream, d_thr, map_w, map_h, d_dst, rw, rh);
	return hipGetLastError();
}

// ---------------------------------------------------------------- launch ----
template <int PROJ, bool STATS, int GWM, int LEAP>
static void launch_one(const DevFrame &f, const RowMap &rows, const double *d_thr, const uint32_t *d_cmap,
                       uint32_t *d_out, int64_t out_stride_px, StatsOut st, dim3 grid, int tiles_y,
                       hipStream_t stream) {
	if constexpr (LEAP == kRecords) { // (nearest sampling only: launch_render_fast has checked)
		hipLaunchKernelGGL((k_render_fast<PROJ, STATS, GWM, LEAP, 0>), grid, dim3(kBlockThreads), 0, stream, f, rows,
		                   d_thr, d_cmap, d_out, out_stride_px, tiles_y, st);
	} else {
		if (f.sampling == 1)
			hipLaunchKernelGGL((k_render_fast<PROJ, STATS, GWM, LEAP, 1>), grid, dim3(kBlockThreads), 0, stream, f, rows,
			                   d_thr, d_cmap, d_out, out_stride_px, tiles_y, st);
		else if (f.sampling == 2) // (d_thr is the float table here, see launch_render_fast)
			hipLaunchKernelGGL((k_render_fast<PROJ, STATS, GWM, LEAP, 2>), grid, dim3(kBlockThreads), 0, stream, f, rows,
			                   d_thr, d_cmap, d_out, out_stride_px, tiles_y, st);
		else
			hipLaunchKernelGGL((k_render_fast<PROJ, STATS, GWM, LEAP, 0>), grid, dim3(kBlockThreads), 0, stream, f, rows,
			                   d_thr, d_cmap, d_out, out_stride_px, tiles_y, st);
	}
}

template <int PROJ, bool STATS, int GWM>
static void launch_leap(FastKernel kernel, const DevFrame &f, const RowMap &rows, const double *d_thr,
                        const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px, StatsOut st, dim3 grid,
                        int tiles_y, hipStream_t stream) {
	if (kernel == kLeaps) launch_one<PROJ, STATS, GWM, kLeaps>(f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream);
	else if (kernel == kRecords) launch_one<PROJ, STATS, GWM, kRecords>(f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream);
	else launch_one<PROJ, STATS, GWM, kPlainGroups>(f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream);
}

template <int PROJ, bool STATS>
static void launch_gwm(FastKernel leap, const DevFrame &f, const RowMap &rows, const double *d_thr,
                       const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px, StatsOut st, dim3 grid,
                       int tiles_y, hipStream_t stream) {
	switch (f.grid_mode) {
	case 0: launch_leap<PROJ, STATS, 0>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream); break;
	case 1: launch_leap<PROJ, STATS, 1>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream); break;
	default: launch_leap<PROJ, STATS, 2>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream); break;
	}
}

template <bool STATS>
static void launch_proj(FastKernel leap, const DevFrame &f, const RowMap &rows, const double *d_thr,
                        const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px, StatsOut st, dim3 grid,
                        int tiles_y, hipStream_t stream) {
	switch (f.projection) {
	case 1: launch_gwm<1, STATS>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream); break;
	case 2: launch_gwm<2, STATS>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream); break;
	default: launch_gwm<3, STATS>(leap, f, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream); break;
	}
}

// Calibration records (RowMap::measure): kMeasureStride words per tile row, see k_render_fast's last lines.
__global__ __launch_bounds__(256) void k_measure_init(unsigned long long *rec, int n_words) {
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i < n_words) rec[i] = 0ull;
}
// -> host_dst[2 t] = start of tile row t, host_dst[2 t + 1] = its longest wave
__global__ __launch_bounds__(256) void k_measure_readback(const unsigned long long *__restrict__ rec, unsigned long long *__restrict__ host_dst, int tile_rows) {
	const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (t >= tile_rows) return;
	const unsigned long long *r = rec + (size_t)t * kMeasureStride;
	unsigned long long m = 0;
	for (int k = 1; k < kMeasureStride; ++k) m = r[k] > m ? r[k] : m;
	host_dst[2 * t] = r[0];
	host_dst[2 * t + 1] = m;
}
hipError_t launch_measure_init(unsigned long long *d_rec, int tile_rows, hipStream_t stream) {
	const int n = tile_rows * kMeasureStride;
	hipLaunchKernelGGL(k_measure_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_rec, n);
	return hipGetLastError();
}
hipError_t launch_measure_readback(const unsigned long long *d_rec, unsigned long long *h_pinned_dev, int tile_rows, hipStream_t stream) {
	hipLaunchKernelGGL(k_measure_readback, dim3((unsigned)((tile_rows + 255) / 256)), dim3(256), 0, stream, d_rec, h_pinned_dev, tile_rows);
	return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_thr_to_float(const double *__restrict__ thr, float *__restrict__ dst, int64_t n) {
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)thr[i];
}

hipError_t launch_thr_to_float(const double *d_thr, float *d_thr32, int64_t n, hipStream_t stream) {
	int64_t blocks = (n + 255) / 256;
	if (blocks > 256 * 16) blocks = 256 * 16;
	hipLaunchKernelGGL(k_thr_to_float, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, stream, d_thr, d_thr32, n);
	return hipGetLastError();
}

hipError_t launch_render_fast(const DevFrame &f, const RowMap &rows, const double *d_thr_f64, const float *d_thr32,
                              const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px,
                              unsigned long long *d_counters, uint32_t *d_steps, double *d_entry, bool stats,
                              FastKernel kernel, const WindowRecord *d_records, hipStream_t stream) {
	if (kernel == kRecords && (f.sampling != 0 || !d_records)) return hipErrorInvalidValue;
	DevFrame with_records;
	if (kernel == kRecords) { // (the record kernel finds its table where the bilinear mode finds its pyramid: frame.hpp)
		with_records = f;
		with_records.mipbuf_bil = reinterpret_cast<const float *>(d_records);
	}
	const DevFrame &fr = kernel == kRecords ? with_records : f;
	const double *d_thr = f.sampling == 2 ? reinterpret_cast<const double *>(d_thr32) : d_thr_f64;
	const int tiles_x = (f.screen_w + kTileW - 1) / kTileW;
	const int tiles_y = (rows.local_rows + kTileH - 1) / kTileH;
	if (tiles_x <= 0 || tiles_y <= 0) return hipSuccess;
	const dim3 grid((unsigned)tiles_x, (unsigned)(tiles_y < 32768 ? tiles_y : 32768), (unsigned)((tiles_y + 32767) / 32768));
	StatsOut st{d_counters, d_steps, d_entry};
#ifdef HMRM_TIMELINE
	{
		static TimelineRec *host = nullptr, *devbuf = nullptr;
		static size_t cap = 0, used = 0;
		static int gx = 0, gy = 0, ty = 0, segs[7] = {0, 0, 0, 0, 0, 0, 0};
		const size_t waves = (size_t)grid.x * grid.y * grid.z * (kBlockThreads / 64);
		if (!stats) {
			if (waves > cap) {
				(void)hipDeviceSynchronize();
				free(host);
				if (devbuf) (void)hipFree(devbuf);
				host = (TimelineRec *)malloc(waves * sizeof(TimelineRec));
				(void)hipMalloc((void **)&devbuf, waves * sizeof(TimelineRec));
				cap = waves;
				(void)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &devbuf, sizeof devbuf);
				static bool registered = false;
				if (!registered) {
					registered = true;
					atexit([] {
						const char *path = getenv("HMRM_TIMELINE_FILE");
						if (!path || !host || !devbuf) return;
						(void)hipDeviceSynchronize();
						if (hipMemcpy(host, devbuf, used * sizeof(TimelineRec), hipMemcpyDeviceToHost) != hipSuccess) return;
						if (FILE *fp = fopen(path, "wb")) {
							const int hdr[12] = {gx, gy, kBlockThreads / 64, (int)used, ty, segs[0], segs[1], segs[2], segs[3], segs[4], segs[5], segs[6]};
							fwrite(hdr, sizeof hdr, 1, fp);
							fwrite(host, sizeof(TimelineRec), used, fp);
							fclose(fp);
						}
					});
				}
			}
			used = waves;
			gx = (int)grid.x;
			gy = (int)(grid.y * grid.z);
			for (int k = 0; k < 3; ++k) segs[k] = rows.seg_first[k];
			for (int k = 0; k < 4; ++k) segs[3 + k] = rows.seg_delta[k];
			ty = tiles_y;
		}
	}
#endif
	if (stats) launch_proj<true>(kernel, fr, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream);
	else launch_proj<false>(kernel, fr, rows, d_thr, d_cmap, d_out, out_stride_px, st, grid, tiles_y, stream);
	return hipGetLastError();
}

void render_tile_shape(int *tile_w, int *tile_h) {
	*tile_w = kTileW;
	*tile_h = kTileH;
}

} // namespace hmrm
