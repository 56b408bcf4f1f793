// frame.hpp -- the per-frame record the host hands to the render kernel.
//
// The reference rebuilds an ImagePlane object and the two box corners every
// frame on the host (main/hmap.cpp:661-672, :952-974) and the pixel loop reads
// them.  Here the same quantities are computed once per frame on the host (with
// glibc's sin/cos/tan, so the bits match the reference's libm calls) and passed
// BY VALUE as a kernel argument: they live in SGPRs / the scalar cache, no
// per-pixel memory traffic.
#pragma once
#include <stdint.h>

namespace hmrm {

struct DevFrame {
	// framebuffer (main/hmap.cpp:31-32)
	int32_t screen_w, screen_h;
	int32_t projection;          // 1 perspective, 2 spherical, 3 orthographic (:104-106)
	// heightmap / colormap dimensions (:54-55,:60-61; equal by :503-515)
	int32_t map_w, map_h;
	uint8_t bg[4];               // bg_r,bg_g,bg_b,255 (:110-112)
	// Ray origin for perspective/spherical = cam_pos (Perspective.cpp:26, Spherical.cpp:21)
	double cam[3];
	// Perspective.cpp:16-22 / Orthographic.cpp:13-16
	double upper_left[3], plane_right[3], plane_down[3];
	// Orthographic.cpp:5 (float-rounded look), used as ray.dir
	double look[3];
	// Spherical.cpp:23-25 is separable: sin/cos(va) depend on the row only,
	// sin/cos(ha) on the column only.  Host-built tables (glibc), device memory.
	const double *col_cos_ha, *col_sin_ha;   // screen_w entries
	const double *row_sin_va, *row_cos_va;   // screen_h entries
	// box corners (hmap.cpp:968-974)
	double c0[3], c1[3];
	double grid_width;           // :65
	double nudge;                // grid_width * 0.01 (:998)
	double step_dist;            // :68
	// ---- derived, bit-preserving helpers (not in the reference) ----
	double inv_grid_width;       // fl(1/grid_width); exact iff grid_pow2 != 0
	int32_t grid_pow2;           // grid_width is a normal power of two: x/gw == x*(1/gw) exactly
	int32_t grid_mode;           // 0: grid_width == 1.0, 1: power of two, 2: general
	double thr_max;              // max over cells of heightmap_buf[i] + c0.z (informational; the kernel uses the pyramid's top plane)
	int64_t step_cap;            // guard for the reference's unbounded while(true) (:1000)
	// window-maximum pyramid over the thr table (render_fast.hip): level l holds the maximum
	// of thr (NaN ignored: z < NaN never hits) over S x S-cell windows, S = 4, 8, 16, .. 256,
	// placed every S/2 cells; floats rounded up.  One buffer of kMipLevels + 1 planes of
	// 1 << mip_plane_shift floats: window (ix, iy) of level l is element
	// (l << mip_plane_shift) + iy * mip_row + ix -- every level uses level 0's row pitch, so the
	// kernel needs no per-level table -- and plane kMipLevels holds one element, the whole-map
	// bound (thr_max rounded up).
	const float *mipbuf;
	const float *mipbuf_bil;     // the same pyramid over the 3x3-dilated table (bilinear quality mode); the record kernel
	                             // (nearest sampling only) finds its WindowRecord table here instead -- see below
	int32_t mip_row;             // row pitch of every plane (windows per row of level 0)
	int32_t mip_plane_shift;     // log2 of the plane pitch
	// Perspective / spherical: every ray starts at cam, so on which side of the origin the box lies per axis
	// is a property of the frame (device_common.hpp slab_points_away): box_side[i] = high word of
	// c0[i] - cam[i] when c0[i] - cam[i] and c1[i] - cam[i] are finite, non-zero, of moderate exponent and
	// of one sign (its sign bit is the side), else 0 with box_side_known[i] = 0.
	// (Kept inside the struct's old size on purpose.  The struct is the kernel argument; a field appended at its
	// end that the miss shade reads -- the background as doubles -- made EVERY wave wait for one more 64-byte
	// line of it at start-up, the compiler hoists scalar loads to the entry: +3...5 % on the all-terrain frames.)
	uint32_t box_side[3];
	int32_t box_side_known[3];
	int32_t diag_mode;           // tools only: what the instrumented kernel writes per pixel
	int32_t sampling;            // 0 nearest cell (the reference), 1 bilinear quality mode, 2 nearest cell with float thresholds
	int32_t min_level;           // finest pyramid level worth an attempt (api.cpp, from min_window)
	int32_t min_window;          // ... as a window size in cells (camera.cpp)
	int32_t finest_pause;        // extra groups marched after a refused attempt at that level (camera.cpp)
	int32_t pad4_;
};

// Window sizes S = 4 * 2^(kLevelStep*l) cells, placed every S/2 cells.  kLevelStep 1 (the build): S = 4, 8, 16, 32,
// 64, 128, 256 -- a ray moves two levels at a time (4, 16, 64, 256) until it has made a few jumps and then one at a time
// (render_fast.hip kAdaptAfter); kLevelStep 2: only S = 4, 16, 64, 256 exist (round 2's pyramid, kept for A/B runs).
#ifndef HMRM_LEVEL_STEP
#define HMRM_LEVEL_STEP 1
#endif
constexpr int kLevelStep = HMRM_LEVEL_STEP;
#ifndef HMRM_MIP_LEVELS
#define HMRM_MIP_LEVELS (HMRM_LEVEL_STEP == 2 ? 4 : 7)
#endif
constexpr int kMipLevels = HMRM_MIP_LEVELS;
// Windows of S cells are placed every S/2 cells on the levels below kDenseFrom (a ray always finds a window with at
// least S/2 cells of room ahead) and every S/4 cells from that level on (at least 3S/4 of room: longer jumps for four
// times the entries, which the coarse levels can afford and the 4-cell level, one entry per cell then, cannot --
// C3, the one BASELINE config that uses it, got slower).  HMRM_DENSE_FROM: 0 = every level dense, 99 = none.
#ifndef HMRM_DENSE_FROM
#define HMRM_DENSE_FROM 1
#endif
constexpr int kDenseFrom = HMRM_DENSE_FROM;
#if defined(__HIPCC__)
__host__ __device__
#endif
constexpr int win_strides(int l) { return l < kDenseFrom ? 2 : 4; }                        // strides per window
#if defined(__HIPCC__)
__host__ __device__
#endif
constexpr int win_cells(int l) { return 4 << (kLevelStep * l); }                             // S
#if defined(__HIPCC__)
__host__ __device__
#endif
constexpr int mip_stride_shift(int l) { return kLevelStep * l + (l < kDenseFrom ? 1 : 0); } // log2(S / strides per window)
// Element of window (ix, iy) inside a plane (row-major with level 0's pitch; 8 x 4-window tiles per
// 128-byte line were tried and change nothing, profiles/r02_experiments.txt).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned mip_index(int ix, int iy, int pitch) { return (unsigned)(iy * pitch + ix); }

// Window records (render_fast.hip, the record kernel): one per window of level kRecLevel (16 x 16 cells, one every 4
// cells, row pitch rec_row(map_w)).  `max2` is the window's maximum with its kRecCells highest cells left out (rounded up
// to float like the pyramid, NaN ignored), xs / ys the places of those cells inside the window, one byte each (255: slot
// not used -- fewer than kRecCells cells stand above max2).  A ray at or above max2 crosses the window without a load if
// its path misses the recorded cells: maps whose windows all hold a few tall cells (needles on a plateau) defeat a flat
// maximum, not this.  Performance only: the kernel that uses the records renders the same pixels.
constexpr int kRecLevel = 2;
constexpr int kRecCells = 8;
struct WindowRecord {
	float max2;
	uint32_t spare0;
	uint32_t xs[2], ys[2];
	uint32_t spare1[2];
};
static_assert(sizeof(WindowRecord) == 32, "two 16-byte loads");
static_assert(kRecLevel >= kDenseFrom && kLevelStep == 1, "the record level's windows: 16 cells, every 4");
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int rec_row(int map_w) { return (map_w + 3) >> 2; }

// Which framebuffer rows a launch covers and where they land in the output.
struct RowMap {
	int32_t row_begin;     // contiguous mode: first global row
	int32_t local_rows;    // rows held by the output buffer
	int32_t band_rows;     // 0 = contiguous; else cyclic bands of this many rows
	int32_t band_index, band_count;
	// Launch order (api.cpp choose_tile_order): the grid's rows are handed out to tile rows in up to kOrderSegs
	// contiguous pieces.  Grid row j belongs to the last piece k with j >= seg_first[k] (seg_first[0] = 0 is
	// implicit; unused pieces have seg_first = INT32_MAX) and renders tile row (j + seg_delta[k]) mod tile rows --
	// the last piece may wrap around the end of the frame.  Together the pieces cover every tile row once.
	int32_t seg_first[3];  // pieces 1..3
	int32_t seg_delta[4];  // pieces 0..3
	// Calibration launch (api.cpp, launch order from measurement): when not null, every wave folds its duration, and
	// a tile row's first wave its start time (s_memrealtime ticks, 10 ns), into the row's record: kMeasureStride words
	// per tile row (render_fast.hip).  Scheduling aid only.
	unsigned long long *measure;
};
constexpr int kMeasureStride = 33;
constexpr int kOrderSegs = 4;

// Host: fill everything except the table pointers / thr_max / step_cap.
// Also fills the spherical tables (host arrays of screen_w / screen_h doubles) when
// projection == 2 and the pointers are not null.
struct HostCamera {
	int32_t width, height, projection;
	uint8_t bg_r, bg_g, bg_b, sampling;
	double hfov, hang, vang, pos[3], ortho_width, step_dist;
};

void build_frame(const HostCamera &cam, int32_t map_w, int32_t map_h,
                 double min_height, double max_height, double grid_width,
                 DevFrame *out,
                 double *col_cos_ha, double *col_sin_ha,   // width entries each (spherical) or null
                 double *row_sin_va, double *row_cos_va);  // height entries each (spherical) or null

// The separable halves of Spherical::GetRay (src/Spherical.cpp:18-25), entries [begin, end): cos / sin of ha per
// column (depend on hang, hfov, width), sin / cos of va per row (depend on vang, hfov, width, height).
void fill_col_tables(const HostCamera &cam, int32_t begin, int32_t end, double *col_cos_ha, double *col_sin_ha);
void fill_row_tables(const HostCamera &cam, int32_t begin, int32_t end, double *row_sin_va, double *row_cos_va);

// Scheduling aid, approximate arithmetic: for every block of `rows_per_sample` screen rows the
// longest in-box ray length (in steps) over a few sample columns of the block's middle row.
// out has ceil(screen_h / rows_per_sample) entries.  Tables as filled by build_frame.
void estimate_row_costs(const DevFrame &f, const double *col_cos_ha, const double *col_sin_ha,
                        const double *row_sin_va, const double *row_cos_va, int rows_per_sample, float *out);

} // namespace hmrm
