// api.cpp -- the C ABI declared in include/hmrm.h: scene residency in HBM,
// per-frame host set-up + kernel launch, config and image-IO entry points.
// There is no CPU rendering path in this library.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/hmrm.h"
#include "config.hpp"
#include "frame.hpp"
#include "host_pool.hpp"
#include "image_io.hpp"
#include "launch_order.hpp"
#include "render.hpp"

namespace {

thread_local std::string g_error;
thread_local double g_last_kernel_ms = -1.0;

int fail(int code, const std::string &msg) {
	g_error = msg;
	return code;
}

} // namespace

namespace hmrm {
// for the other translation units of the library (record.cpp)
int set_error(int code, const char *msg) { return fail(code, msg ? msg : ""); }
} // namespace hmrm

namespace {

#define HIP_TRY(expr)                                                                         \
	do {                                                                                      \
		hipError_t e_ = (expr);                                                               \
		if (e_ != hipSuccess)                                                                 \
			return fail(HMRM_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));    \
	} while (0)

// Environment knobs: read ONCE per scene at creation; the launch path never calls
// getenv.  hmrm_debug_reload_env() re-reads them for a scene (tests and tools switch kernels on a
// live scene).
struct Knobs {
	int64_t step_cap = (int64_t)1 << 26; // HMRM_STEP_CAP
	int kernel = 0;                      // HMRM_KERNEL: 0 leap, 1 group, 2 simple, 3 rec (window records; nearest sampling, else the plain groups)
	bool tile_order = true;              // HMRM_TILE_ORDER=0 -> row-major launch order
	int order_mode = 2;                  // HMRM_TILE_ORDER=1 -> plain rotation; 2 (default) -> rotation, then calibrated from measurement
	int diag_mode = 0;                   // HMRM_DIAG_ITERS (tools)
	int min_level = -1;                  // HMRM_MIN_LEVEL (tools)
	int finest_pause = -1;               // HMRM_FINEST_PAUSE (tools)
	bool order_verbose = false;          // HMRM_ORDER_VERBOSE=1 (tools): report every calibration on stderr
	bool try_group = true;               // HMRM_TRY_GROUP=0: the calibration does not time the plain-groups kernel against the leap kernel
	int min_plane_shift = 0;             // HMRM_DEBUG_PLANE_SHIFT (tests): log2 of the pyramid's plane pitch is at least this
	int seg_n = 0;                       // HMRM_TILE_SEGMENTS=b0:c0,b1:c1,.. (tools): tile-row pieces to start first, in this order
	int seg_b[3] = {0, 0, 0}, seg_c[3] = {0, 0, 0};
};

Knobs read_knobs() {
	Knobs k;
	// The reference loop has no cap (hmap.cpp:1000).  2^26 steps is > 2800x the longest legitimate
	// march of the largest BASELINE config (8192*sqrt(2)/0.5).  Both kernels count the budget in 32
	// bits, so the cap is at most 2^31-1.
	if (const char *s = getenv("HMRM_STEP_CAP")) {
		const long long v = atoll(s);
		if (v > 0) k.step_cap = v > 0x7fffffffLL ? 0x7fffffffLL : (int64_t)v;
	}
	if (const char *s = getenv("HMRM_KERNEL")) k.kernel = strcmp(s, "simple") == 0 ? 2 : (strcmp(s, "group") == 0 ? 1 : (strcmp(s, "rec") == 0 ? 3 : 0));
	if (const char *s = getenv("HMRM_TILE_ORDER")) {
		k.tile_order = s[0] != '0';
		if (s[0] == '1' || s[0] == '2') k.order_mode = s[0] - '0';
	}
	if (const char *s = getenv("HMRM_DIAG_ITERS")) k.diag_mode = atoi(s);
	if (const char *s = getenv("HMRM_ORDER_VERBOSE")) k.order_verbose = s[0] == '1';
	if (const char *s = getenv("HMRM_TRY_GROUP")) k.try_group = s[0] != '0';
	if (const char *s = getenv("HMRM_TILE_SEGMENTS")) {
		int b = 0, c = 0, used = 0;
		while (k.seg_n < 3 && sscanf(s, "%d:%d%n", &b, &c, &used) == 2 && b >= 0 && c > 0) {
			k.seg_b[k.seg_n] = b;
			k.seg_c[k.seg_n] = c;
			++k.seg_n;
			s += used;
			if (*s != ',') break;
			++s;
		}
	}
	if (const char *s = getenv("HMRM_DEBUG_PLANE_SHIFT")) {
		const int v = atoi(s);
		if (v > 0 && v <= 28) k.min_plane_shift = v; // (8 planes of 2^28 floats: 8 GiB, the layout of the largest legal map)
	}
	if (const char *s = getenv("HMRM_MIN_LEVEL"))
		if (s[0] >= '0' && s[0] < '0' + hmrm::kMipLevels) k.min_level = s[0] - '0';
	if (const char *s = getenv("HMRM_FINEST_PAUSE"))
		if (s[0] >= '0' && s[0] <= '9') k.finest_pause = s[0] - '0';
	return k;
}

} // namespace

using hmrm::kCostRows;
constexpr int kFrameSlots = 64;  // cached per-frame records (and spherical tables) per stream: a 64-frame orbit fits
constexpr int kMaxMeasRows = 512;  // tile rows (8192 frame rows) a launch order is calibrated for; taller frames keep the rotation
constexpr int kProbeAfterFrames = 6; // full frames of never-repeating cameras before the scene's shadow probe
constexpr int kMaxStreamCtx = 32; // streams a scene keeps launch state for (more: the least recently used one is recycled, with a stream sync)

// One cached per-frame record: the result of the host set-up (camera.cpp) for one camera, and for
// spherical cameras its sin/cos tables on the device.  A pure function of (camera, scene params,
// map size, thr_max), so a repeated camera re-renders without the libm calls or the upload.
struct FrameSlot {
	bool valid = false;
	uint64_t stamp = 0; // last use (LRU)
	hmrm_camera cam{};
	hmrm_scene_params params{};
	uint64_t thr_max_bits = 0;
	hmrm::DevFrame frame{};
	std::vector<float> row_cost; // per kCostRows screen rows: longest in-box ray, in steps (launch order hint)
	double *d_tables = nullptr;  // spherical sin/cos tables: this slot's piece of the context's arenas
	double *h_tables = nullptr;  // pinned staging, likewise
	hipEvent_t uploaded = nullptr; // after the H2D copy out of h_tables: the host may rewrite them then
	// launch order calibrated by measurement, and -- for one record per scene -- the kernel probe: launch_order.hpp
	hmrm::OrderCalibration cal;
	int meas_rows = 0;             // tile rows of the measured launch in flight
	hipEvent_t measured = nullptr; // after the read-back of a measured launch
};

// Everything a launch mutates, per HIP stream: launches on different streams of one scene never
// share a table or a counter (hmrm_render_rows_device takes the caller's stream).
struct StreamCtx {
	hipStream_t stream = nullptr;
	bool scene_owned = false; // the scene's own stream or one of its launch lanes: never recycled
	uint64_t stamp = 0;
	FrameSlot slots[kFrameSlots];
	// table storage of all slots: one device and one pinned allocation per context (a slot's first use would
	// otherwise cost a hipMalloc + hipHostMalloc, ~0.1 ms each, in the middle of a sequence of frames)
	double *d_arena = nullptr, *h_arena = nullptr;
	double *h_arena_dev = nullptr; // the pinned arena as the device sees it (the upload kernel reads it)
	// calibration records (allocated with the first measured launch): the device buffer of the ONE measured launch a
	// context has in flight at a time, and per slot kMaxMeasRows x {start, longest wave} in pinned memory
	unsigned long long *d_meas = nullptr, *h_meas = nullptr, *h_meas_dev = nullptr;
	FrameSlot *meas_owner = nullptr; // the slot whose measured launch uses d_meas (until its event is done)
	size_t arena_n = 0; // doubles per slot
	// [0] steps [1] hits [2] capped rays (cumulative, never reset) [4..7] traversal diagnostics; [8..15] the same for the
	// second launch of a shadow probe (its rays must not be counted twice)
	unsigned long long *d_counters = nullptr;
	bool probe_in_flight = false; // the scene's shadow probe uses this context's calibration records
	unsigned long long capped_seen = 0; // value of [2] the host has already reported
	// recorded behind every launch: what a recycled context waits for (the caller's stream handle may be gone by then)
	hipEvent_t last_launch = nullptr;
	bool launched = false; // ... has been recorded at least once
};

// One slot of the asynchronous read-back ring (hmrm_render_begin/_wait/_release): a device frame
// the kernel writes, a pinned host frame the copy engine fills while the next kernel runs, and the
// events that order the two streams.
struct RingFrame {
	StreamCtx *ctx = nullptr;   // launch state of the lane the frame's kernel ran on (its capped-ray counter)
	uint32_t *d_frame = nullptr;
	uint8_t *h_frame = nullptr; // pinned
	unsigned long long *h_capped = nullptr; // pinned: the stream's cumulative capped-ray counter after this frame
	size_t px = 0;
	int32_t width = 0, height = 0;
	hipEvent_t kernel_done = nullptr, copy_done = nullptr;
	bool busy = false;
};
constexpr int kMaxRing = 64;
// Launch lanes: the ticketed entry points (hmrm_render_begin, hmrm_render_device_begin) send consecutive frames to
// kLanes scene-owned streams in turn, so that one launch's tail (a few long waves) overlaps the next launch's start
// without the caller managing streams -- what bench.py's `frames_in_flight` block does by hand (C2: 0.064 -> 0.045 ms per
// frame, profiles/r03_experiments.txt section 6).
constexpr int kLanes = 3;
// A frame rendered into the caller's DEVICE memory through a lane (hmrm_render_device_begin / _wait).
struct DevTicket {
	StreamCtx *ctx = nullptr;
	hipEvent_t done = nullptr;
	unsigned long long *h_capped = nullptr; // pinned: the lane's cumulative capped-ray counter after this frame
	bool busy = false;
};

struct hmrm_scene {
	int device = 0;
	int32_t map_w = 0, map_h = 0;
	hmrm_scene_params params{};
	Knobs knobs;
	uint8_t *d_rgb = nullptr;   // W*H*3  base_heightmap_buf (hmap.cpp:51)
	uint32_t *d_cmap = nullptr; // W*H    colormap_buf as packed RGBA (hmap.cpp:59)
	double *d_thr = nullptr;    // W*H    heightmap_buf[i] + min_height
	float *d_thr32 = nullptr;   // W*H    the same rounded to float, right behind d_thr in the same allocation: the "float heights"
	                            //        mode reads it; rebuilt with every update (never lazily beside frames in flight)
	double thr_max = 0.0;
	double thr_max_bil = 0.0; // whole-map bound of the interpolated thresholds (bilinear mode)
	bool bil_valid = false;
	bool records_valid = false; // d_records describes the current heights (ensure_records)
	float *d_mipbuf_bil = nullptr; // the same over the 3x3-dilated table (bilinear quality mode)
	float *d_mipbuf = nullptr; // window maxima over d_thr: 4/8/../256-cell windows every 2/4/../128 cells
	hmrm::WindowRecord *d_records = nullptr; // frame.hpp: the 16-cell windows' maxima without their 8 highest cells, and those cells (null: map too tall for the build's grid)
	int32_t mip_w[hmrm::kMipLevels] = {}, mip_h[hmrm::kMipLevels] = {};
	int32_t mip_row = 0, mip_plane_shift = 0; // plane layout of both pyramids, see DevFrame
	size_t mip_floats() const { return (size_t)(hmrm::kMipLevels + 1) << mip_plane_shift; }
	float *plane(float *buf, int l) const { return buf + ((size_t)l << mip_plane_shift); }
	hipStream_t stream = nullptr; // the scene's own stream (hmrm_render, updates)
	hipStream_t copy_stream = nullptr; // device-to-host copies of the asynchronous ring
	hipStream_t lanes[kLanes] = {};    // launch lanes of the ticketed entry points (created on first use)
	uint32_t lane_next = 0;
	std::vector<RingFrame *> ring;
	std::vector<DevTicket *> dev_tickets;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	unsigned long long *d_maxkey = nullptr; // UpdateHeightmap's max(thr) reduction
	// scratch of the host-memory entry points (hmrm_render*, one caller at a time)
	uint32_t *d_frame = nullptr;
	size_t frame_px = 0;
	uint8_t *h_stage = nullptr; // pinned staging strip of hmrm_render_multi (pageable destination frames)
	size_t h_stage_bytes = 0;
	uint32_t *d_steps = nullptr;
	double *d_entry = nullptr;
	size_t stats_px = 0;
	// launch state per stream; `mu` guards the list and the slot choice (launches themselves are
	// asynchronous), so that threads driving different streams of one scene do not collide
	std::mutex mu;
	std::vector<StreamCtx *> ctxs;
	uint64_t clock = 0;
	// launch orders already settled by some stream's calibration, by camera: a record built for the same camera on
	// another stream (frames in flight) adopts the result instead of measuring again
	struct SettledOrder {
		hmrm_camera cam;
		hmrm_scene_params params;
		uint64_t thr_max_bits;
		hmrm::OrderTrial order;
	};
	// Which kernel suits the scene's content (launch_order.hpp KernelChoice): probed ONCE per scene and height update, by
	// the first camera that gets calibrated (two launches of the plain groups cost 3-6 ms on a 4K frame where leaps pay: too
	// much to spend per camera); every other camera -- a moving one is never calibrated -- renders with that verdict.
	hmrm::KernelChoice choice;
	// ... and a scene whose cameras never repeat (a moving camera: nothing is ever calibrated) is probed on its
	// kProbeAfterFrames-th full frame instead: that frame is launched twice into the same buffer -- production kernel,
	// then plain groups; same pixels -- both measured like a calibration launch, and the verdict is read when the second
	// has finished (any later launch looks).  One 3-6 ms hiccup per scene on a 4K frame; a caller that cannot have it
	// passes HMRM_NO_PROBE with its ticketed frames (hmrm.h).
	bool probe_pending = false;
	uint32_t probe_epoch = 0, probe_epoch_launched = 0; // (a height update or a knob reload voids a probe in flight)
	int probe_rows = 0;
	StreamCtx *probe_ctx = nullptr;
	hipEvent_t probe_done = nullptr;
	unsigned long long *h_probe = nullptr, *h_probe_dev = nullptr; // pinned: 2 x (2 x kMaxMeasRows) words
	// A measured launch (calibration trial or probe) wants the chip to itself: it is issued only when the scene's other
	// streams are idle, and launches on those streams wait for it (overlapping lanes inflated or deflated trials).
	hipEvent_t measure_fence = nullptr;
	StreamCtx *measure_fence_ctx = nullptr;
	std::vector<SettledOrder> settled;
};

// cached records and settled launch orders are looked up with memcmp on these two: no padding bytes allowed
static_assert(sizeof(hmrm_camera) == 3 * sizeof(int32_t) + 4 + 8 * sizeof(double), "hmrm_camera has padding");
static_assert(sizeof(hmrm_scene_params) == 6 * sizeof(double), "hmrm_scene_params has padding");

struct hmrm_config {
	hmrm::Config cfg;
	std::string log_cache, warn_cache;
};

namespace {

void destroy_ctx(StreamCtx *c) {
	if (!c) return;
	for (FrameSlot &sl : c->slots) {
		if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
		if (sl.measured) (void)hipEventDestroy(sl.measured);
	}
	if (c->d_meas) (void)hipFree(c->d_meas);
	if (c->h_meas) (void)hipHostFree(c->h_meas);
	if (c->d_arena) (void)hipFree(c->d_arena);
	if (c->h_arena) (void)hipHostFree(c->h_arena);
	if (c->d_counters) (void)hipFree(c->d_counters);
	if (c->last_launch) (void)hipEventDestroy(c->last_launch);
	delete c;
}

// The launch state of `stream` (created on first use; the least recently used one is dropped, after
// its last launch has finished, when a scene is driven from more than kMaxStreamCtx streams).
int ctx_for(hmrm_scene *s, hipStream_t stream, StreamCtx **out) {
	for (StreamCtx *c : s->ctxs)
		if (c->stream == stream) {
			c->stamp = ++s->clock;
			*out = c;
			return HMRM_OK;
		}
	if ((int)s->ctxs.size() >= kMaxStreamCtx) {
		size_t victim = 0; // (never the scene's own stream -- entry 0 -- or one of its launch lanes)
		for (size_t i = 1; i < s->ctxs.size(); ++i)
			if (!s->ctxs[i]->scene_owned && (victim == 0 || s->ctxs[i]->stamp < s->ctxs[victim]->stamp)) victim = i;
		if (victim == 0) return fail(HMRM_E_ARG, "too many streams");
		// its kernels may still read the tables / counters about to be freed.  The stream belongs to the caller and
		// may have been destroyed since (no call may name it any more): wait on the context's own event instead
		(void)hipEventSynchronize(s->ctxs[victim]->last_launch);
		if (s->probe_ctx == s->ctxs[victim]) s->probe_ctx = nullptr;
		if (s->measure_fence_ctx == s->ctxs[victim]) s->measure_fence_ctx = nullptr;
		destroy_ctx(s->ctxs[victim]);
		s->ctxs.erase(s->ctxs.begin() + (long)victim);
	}
	StreamCtx *c = new (std::nothrow) StreamCtx();
	if (!c) return fail(HMRM_E_ARG, "out of memory");
	c->stream = stream;
	c->scene_owned = stream == s->stream;
	for (hipStream_t lane : s->lanes) c->scene_owned = c->scene_owned || (lane && stream == lane);
	c->stamp = ++s->clock;
	hipError_t e = hipMalloc((void **)&c->d_counters, 16 * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->last_launch, hipEventDisableTiming);
	// (zeroed on the scene's stream and waited for: the caller's stream may be anything)
	if (e == hipSuccess) e = hipMemsetAsync(c->d_counters, 0, 16 * sizeof(unsigned long long), s->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
	if (e != hipSuccess) {
		destroy_ctx(c);
		return fail(HMRM_E_DEVICE, std::string("stream context: ") + hipGetErrorString(e));
	}
	s->ctxs.push_back(c);
	*out = c;
	return HMRM_OK;
}

int check_camera(const hmrm_camera *cam) {
	if (!cam) return fail(HMRM_E_ARG, "camera is NULL");
	if (cam->width <= 0 || cam->height <= 0) return fail(HMRM_E_ARG, "resolution must be positive");
	if ((int64_t)cam->width * cam->height > ((int64_t)1 << 31) / 4)
		return fail(HMRM_E_ARG, "resolution too large (the reference indexes the framebuffer with int)");
	if (cam->projection < 1 || cam->projection > 3) return fail(HMRM_E_ARG, "projection must be 1, 2 or 3");
	if (cam->sampling > HMRM_NEAREST_F32) return fail(HMRM_E_ARG, "sampling must be 0 (nearest), 1 (bilinear) or 2 (nearest, float heights)");
	return HMRM_OK;
}

int ensure_bilinear_pyramid(hmrm_scene *s);
int ensure_records(hmrm_scene *s);

void to_host_camera(const hmrm_camera *cam, hmrm::HostCamera *hc) {
	*hc = hmrm::HostCamera{};
	hc->width = cam->width;
	hc->height = cam->height;
	hc->projection = cam->projection;
	hc->bg_r = cam->bg_r;
	hc->bg_g = cam->bg_g;
	hc->bg_b = cam->bg_b;
	hc->sampling = cam->sampling;
	hc->hfov = cam->hfov;
	hc->hang = cam->hang;
	hc->vang = cam->vang;
	hc->pos[0] = cam->pos[0];
	hc->pos[1] = cam->pos[1];
	hc->pos[2] = cam->pos[2];
	hc->ortho_width = cam->ortho_width;
	hc->step_dist = cam->step_dist;
}

// Host set-up for one frame (the reference rebuilds its ImagePlane every frame, hmap.cpp:952-965;
// same values): finds the stream's cached record for this camera or builds it into the least
// recently used slot; for spherical cameras the table upload is enqueued on the launch stream, in
// front of the kernel that reads it.  *slot_out stays valid until the next call on this context.
int prepare_frame(hmrm_scene *s, StreamCtx *c, const hmrm_camera *cam, hmrm::DevFrame *f, FrameSlot **slot_out) {
	const uint64_t thr_bits = *(const uint64_t *)&s->thr_max;
	FrameSlot *slot = nullptr;
	for (FrameSlot &sl : c->slots)
		if (sl.valid && memcmp(&sl.cam, cam, sizeof *cam) == 0 && memcmp(&sl.params, &s->params, sizeof s->params) == 0 &&
		    sl.thr_max_bits == thr_bits)
			slot = &sl;
	if (!slot) {
		slot = &c->slots[0];
		for (FrameSlot &sl : c->slots)
			if (!sl.valid || sl.stamp < slot->stamp) {
				slot = &sl;
				if (!sl.valid) break;
			}
		slot->valid = false;
		if (slot->cal.in_flight >= 0) HIP_TRY(hipEventSynchronize(slot->measured)); // (its records are about to be reused)
		slot->cal.reset();
		hmrm::HostCamera hc;
		to_host_camera(cam, &hc);
		double *cc = nullptr, *cs = nullptr, *rs = nullptr, *rc = nullptr;
		const size_t W = (size_t)cam->width, H = (size_t)cam->height;
		if (cam->projection == HMRM_SPHERICAL) {
			const size_t n = 2 * W + 2 * H;
			if (n > c->arena_n) {
				// (kernels of this stream may still read the old tables; every cached spherical record of the context
				// loses its tables with the old arena)
				HIP_TRY(hipStreamSynchronize(c->stream));
				if (c->d_arena) (void)hipFree(c->d_arena);
				if (c->h_arena) (void)hipHostFree(c->h_arena);
				c->d_arena = c->h_arena = nullptr;
				c->arena_n = 0;
				for (FrameSlot &sl : c->slots) {
					if (sl.cam.projection == HMRM_SPHERICAL) sl.valid = false;
					sl.d_tables = sl.h_tables = nullptr;
				}
				const size_t per = (n + 31) & ~(size_t)31; // (slots start on 256-byte boundaries)
				HIP_TRY(hipMalloc((void **)&c->d_arena, per * kFrameSlots * sizeof(double)));
				HIP_TRY(hipHostMalloc((void **)&c->h_arena, per * kFrameSlots * sizeof(double), hipHostMallocMapped));
				HIP_TRY(hipHostGetDevicePointer((void **)&c->h_arena_dev, c->h_arena, 0));
				c->arena_n = per;
			}
			const size_t idx = (size_t)(slot - c->slots);
			slot->d_tables = c->d_arena + idx * c->arena_n;
			slot->h_tables = c->h_arena + idx * c->arena_n;
			if (!slot->uploaded) HIP_TRY(hipEventCreateWithFlags(&slot->uploaded, hipEventDisableTiming));
			else HIP_TRY(hipEventSynchronize(slot->uploaded)); // the previous upload out of h_tables is done
			cc = slot->h_tables;
			cs = cc + W;
			rs = cs + W;
			rc = rs + H;
		}
		hmrm::DevFrame &fr = slot->frame;
		hmrm::build_frame(hc, s->map_w, s->map_h, s->params.min_height, s->params.max_height, s->params.grid_width,
		                  &fr, nullptr, nullptr, nullptr, nullptr);
		if (cam->projection == HMRM_SPHERICAL) {
			// A moving camera seldom changes everything: the row tables (sin / cos of va) depend on vang, hfov and
			// the resolution only -- they survive any orbit or translation -- and the column tables on hang, hfov
			// and the width.  A half some cached record of this stream already holds is copied from its staging
			// memory; what is left is filled by the host pool (glibc sin / cos, ~15 ns per call: 12 000 calls for
			// a fresh 4K camera would otherwise cost more host time than the kernel takes on the GPU).
			const FrameSlot *row_donor = nullptr, *col_donor = nullptr;
			for (const FrameSlot &o : c->slots) {
				if (!o.valid || &o == slot || o.cam.projection != HMRM_SPHERICAL || !o.h_tables) continue;
				if (o.cam.width != cam->width || o.cam.hfov != cam->hfov) continue;
				if (!col_donor && o.cam.hang == cam->hang) col_donor = &o;
				if (!row_donor && o.cam.height == cam->height && o.cam.vang == cam->vang) row_donor = &o;
			}
			if (col_donor) memcpy(cc, col_donor->h_tables, 2 * W * sizeof(double));
			if (row_donor) memcpy(rs, row_donor->h_tables + 2 * (size_t)row_donor->cam.width, 2 * H * sizeof(double));
			const int nc = col_donor ? 0 : cam->width, nr = row_donor ? 0 : cam->height;
			if (nc + nr > 0)
				hmrm::parallel_ranges(nc + nr, 1024, [&](int b, int e) {
					// items [0, nc) are columns, [nc, nc + nr) rows
					if (b < nc) hmrm::fill_col_tables(hc, b, std::min(e, nc), cc, cs);
					if (e > nc) hmrm::fill_row_tables(hc, std::max(b, nc) - nc, e - nc, rs, rc);
				});
		}
		slot->row_cost.assign(((size_t)cam->height + kCostRows - 1) / kCostRows, 0.0f);
		hmrm::estimate_row_costs(fr, cc, cs, rs, rc, kCostRows, slot->row_cost.data());
		if (cam->projection == HMRM_SPHERICAL) {
			// stream order puts the upload (a small kernel reading the pinned staging memory) behind every earlier
			// kernel of this stream that read the slot's device tables and in front of the one about to be launched
			HIP_TRY(hmrm::launch_upload_tables(c->h_arena_dev + (slot->h_tables - c->h_arena), slot->d_tables, 2 * W + 2 * H, c->stream));
			HIP_TRY(hipEventRecord(slot->uploaded, c->stream));
			fr.col_cos_ha = slot->d_tables;
			fr.col_sin_ha = slot->d_tables + W;
			fr.row_sin_va = slot->d_tables + 2 * W;
			fr.row_cos_va = slot->d_tables + 2 * W + H;
		}
		if (cam->sampling == HMRM_BILINEAR) {
			const int rc2 = ensure_bilinear_pyramid(s);
			if (rc2 != HMRM_OK) return rc2;
		}
		// (informational: the kernel reads the whole-map bound from the pyramid's top plane, rounded up to
		// float like every window maximum -- which also bounds the float copy of a threshold)
		fr.thr_max = cam->sampling == HMRM_BILINEAR ? s->thr_max_bil : s->thr_max;
		// the finest level whose windows have at least min_window cells (camera.cpp's hint)
		fr.min_level = 0;
		while (fr.min_level < hmrm::kMipLevels - 1 && hmrm::win_cells(fr.min_level) < fr.min_window)
			++fr.min_level;
		fr.mipbuf = s->d_mipbuf;
		fr.mipbuf_bil = s->d_mipbuf_bil;
		fr.mip_row = s->mip_row;
		fr.mip_plane_shift = s->mip_plane_shift;
		slot->cam = *cam;
		slot->params = s->params;
		slot->thr_max_bits = thr_bits;
		slot->valid = true;
		for (size_t k = s->settled.size(); k-- > 0;) { // (newest first)
			const hmrm_scene::SettledOrder &so = s->settled[k];
			if (memcmp(&so.cam, cam, sizeof *cam) != 0 || memcmp(&so.params, &s->params, sizeof s->params) != 0 || so.thr_max_bits != thr_bits) continue;
			slot->cal.adopt(so.order);
			break;
		}
	}
	slot->stamp = ++s->clock;
	*f = slot->frame;
	// per-scene settings (not part of the cached record)
	f->step_cap = s->knobs.step_cap;
	f->diag_mode = s->knobs.diag_mode;
	if (s->knobs.min_level >= 0) f->min_level = s->knobs.min_level;
	if (s->knobs.finest_pause >= 0) f->finest_pause = s->knobs.finest_pause;
	if (slot_out) *slot_out = slot;
	return HMRM_OK;
}

// The calibration records of a context: the device buffer of the ONE measured launch it has in flight at a time and, per
// cached record, kMaxMeasRows x {start, longest wave} in pinned memory (allocated with the first measured launch).
int ensure_meas(StreamCtx *c) {
	if (!c->d_meas) HIP_TRY(hipMalloc((void **)&c->d_meas, (size_t)kMaxMeasRows * hmrm::kMeasureStride * sizeof(unsigned long long)));
	if (!c->h_meas) {
		const size_t n = (size_t)kFrameSlots * 2 * kMaxMeasRows * sizeof(unsigned long long);
		HIP_TRY(hipHostMalloc((void **)&c->h_meas, n, hipHostMallocMapped));
		HIP_TRY(hipHostGetDevicePointer((void **)&c->h_meas_dev, c->h_meas, 0));
	}
	return HMRM_OK;
}

// ---- measured launches: the HIP side of launch_order.hpp's calibration ----
// Is nothing of the scene running on another of its own streams?  (A measured launch wants the chip to itself.)
bool others_idle(hmrm_scene *s, StreamCtx *c) {
	for (StreamCtx *o : s->ctxs)
		if (o != c && o->scene_owned && o->launched && hipEventQuery(o->last_launch) != hipSuccess) return false;
	return true;
}

// A measured launch of another context is still running: this context's launch waits for it (stream order, no host wait).
int wait_for_measure_fence(hmrm_scene *s, StreamCtx *c) {
	if (!s->measure_fence_ctx || s->measure_fence_ctx == c) return HMRM_OK;
	if (hipEventQuery(s->measure_fence) == hipSuccess) s->measure_fence_ctx = nullptr;
	else HIP_TRY(hipStreamWaitEvent(c->stream, s->measure_fence, 0));
	return HMRM_OK;
}

int raise_measure_fence(hmrm_scene *s, StreamCtx *c) {
	if (!s->measure_fence) HIP_TRY(hipEventCreateWithFlags(&s->measure_fence, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(s->measure_fence, c->stream));
	s->measure_fence_ctx = c;
	return HMRM_OK;
}

// A finished measured launch of this record is folded into its calibration; a calibration that settles is published.
void poll_measured(hmrm_scene *s, StreamCtx *c, FrameSlot *slot, int tiles_y, int rot) {
	if (slot->cal.in_flight >= 0 && slot->meas_rows == tiles_y && hipEventQuery(slot->measured) == hipSuccess) {
		const int trial = slot->cal.in_flight;
		const unsigned long long *h_rec = c->h_meas + (size_t)(slot - c->slots) * 2 * kMaxMeasRows;
		const bool settled = slot->cal.on_measured(h_rec, tiles_y, rot, s->knobs.kernel == 0 && s->knobs.try_group, s->choice);
		if (s->knobs.order_verbose) {
			const hmrm::OrderTrial &t = slot->cal.trials[trial];
			fprintf(stderr, "hmrm order: trial %d of %d tile rows measured, makespan %.1f us:", trial, tiles_y, t.makespan / 100.0);
			for (int k = 0; k < t.n; ++k) fprintf(stderr, " [%d,%d)", t.b[k], t.b[k] + t.c[k]);
			fprintf(stderr, "%s\n", t.group ? " plain groups, rotation" : (t.n ? "" : " rotation"));
			if (settled) fprintf(stderr, "hmrm order: settled on trial %d\n", slot->cal.best);
		}
		if (settled) {
			if (s->settled.size() >= 256) s->settled.erase(s->settled.begin());
			s->settled.push_back(hmrm_scene::SettledOrder{slot->cam, slot->params, slot->thr_max_bits, slot->cal.trials[slot->cal.best]});
		}
	}
	// (one measured launch per context at a time: they share the device records)
	if (c->meas_owner && (c->meas_owner->cal.in_flight < 0 || hipEventQuery(c->meas_owner->measured) == hipSuccess)) c->meas_owner = nullptr;
}

// The verdict of a shadow probe in flight, once both of its launches have finished.
void poll_shadow_probe(hmrm_scene *s) {
	if (!s->probe_pending || hipEventQuery(s->probe_done) != hipSuccess) return;
	s->probe_pending = false;
	if (s->probe_ctx) s->probe_ctx->probe_in_flight = false;
	if (s->probe_epoch_launched != s->probe_epoch) return; // (the heights or the knobs changed meanwhile)
	hmrm::fold_shadow_probe(s->choice, s->h_probe, s->h_probe + 2 * kMaxMeasRows, s->probe_rows);
	if (s->knobs.order_verbose)
		fprintf(stderr, "hmrm probe: production kernel %.1f us, the other kernel %.1f us -> %s\n", hmrm::measured_makespan(s->h_probe, s->probe_rows) / 100.0,
		        hmrm::measured_makespan(s->h_probe + 2 * kMaxMeasRows, s->probe_rows) / 100.0, s->choice.use_group ? (s->choice.with_records ? "groups + window records" : "plain groups") : "production kernel");
}

// The shadow probe: this frame twice into the same buffer -- production kernel, then the other kernel (launch_kernel: the
// plain groups, with window records for nearest-sampling frames); same pixels, the
// second launch's capped rays counted apart -- both measured like a calibration launch.
int launch_shadow_probe(hmrm_scene *s, StreamCtx *c, const hmrm::DevFrame &f, const hmrm::RowMap &rows_in_order, int tiles_y,
                        uint32_t *d_out, int64_t out_stride_px) {
	if (!s->h_probe) {
		HIP_TRY(hipHostMalloc((void **)&s->h_probe, (size_t)4 * kMaxMeasRows * sizeof(unsigned long long), hipHostMallocMapped));
		HIP_TRY(hipHostGetDevicePointer((void **)&s->h_probe_dev, s->h_probe, 0));
		HIP_TRY(hipEventCreateWithFlags(&s->probe_done, hipEventDisableTiming));
	}
	const int rc_m = ensure_meas(c);
	if (rc_m) return rc_m;
	hmrm::RowMap measured = rows_in_order;
	measured.measure = c->d_meas;
	const bool records_ok = f.sampling == 0 && s->d_records; // (the other kernel of this frame: launch_kernel)
	s->choice.with_records = records_ok;
	if (records_ok) {
		const int rc_r = ensure_records(s);
		if (rc_r) return rc_r;
	}
	for (int pass = 0; pass < 2; ++pass) {
		HIP_TRY(hmrm::launch_measure_init(measured.measure, tiles_y, c->stream));
		HIP_TRY(hmrm::launch_render_fast(f, measured, s->d_thr, s->d_thr32, s->d_cmap, d_out, out_stride_px,
		                                 c->d_counters + (pass ? 8 : 0), nullptr, nullptr, false,
		                                 pass == 0 ? hmrm::kLeaps : (records_ok ? hmrm::kRecords : hmrm::kPlainGroups), s->d_records, c->stream));
		HIP_TRY(hmrm::launch_measure_readback(measured.measure, s->h_probe_dev + (size_t)pass * 2 * kMaxMeasRows, tiles_y, c->stream));
	}
	HIP_TRY(hipEventRecord(s->probe_done, c->stream));
	s->probe_pending = true;
	s->choice.probed = true;
	s->probe_rows = tiles_y;
	s->probe_epoch_launched = s->probe_epoch;
	s->probe_ctx = c;
	c->probe_in_flight = true;
	return raise_measure_fence(s, c);
}

// The render kernel itself: the literal loop (HMRM_KERNEL=simple, or a map with a side of 2^24 cells or more -- the production
// kernel indexes cells and windows with 24-bit multiplies, leap_common.hpp index_2d), else the production kernel or the other one (the plain groups,
// with leaps over window records where the frame's sampling allows them).
int launch_kernel(hmrm_scene *s, StreamCtx *c, const hmrm::DevFrame &f, const hmrm::RowMap &rows_in_order, uint32_t *d_out,
                  int64_t out_stride_px, uint32_t *d_steps, double *d_entry, bool stats, bool use_group) {
	const bool huge_side = s->map_w >= (1 << 24) || s->map_h >= (1 << 24);
	if (huge_side && f.sampling != 0)
		return fail(HMRM_E_ARG, "maps with a side of 2^24 cells or more support nearest sampling only");
	if ((s->knobs.kernel == 2 || huge_side) && f.sampling == 0) { // (the literal loop only knows the reference's sampling)
		HIP_TRY(hmrm::launch_render(f, rows_in_order, s->d_thr, s->d_cmap, d_out, out_stride_px, c->d_counters, d_steps,
		                            d_entry, stats, c->stream));
	} else {
		// (the records bound the nearest cell's double thresholds: launch_order.hpp pick_fast_kernel)
		const hmrm::FastKernel k = (hmrm::FastKernel)hmrm::pick_fast_kernel(s->knobs.kernel, use_group, f.sampling == 0 && s->d_records, s->choice);
		if (k == hmrm::kRecords) {
			const int rc_r = ensure_records(s);
			if (rc_r) return rc_r;
		}
		HIP_TRY(hmrm::launch_render_fast(f, rows_in_order, s->d_thr, s->d_thr32, s->d_cmap, d_out, out_stride_px,
		                                 c->d_counters, d_steps, d_entry, stats, k, s->d_records, c->stream));
	}
	return HMRM_OK;
}

// Behind every launch on a stream other than the scene's own: the event a recycled context waits for (ctx_for) and a measured
// launch on another of the scene's streams looks at (others_idle).
int note_launch(hmrm_scene *s, StreamCtx *c) {
	if (c->scene_owned && c->stream == s->stream) return HMRM_OK;
	HIP_TRY(hipEventRecord(c->last_launch, c->stream));
	c->launched = true;
	return HMRM_OK;
}

// The kernel launch -- bracketed, when it is a measured one, by the set-up of the records in front of it and their read-back,
// the record's event and the scene's fence behind it.
int launch_maybe_measured(hmrm_scene *s, StreamCtx *c, const hmrm::DevFrame &f, FrameSlot *slot, hmrm::RowMap &rows_in_order, int tiles_y,
                          bool measure_now, uint32_t *d_out, int64_t out_stride_px, uint32_t *d_steps, double *d_entry, bool stats, bool use_group) {
	if (measure_now) {
		const int rc_m = ensure_meas(c);
		if (rc_m) return rc_m;
		if (!slot->measured) HIP_TRY(hipEventCreateWithFlags(&slot->measured, hipEventDisableTiming));
		rows_in_order.measure = c->d_meas;
		HIP_TRY(hmrm::launch_measure_init(rows_in_order.measure, tiles_y, c->stream));
	}
	const int rc_k = launch_kernel(s, c, f, rows_in_order, d_out, out_stride_px, d_steps, d_entry, stats, use_group);
	if (rc_k || !measure_now) return rc_k;
	const size_t idx = (size_t)(slot - c->slots);
	HIP_TRY(hmrm::launch_measure_readback(rows_in_order.measure, c->h_meas_dev + idx * 2 * kMaxMeasRows, tiles_y, c->stream));
	HIP_TRY(hipEventRecord(slot->measured, c->stream));
	slot->meas_rows = tiles_y;
	c->meas_owner = slot;
	return raise_measure_fence(s, c);
}

// One frame (or row strip) on the context's stream.  Kernel variant: "leap" (default; speculative
// groups + exact leaps), "group" (speculative groups only), "simple" (the literal
// one-step-at-a-time loop, kept for A/B runs and as an in-library cross-check).  All produce
// identical pixels and counts.  `no_probe`: the caller cannot have this frame launched twice (HMRM_NO_PROBE).
int launch_frame(hmrm_scene *s, StreamCtx *c, const hmrm::DevFrame &f, FrameSlot *slot, const hmrm::RowMap &rows,
                 uint32_t *d_out, int64_t out_stride_px, uint32_t *d_steps, double *d_entry, bool stats, bool no_probe = false) {
	hmrm::RowMap rows_in_order = rows;
	int tile_w = 1, tile_h = 1;
	hmrm::render_tile_shape(&tile_w, &tile_h);
	const int tiles_y = (rows.local_rows + tile_h - 1) / tile_h;
	const bool pieces = s->knobs.seg_n > 0 && rows.band_rows == 0 && rows.row_begin == 0; // (HMRM_TILE_SEGMENTS: an explicit order)
	const int rot = hmrm::choose_tile_rot(s->knobs.tile_order, slot->row_cost, rows, tile_h);
	const bool may_probe = !stats && s->knobs.kernel == 0 && s->knobs.try_group;
	// calibration (launch_order.hpp): full frames of the fast kernels only
	const bool eligible = !pieces && !stats && s->knobs.tile_order && s->knobs.order_mode == 2 && s->knobs.kernel != 2 &&
	                      rows.band_rows == 0 && rows.row_begin == 0 && rows.local_rows == f.screen_h && tiles_y >= 12 &&
	                      tiles_y <= kMaxMeasRows && s->map_w < (1 << 24) && s->map_h < (1 << 24);
	int rc = wait_for_measure_fence(s, c);
	if (rc) return rc;
	poll_shadow_probe(s);
	hmrm::OrderTrial order; // (the rotation)
	if (pieces) {
		order.n = s->knobs.seg_n;
		for (int k = 0; k < 3; ++k) { order.b[k] = s->knobs.seg_b[k]; order.c[k] = s->knobs.seg_c[k]; }
	}
	bool use_group = may_probe && s->choice.use_group; // strips, bands, small frames: the scene's verdict
	bool measure_now = false;
	if (eligible) {
		poll_measured(s, c, slot, tiles_y, rot);
		// (whether a measurement may run is only looked up -- event queries on the scene's other streams -- while one is wanted)
		const bool probe_open = may_probe && !no_probe && !s->choice.probed && !s->probe_pending;
		const bool quiet = (slot->cal.wants_measure() || probe_open) && c->meas_owner == nullptr && !c->probe_in_flight && others_idle(s, c);
		const hmrm::LaunchPlan p = slot->cal.plan(quiet, s->choice);
		order = slot->cal.trials[p.trial];
		use_group = p.use_group;
		measure_now = p.measure;
		if (!measure_now && probe_open && quiet && slot->cal.in_flight < 0 && ++s->choice.unprobed_frames >= (unsigned)kProbeAfterFrames) {
			hmrm::set_tile_order(&rows_in_order, tiles_y, rot, order.n, order.b, order.c);
			if ((rc = launch_shadow_probe(s, c, f, rows_in_order, tiles_y, d_out, out_stride_px))) return rc;
			return note_launch(s, c);
		}
	}
	hmrm::set_tile_order(&rows_in_order, tiles_y, rot, order.n, order.b, order.c);
	// (a measured launch that cannot be issued or reported must not stay "in flight" in the record: nothing would ever
	// report it, and the record's next reuse would wait for an event that was never recorded)
	if ((rc = launch_maybe_measured(s, c, f, slot, rows_in_order, tiles_y, measure_now, d_out, out_stride_px, d_steps, d_entry, stats, use_group))) {
		if (measure_now) slot->cal.drop_in_flight();
		return rc;
	}
	return note_launch(s, c);
}

// Pyramid layout of a map (DevFrame): windows per level, the common row pitch (level 0's) and the log2 of the
// power-of-two plane pitch (at least min_shift: tests force far-apart planes on small maps).  The kernel forms the
// ELEMENT index (lev << shift) + index in 32 bits -- (kMipLevels + 1) << shift never exceeds 2^31 for a map within
// hmrm_scene_create's 2^29-cell limit -- and the byte offset in 64.  Returns whether 32-bit BYTE offsets would have been
// enough (informational since round 5: every square-ish map; not 16385 x 32766, whose level 0 has 8193 x 16383 windows,
// shift 28 -- round 4 rendered such maps with the literal loop).
bool mip_layout(int32_t map_w, int32_t map_h, int32_t *mip_w, int32_t *mip_h, int32_t *mip_row, int32_t *plane_shift, int32_t min_shift = 0) {
	for (int l = 0; l < hmrm::kMipLevels; ++l) {
		const int stride = 1 << hmrm::mip_stride_shift(l); // windows of win_cells(l) cells every stride cells
		mip_w[l] = (map_w + stride - 1) / stride;
		mip_h[l] = (map_h + stride - 1) / stride;
	}
	*mip_row = mip_w[0];
	*plane_shift = min_shift;
	while (((size_t)1 << *plane_shift) < (size_t)hmrm::mip_index(mip_w[0] - 1, mip_h[0] - 1, *mip_row) + 1) ++*plane_shift;
	return ((uint64_t)(hmrm::kMipLevels + 1) << *plane_shift) <= ((uint64_t)1 << 30);
}

// The launch state of the next lane in turn (lanes are created on first use; call with s->mu held).
int next_lane(hmrm_scene *s, StreamCtx **out) {
	const uint32_t i = s->lane_next++ % kLanes;
	if (!s->lanes[i]) HIP_TRY(hipStreamCreateWithFlags(&s->lanes[i], hipStreamNonBlocking));
	return ctx_for(s, s->lanes[i], out);
}

int ensure_frame(hmrm_scene *s, size_t px) {
	if (px <= s->frame_px) return HMRM_OK;
	if (s->d_frame) (void)hipFree(s->d_frame);
	s->d_frame = nullptr;
	s->frame_px = 0;
	HIP_TRY(hipMalloc((void **)&s->d_frame, px * sizeof(uint32_t)));
	s->frame_px = px;
	return HMRM_OK;
}

int ensure_stats(hmrm_scene *s, size_t px) {
	if (px <= s->stats_px) return HMRM_OK;
	if (s->d_steps) (void)hipFree(s->d_steps);
	if (s->d_entry) (void)hipFree(s->d_entry);
	s->d_steps = nullptr;
	s->d_entry = nullptr;
	s->stats_px = 0;
	HIP_TRY(hipMalloc((void **)&s->d_steps, px * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void **)&s->d_entry, px * sizeof(double)));
	s->stats_px = px;
	return HMRM_OK;
}

// The window records (frame.hpp), built the first time the record kernel is about to run after a height update -- by the
// scene's probe, a verdict for it, or HMRM_KERNEL=rec -- so that a caller who changes the heights every frame, and is never
// probed, does not pay for a table nobody reads (0.26-0.48 ms at 4096 x 4096).  Nothing reads the table while it is
// written: a height update drains the scene's streams, and no record kernel has been launched since.
int ensure_records(hmrm_scene *s) {
	if (s->records_valid) return HMRM_OK;
	if (!s->d_records) return fail(HMRM_E_ARG, "this scene has no window records");
	HIP_TRY(hmrm::launch_build_records(s->d_thr, s->map_w, s->map_h, s->d_records, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->records_valid = true;
	return HMRM_OK;
}

// Bilinear quality mode only: the second pyramid, over the 3x3-dilated (and margin-padded)
// table, built the first time a bilinear frame is asked for after a height update.
int ensure_bilinear_pyramid(hmrm_scene *s) {
	if (s->bil_valid) return HMRM_OK;
	const int64_t n = (int64_t)s->map_w * s->map_h;
	if (!s->d_mipbuf_bil) HIP_TRY(hipMalloc((void **)&s->d_mipbuf_bil, s->mip_floats() * sizeof(float)));
	double *tmp = nullptr;
	HIP_TRY(hipMalloc((void **)&tmp, (size_t)n * sizeof(double)));
	hipError_t e = hmrm::launch_dilate3x3(s->d_thr, s->map_w, s->map_h, tmp, s->stream);
	if (e == hipSuccess)
		e = hmrm::launch_build_mip0(tmp, s->map_w, s->map_h, s->plane(s->d_mipbuf_bil, 0), s->mip_w[0], s->mip_h[0],
		                            s->mip_row, s->stream);
	for (int l = 1; l < hmrm::kMipLevels && e == hipSuccess; ++l)
		e = hmrm::launch_build_mip_up(s->plane(s->d_mipbuf_bil, l - 1), s->mip_w[l - 1], s->mip_h[l - 1],
		                              s->plane(s->d_mipbuf_bil, l), s->mip_w[l], s->mip_h[l], s->mip_row, l - 1, s->stream);
	// whole-map bound = max over the coarsest level (its windows cover every cell)
	const int top = hmrm::kMipLevels - 1;
	const size_t top_span = (size_t)hmrm::mip_index(s->mip_w[top] - 1, s->mip_h[top] - 1, s->mip_row) + 1;
	std::vector<float> coarse(top_span);
	if (e == hipSuccess)
		e = hipMemcpyAsync(coarse.data(), s->plane(s->d_mipbuf_bil, top), top_span * sizeof(float), hipMemcpyDeviceToHost, s->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
	(void)hipFree(tmp);
	if (e != hipSuccess) return fail(HMRM_E_DEVICE, std::string("bilinear pyramid: ") + hipGetErrorString(e));
	float m = -__builtin_huge_valf();
	for (int iy = 0; iy < s->mip_h[top]; ++iy)
		for (int ix = 0; ix < s->mip_w[top]; ++ix) {
			const float v = coarse[hmrm::mip_index(ix, iy, s->mip_row)];
			if (v > m) m = v;
		}
	// (pageable source: the copy is staged before the call returns)
	e = hipMemcpyAsync(s->plane(s->d_mipbuf_bil, hmrm::kMipLevels), &m, sizeof m, hipMemcpyHostToDevice, s->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
	if (e != hipSuccess) return fail(HMRM_E_DEVICE, std::string("bilinear pyramid: ") + hipGetErrorString(e));
	s->thr_max_bil = (double)m;
	s->bil_valid = true;
	return HMRM_OK;
}

// UpdateHeightmap + pyramid on the scene's stream.  Streams other than the scene's own must not
// have launches in flight while the heights change (as in the reference, where UpdateHeightmap runs
// between frames, hmap.cpp:517-519).
int run_update_heights(hmrm_scene *s) {
	const int64_t n = (int64_t)s->map_w * s->map_h;
	// The ticketed entry points launch on the scene's lanes, not on s->stream: frames still in flight there (and ring
	// frames the copy stream has not delivered yet) read the tables about to be rewritten.  Drain every stream the
	// scene owns first; s->stream itself is ordered by the stream.  (Callers' own streams: their business, hmrm.h.)
	for (StreamCtx *c : s->ctxs)
		if (c->scene_owned && c->stream != s->stream) HIP_TRY(hipStreamSynchronize(c->stream));
	if (s->copy_stream) HIP_TRY(hipStreamSynchronize(s->copy_stream));
	HIP_TRY(hipMemsetAsync(s->d_maxkey, 0, sizeof(unsigned long long), s->stream));
	HIP_TRY(hmrm::launch_prepare_heights(s->d_rgb, s->d_thr, n, s->params.lum_r, s->params.lum_g,
	                                     s->params.lum_b, s->params.min_height, s->params.max_height,
	                                     false, s->d_maxkey, s->stream));
	HIP_TRY(hmrm::launch_thr_to_float(s->d_thr, s->d_thr32, n, s->stream)); // (float)thr, round to nearest
	// window-maximum pyramid for the exact-leap traversal (render_fast.hip)
	HIP_TRY(hmrm::launch_build_mip0(s->d_thr, s->map_w, s->map_h, s->plane(s->d_mipbuf, 0), s->mip_w[0], s->mip_h[0],
	                                s->mip_row, s->stream));
	for (int l = 1; l < hmrm::kMipLevels; ++l)
		HIP_TRY(hmrm::launch_build_mip_up(s->plane(s->d_mipbuf, l - 1), s->mip_w[l - 1], s->mip_h[l - 1],
		                                  s->plane(s->d_mipbuf, l), s->mip_w[l], s->mip_h[l], s->mip_row, l - 1, s->stream));
	s->bil_valid = false; // rebuilt by the next bilinear frame
	s->records_valid = false; // ... and the window records by the next launch of the record kernel
	for (StreamCtx *c : s->ctxs)
		for (FrameSlot &sl : c->slots) sl.valid = false;
	s->settled.clear();
	s->choice.reset(); // (new heights: new content)
	++s->probe_epoch;
	unsigned long long key = 0;
	HIP_TRY(hipMemcpyAsync(&key, s->d_maxkey, sizeof key, hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->thr_max = key ? hmrm::max_key_to_double(key) : -__builtin_huge_val();
	// the top plane's one element: the whole-map bound, rounded like the windows
	const float top = hmrm::round_up_to_float_host(s->thr_max);
	HIP_TRY(hipMemcpyAsync(s->plane(s->d_mipbuf, hmrm::kMipLevels), &top, sizeof top, hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return HMRM_OK;
}

// Rays of this context that reached the step cap since the host last asked (the stream must be idle).
int take_capped(StreamCtx *c, unsigned long long *out) {
	unsigned long long now = 0;
	HIP_TRY(hipMemcpy(&now, c->d_counters + 2, sizeof now, hipMemcpyDeviceToHost));
	*out = now - c->capped_seen;
	c->capped_seen = now;
	return HMRM_OK;
}

int noterm(unsigned long long capped) {
	char buf[160];
	snprintf(buf, sizeof buf, "%llu ray(s) reached the step cap; the reference's loop would not terminate for them",
	         capped);
	return fail(HMRM_E_NOTERM, buf);
}

} // namespace

extern "C" {

int hmrm_abi_version(void) { return HMRM_ABI_VERSION; }
const char *hmrm_last_error(void) { return g_error.c_str(); }

int hmrm_device_count(void) {
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) return fail(HMRM_E_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
	return n;
}

int hmrm_set_device(int device) {
	HIP_TRY(hipSetDevice(device));
	return HMRM_OK;
}

int hmrm_scene_create(const uint8_t *height_rgb, const uint8_t *color_rgba, int32_t map_w,
                      int32_t map_h, const hmrm_scene_params *params, hmrm_scene **out) {
	if (!height_rgb || !color_rgba || !params || !out) return fail(HMRM_E_ARG, "NULL argument");
	if (map_w <= 0 || map_h <= 0) return fail(HMRM_E_ARG, "map dimensions must be positive");
	// the reference indexes (gridx + gridy*W)*4 with int (hmap.cpp:1018)
	if ((int64_t)map_w * map_h > ((int64_t)1 << 31) / 4)
		return fail(HMRM_E_ARG, "map too large (the reference indexes the colormap with int)");
	*out = nullptr;
	hmrm_scene *s = new (std::nothrow) hmrm_scene();
	if (!s) return fail(HMRM_E_ARG, "out of memory");
	const size_t n = (size_t)map_w * (size_t)map_h;
	s->map_w = map_w;
	s->map_h = map_h;
	s->params = *params;
	s->knobs = read_knobs();
	int rc = HMRM_OK;
	auto body = [&]() -> int {
		HIP_TRY(hipGetDevice(&s->device));
		HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
		HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreate(&s->ev0));
		HIP_TRY(hipEventCreate(&s->ev1));
		HIP_TRY(hipMalloc((void **)&s->d_rgb, n * 3));
		HIP_TRY(hipMalloc((void **)&s->d_cmap, n * 4));
		HIP_TRY(hipMalloc((void **)&s->d_thr, n * (sizeof(double) + sizeof(float)))); // the table, then its float copy
		s->d_thr32 = reinterpret_cast<float *>(s->d_thr + n);
		// every plane has level 0's row pitch and a power-of-two plane pitch (DevFrame); the pyramid of the
		// bilinear mode is allocated by its first frame
		(void)mip_layout(map_w, map_h, s->mip_w, s->mip_h, &s->mip_row, &s->mip_plane_shift, s->knobs.min_plane_shift);
		HIP_TRY(hipMalloc((void **)&s->d_mipbuf, s->mip_floats() * sizeof(float)));
		if ((map_h + 3) / 4 <= 65535 * 16)
			HIP_TRY(hipMalloc((void **)&s->d_records, (size_t)hmrm::rec_row(map_w) * (size_t)((map_h + 3) / 4) * sizeof(hmrm::WindowRecord)));
		HIP_TRY(hipMalloc((void **)&s->d_maxkey, sizeof(unsigned long long)));
		HIP_TRY(hipMemcpyAsync(s->d_rgb, height_rgb, n * 3, hipMemcpyHostToDevice, s->stream));
		HIP_TRY(hipMemcpyAsync(s->d_cmap, color_rgba, n * 4, hipMemcpyHostToDevice, s->stream));
		StreamCtx *own = nullptr;
		const int rc2 = ctx_for(s, s->stream, &own); // entry 0: the scene's own stream
		if (rc2) return rc2;
		return run_update_heights(s);
	};
	rc = body();
	if (rc != HMRM_OK) {
		std::string keep = g_error;
		hmrm_scene_destroy(s);
		g_error = keep;
		return rc;
	}
	*out = s;
	return HMRM_OK;
}

int hmrm_scene_update(hmrm_scene *s, const hmrm_scene_params *params) {
	if (!s || !params) return fail(HMRM_E_ARG, "NULL argument");
	HIP_TRY(hipSetDevice(s->device));
	std::lock_guard<std::mutex> lk(s->mu);
	s->params = *params;
	return run_update_heights(s);
}

void hmrm_scene_destroy(hmrm_scene *s) {
	if (!s) return;
	(void)hipSetDevice(s->device);
	for (StreamCtx *c : s->ctxs) {
		// (a caller's stream may already be gone: only the scene's own streams are drained)
		if (c->scene_owned && c->stream) (void)hipStreamSynchronize(c->stream);
		destroy_ctx(c);
	}
	s->ctxs.clear();
	if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
	for (RingFrame *r : s->ring) {
		if (r->d_frame) (void)hipFree(r->d_frame);
		if (r->h_frame) (void)hipHostFree(r->h_frame);
		if (r->h_capped) (void)hipHostFree(r->h_capped);
		if (r->kernel_done) (void)hipEventDestroy(r->kernel_done);
		if (r->copy_done) (void)hipEventDestroy(r->copy_done);
		delete r;
	}
	s->ring.clear();
	for (DevTicket *t : s->dev_tickets) {
		if (t->done) (void)hipEventDestroy(t->done);
		if (t->h_capped) (void)hipHostFree(t->h_capped);
		delete t;
	}
	s->dev_tickets.clear();
	for (hipStream_t &lane : s->lanes)
		if (lane) {
			(void)hipStreamDestroy(lane);
			lane = nullptr;
		}
	if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
	if (s->d_rgb) (void)hipFree(s->d_rgb);
	if (s->d_cmap) (void)hipFree(s->d_cmap);
	if (s->d_thr) (void)hipFree(s->d_thr);
	if (s->d_mipbuf) (void)hipFree(s->d_mipbuf);
	if (s->d_records) (void)hipFree(s->d_records);
	if (s->d_mipbuf_bil) (void)hipFree(s->d_mipbuf_bil);
	if (s->d_maxkey) (void)hipFree(s->d_maxkey);
	if (s->d_frame) (void)hipFree(s->d_frame);
	if (s->h_stage) (void)hipHostFree(s->h_stage);
	if (s->h_probe) (void)hipHostFree(s->h_probe);
	if (s->probe_done) (void)hipEventDestroy(s->probe_done);
	if (s->measure_fence) (void)hipEventDestroy(s->measure_fence);
	if (s->d_steps) (void)hipFree(s->d_steps);
	if (s->d_entry) (void)hipFree(s->d_entry);
	if (s->ev0) (void)hipEventDestroy(s->ev0);
	if (s->ev1) (void)hipEventDestroy(s->ev1);
	if (s->stream) (void)hipStreamDestroy(s->stream);
	delete s;
}

int hmrm_debug_kernel_choice(const hmrm_scene *s) {
	if (!s) return fail(HMRM_E_ARG, "NULL argument");
	std::lock_guard<std::mutex> lk(const_cast<hmrm_scene *>(s)->mu);
	if (s->knobs.kernel != 0) return s->knobs.kernel; // forced by HMRM_KERNEL: 1 group, 2 simple, 3 rec
	return s->choice.use_group ? (s->choice.with_records ? 3 : 1) : 0;
}

int hmrm_debug_reload_env(hmrm_scene *s) {
	if (!s) return fail(HMRM_E_ARG, "NULL argument");
	HIP_TRY(hipSetDevice(s->device));
	std::lock_guard<std::mutex> lk(s->mu);
	s->knobs = read_knobs();
	// A launch order is calibrated for ONE kernel variant and level policy: makespans measured before the knobs
	// changed must not be compared with ones measured after, and an order settled for the old kernel must not be
	// adopted for the new one.  Every record goes back to "uncalibrated" (pixels never depended on any of this).
	s->settled.clear();
	s->choice.reset();
	++s->probe_epoch;
	for (StreamCtx *c : s->ctxs)
		for (FrameSlot &sl : c->slots) {
			if (sl.cal.in_flight >= 0 && sl.measured) HIP_TRY(hipEventSynchronize(sl.measured)); // (its read-back targets the slot's records)
			sl.cal.reset();
			if (c->meas_owner == &sl) c->meas_owner = nullptr;
		}
	return HMRM_OK;
}

int hmrm_scene_read_heights(const hmrm_scene *cs, double *out) {
	hmrm_scene *s = const_cast<hmrm_scene *>(cs);
	if (!s || !out) return fail(HMRM_E_ARG, "NULL argument");
	HIP_TRY(hipSetDevice(s->device));
	const int64_t n = (int64_t)s->map_w * s->map_h;
	double *tmp = nullptr;
	HIP_TRY(hipMalloc((void **)&tmp, (size_t)n * sizeof(double)));
	hipError_t e = hmrm::launch_prepare_heights(s->d_rgb, tmp, n, s->params.lum_r, s->params.lum_g,
	                                            s->params.lum_b, s->params.min_height,
	                                            s->params.max_height, true, nullptr, s->stream);
	if (e == hipSuccess)
		e = hipMemcpyAsync(out, tmp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
	(void)hipFree(tmp);
	if (e != hipSuccess) return fail(HMRM_E_DEVICE, std::string("read_heights: ") + hipGetErrorString(e));
	return HMRM_OK;
}

static int render_common(hmrm_scene *s, const hmrm_camera *cam, uint8_t *rgba, size_t stride_bytes,
                         hmrm_stats *stats, uint32_t *steps_pp, double *entry_d, bool want_stats) {
	int rc = check_camera(cam);
	if (rc) return rc;
	if (!s || !rgba) return fail(HMRM_E_ARG, "NULL argument");
	const size_t W = (size_t)cam->width, H = (size_t)cam->height;
	if (stride_bytes < W * 4) return fail(HMRM_E_ARG, "stride_bytes < width*4");
	HIP_TRY(hipSetDevice(s->device));
	std::lock_guard<std::mutex> lk(s->mu);
	if ((rc = ensure_frame(s, W * H))) return rc;
	if (want_stats && (rc = ensure_stats(s, W * H))) return rc;
	StreamCtx *c = nullptr;
	if ((rc = ctx_for(s, s->stream, &c))) return rc;
	hmrm::DevFrame f;
	FrameSlot *slot = nullptr;
	if ((rc = prepare_frame(s, c, cam, &f, &slot))) return rc;
	hmrm::RowMap rows{0, cam->height, 0, 0, 1, {}, {}, nullptr};
	if (want_stats) {
		HIP_TRY(hipMemsetAsync(c->d_counters, 0, 2 * sizeof(unsigned long long), s->stream));
		HIP_TRY(hipMemsetAsync(c->d_counters + 4, 0, 4 * sizeof(unsigned long long), s->stream));
	}
	HIP_TRY(hipEventRecord(s->ev0, s->stream));
	if ((rc = launch_frame(s, c, f, slot, rows, s->d_frame, (int64_t)W, want_stats ? s->d_steps : nullptr,
	                       want_stats ? s->d_entry : nullptr, want_stats)))
		return rc;
	HIP_TRY(hipEventRecord(s->ev1, s->stream));
	HIP_TRY(hipMemcpy2DAsync(rgba, stride_bytes, s->d_frame, W * 4, W * 4, H, hipMemcpyDeviceToHost,
	                         s->stream));
	unsigned long long counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	HIP_TRY(hipMemcpyAsync(counters, c->d_counters, sizeof counters, hipMemcpyDeviceToHost, s->stream));
	if (want_stats && steps_pp)
		HIP_TRY(hipMemcpyAsync(steps_pp, s->d_steps, W * H * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
	if (want_stats && entry_d)
		HIP_TRY(hipMemcpyAsync(entry_d, s->d_entry, W * H * sizeof(double), hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	float ms = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
	g_last_kernel_ms = ms;
	const unsigned long long capped = counters[2] - c->capped_seen;
	c->capped_seen = counters[2];
	if (stats) {
		stats->rays = (uint64_t)W * H;
		stats->steps = counters[0];
		stats->hits = counters[1];
		stats->capped = capped;
		stats->leap_attempts = counters[4];
		stats->leaps = counters[5];
		stats->groups = counters[6];
		stats->leaped_steps = counters[7];
	}
	if (capped) return noterm(capped);
	return HMRM_OK;
}

int hmrm_render(const hmrm_scene *scene, const hmrm_camera *cam, uint8_t *rgba, size_t stride_bytes) {
	return render_common(const_cast<hmrm_scene *>(scene), cam, rgba, stride_bytes, nullptr, nullptr,
	                     nullptr, false);
}

// Progressive refresh of the reference's frame driver (hmap.cpp:976-983): only pixels
// p = cycle, cycle + cycle_period, ... (p = x + y*width) of the caller's framebuffer are
// rewritten, the rest keeps what earlier frames left there.  The whole frame is rendered on
// the device (sub-millisecond) and the selected pixels are merged on the host.
int hmrm_render_cycle(const hmrm_scene *scene, const hmrm_camera *cam, uint8_t *rgba, size_t stride_bytes,
                      int32_t cycle, int32_t cycle_period) {
	if (cycle_period <= 0 || cycle < 0 || cycle >= cycle_period) return fail(HMRM_E_ARG, "need 0 <= cycle < cycle_period");
	int rc = check_camera(cam);
	if (rc) return rc;
	if (!rgba) return fail(HMRM_E_ARG, "NULL argument");
	if (stride_bytes < (size_t)cam->width * 4) return fail(HMRM_E_ARG, "stride_bytes < width*4");
	const size_t W = (size_t)cam->width, H = (size_t)cam->height;
	std::vector<uint8_t> full(W * H * 4);
	rc = hmrm_render(scene, cam, full.data(), W * 4);
	if (rc != HMRM_OK && rc != HMRM_E_NOTERM) return rc;
	for (size_t p = (size_t)cycle; p < W * H; p += (size_t)cycle_period) {
		const size_t x = p % W, y = p / W; // hmap.cpp:982-983
		memcpy(rgba + y * stride_bytes + x * 4, &full[p * 4], 4);
	}
	return rc;
}

int hmrm_render_stats(const hmrm_scene *scene, const hmrm_camera *cam, uint8_t *rgba,
                      size_t stride_bytes, hmrm_stats *stats, uint32_t *steps_per_pixel, double *entry_d) {
	return render_common(const_cast<hmrm_scene *>(scene), cam, rgba, stride_bytes, stats,
	                     steps_per_pixel, entry_d, true);
}

int hmrm_render_rows_device(const hmrm_scene *scene, const hmrm_camera *cam, void *d_rgba,
                            size_t stride_bytes, int32_t row_begin, int32_t row_end, int32_t band_rows,
                            int32_t band_index, int32_t band_count, void *hip_stream) {
	hmrm_scene *s = const_cast<hmrm_scene *>(scene);
	int rc = check_camera(cam);
	if (rc) return rc;
	if (!s || !d_rgba) return fail(HMRM_E_ARG, "NULL argument");
	if (stride_bytes < (size_t)cam->width * 4 || (stride_bytes & 3) || stride_bytes / 4 > 0x7fffffffu)
		return fail(HMRM_E_ARG, "stride_bytes must be >= width*4, a multiple of 4 and below 2^33");
	hmrm::RowMap rows{};
	if (band_rows > 0) {
		if (band_count <= 0 || band_index < 0 || band_index >= band_count)
			return fail(HMRM_E_ARG, "bad band_index/band_count");
		// bands band_index, band_index+band_count, ... packed back to back; a trailing
		// partial band still occupies band_rows rows of the strip (its tail is not written)
		const int64_t local = hmrm_band_local_rows(cam->height, band_rows, band_index, band_count);
		rows.row_begin = 0;
		rows.local_rows = (int32_t)local;
		rows.band_rows = band_rows;
		rows.band_index = band_index;
		rows.band_count = band_count;
	} else {
		if (row_begin < 0 || row_end > cam->height || row_begin > row_end)
			return fail(HMRM_E_ARG, "bad row range");
		rows.row_begin = row_begin;
		rows.local_rows = row_end - row_begin;
		rows.band_rows = 0;
		rows.band_index = 0;
		rows.band_count = 1;
	}
	HIP_TRY(hipSetDevice(s->device));
	std::lock_guard<std::mutex> lk(s->mu);
	StreamCtx *c = nullptr;
	if ((rc = ctx_for(s, (hipStream_t)hip_stream, &c))) return rc;
	hmrm::DevFrame f;
	FrameSlot *slot = nullptr;
	if ((rc = prepare_frame(s, c, cam, &f, &slot))) return rc;
	return launch_frame(s, c, f, slot, rows, (uint32_t *)d_rgba, (int64_t)(stride_bytes / 4), nullptr, nullptr, false);
}

// One frame over several scenes -- one per GPU, same maps (BASELINE config C4's sharding):
// scene i renders the cyclic 16-row bands i, i+n, ... into a strip on its own device and the strip crosses
// its own PCIe link; no exchange between devices (a gather to one GPU first would funnel every byte through
// that GPU's link).  Three passes so that no device ever waits for another one's copy:
//   1. every scene's kernel is launched (asynchronous);
//   2. every scene's device-to-host copy is enqueued behind its kernel.  A copy into PAGEABLE host memory is staged
//      by the runtime and blocks the calling thread until it is done -- device i+1's copy would start after device
//      i's had finished -- so the copies go to the caller's frame directly only when that memory is pinned (the
//      caller registered it, hipHostRegister, or got it from hipHostMalloc), else to a pinned staging strip the
//      scene owns;
//   3. scenes are waited for in order and staged strips are copied into the caller's rows by the host pool while
//      the later devices' copies are still in flight.
int hmrm_render_multi(hmrm_scene *const *scenes, int32_t n_scenes, const hmrm_camera *cam, uint8_t *rgba,
                      size_t stride_bytes) {
	int rc = check_camera(cam);
	if (rc) return rc;
	if (!scenes || n_scenes <= 0 || !rgba) return fail(HMRM_E_ARG, "NULL argument");
	const size_t W = (size_t)cam->width;
	const int H = cam->height;
	if (stride_bytes < W * 4) return fail(HMRM_E_ARG, "stride_bytes < width*4");
	constexpr int kBand = 16;
	const int n = std::min<int>(n_scenes, (H + kBand - 1) / kBand);
	for (int i = 0; i < n; ++i)
		if (!scenes[i]) return fail(HMRM_E_ARG, "NULL scene");
	// is the caller's frame pinned?  (an address the runtime does not know is ordinary pageable memory: the query
	// fails with hipErrorInvalidValue, which is not an error of this call)
	bool pinned_dst = false;
	{
		// (first AND last byte: a frame only partly inside a registered block would send the later bands' copies
		// through the runtime's staging, which blocks the calling thread per band)
		hipPointerAttribute_t attr{}, attr_end{};
		const uint8_t *last = rgba + (size_t)(H - 1) * stride_bytes + W * 4 - 1;
		if (hipPointerGetAttributes(&attr, rgba) == hipSuccess && hipPointerGetAttributes(&attr_end, last) == hipSuccess)
			pinned_dst = attr.type == hipMemoryTypeHost && attr_end.type == hipMemoryTypeHost;
		else (void)hipGetLastError();
	}
	int launched = 0; // scenes with work in flight: drained before an error return (the caller may free `rgba` at once)
	auto drain = [&](int rc_keep) -> int {
		const std::string keep = g_error;
		for (int j = 0; j < launched; ++j)
			if (hipSetDevice(scenes[j]->device) == hipSuccess) (void)hipStreamSynchronize(scenes[j]->stream);
		g_error = keep;
		return rc_keep;
	};
	auto launch = [&](int i) -> int {
		hmrm_scene *s = scenes[i];
		HIP_TRY(hipSetDevice(s->device));
		std::lock_guard<std::mutex> lk(s->mu);
		const int32_t local = hmrm_band_local_rows(H, kBand, i, n);
		int rc2;
		if ((rc2 = ensure_frame(s, W * (size_t)local))) return rc2;
		if (!pinned_dst && s->h_stage_bytes < W * 4 * (size_t)local) {
			if (s->h_stage) (void)hipHostFree(s->h_stage);
			s->h_stage = nullptr;
			s->h_stage_bytes = 0;
			HIP_TRY(hipHostMalloc((void **)&s->h_stage, W * 4 * (size_t)local, hipHostMallocDefault));
			s->h_stage_bytes = W * 4 * (size_t)local;
		}
		StreamCtx *c = nullptr;
		if ((rc2 = ctx_for(s, s->stream, &c))) return rc2;
		hmrm::DevFrame f;
		FrameSlot *slot = nullptr;
		if ((rc2 = prepare_frame(s, c, cam, &f, &slot))) return rc2;
		hmrm::RowMap rows{0, local, kBand, i, n, {}, {}, nullptr};
		return launch_frame(s, c, f, slot, rows, s->d_frame, (int64_t)W, nullptr, nullptr, false);
	};
	auto enqueue_copy = [&](int i) -> int {
		hmrm_scene *s = scenes[i];
		HIP_TRY(hipSetDevice(s->device));
		const int32_t local = hmrm_band_local_rows(H, kBand, i, n);
		if (!pinned_dst) { // the whole strip in one transfer
			HIP_TRY(hipMemcpyAsync(s->h_stage, s->d_frame, W * 4 * (size_t)local, hipMemcpyDeviceToHost, s->stream));
			return HMRM_OK;
		}
		// band b of this scene's strip is frame rows [(i + b*n) * kBand, ...): contiguous in both
		for (int b = 0; (i + b * n) * kBand < H; ++b) {
			const int row0 = (i + b * n) * kBand, nrows = std::min(kBand, H - row0);
			HIP_TRY(hipMemcpy2DAsync(rgba + (size_t)row0 * stride_bytes, stride_bytes,
			                         s->d_frame + (size_t)b * kBand * W, W * 4, W * 4, (size_t)nrows,
			                         hipMemcpyDeviceToHost, s->stream));
		}
		return HMRM_OK;
	};
	for (int i = 0; i < n; ++i) {
		rc = launch(i);
		launched = i + 1; // (a failed launch may still have enqueued a table upload)
		if (rc != HMRM_OK) return drain(rc);
	}
	for (int i = 0; i < n; ++i)
		if ((rc = enqueue_copy(i)) != HMRM_OK) return drain(rc);
	unsigned long long capped = 0;
	for (int i = 0; i < n; ++i) {
		hmrm_scene *s = scenes[i];
		hipError_t e = hipSetDevice(s->device);
		if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
		if (e != hipSuccess) return drain(fail(HMRM_E_DEVICE, std::string("hmrm_render_multi: ") + hipGetErrorString(e)));
		if (!pinned_dst) {
			const int bands = (H - i * kBand + n * kBand - 1) / (n * kBand); // bands i, i+n, .. below H
			const uint8_t *src = s->h_stage;
			hmrm::parallel_ranges(bands, 4, [&](int b0, int b1) {
				for (int b = b0; b < b1; ++b) {
					const int row0 = (i + b * n) * kBand, nrows = std::min(kBand, H - row0);
					for (int r = 0; r < nrows; ++r)
						memcpy(rgba + (size_t)(row0 + r) * stride_bytes, src + ((size_t)b * kBand + (size_t)r) * W * 4, W * 4);
				}
			});
		}
		std::lock_guard<std::mutex> lk(s->mu);
		StreamCtx *c = nullptr;
		if ((rc = ctx_for(s, s->stream, &c))) return drain(rc);
		unsigned long long here = 0;
		if ((rc = take_capped(c, &here))) return drain(rc);
		capped += here;
	}
	return capped ? noterm(capped) : HMRM_OK;
}

int hmrm_scene_take_capped(const hmrm_scene *scene, void *hip_stream, uint64_t *capped) {
	hmrm_scene *s = const_cast<hmrm_scene *>(scene);
	if (!s || !capped) return fail(HMRM_E_ARG, "NULL argument");
	HIP_TRY(hipSetDevice(s->device));
	std::lock_guard<std::mutex> lk(s->mu);
	*capped = 0;
	for (StreamCtx *c : s->ctxs)
		if (c->stream == (hipStream_t)hip_stream) {
			HIP_TRY(hipStreamSynchronize(c->stream));
			unsigned long long n = 0;
			const int rc = take_capped(c, &n);
			if (rc) return rc;
			*capped = n;
			return n ? noterm(n) : HMRM_OK;
		}
	return HMRM_OK; // nothing was ever launched on that stream
}

// ---- asynchronous frames: kernel k+1 runs while frame k crosses PCIe (the reference's counterpart is
// the blit of the finished framebuffer, SDL_UpdateTexture, hmap.cpp:1082) ----
int hmrm_render_begin(const hmrm_scene *scene, const hmrm_camera *cam, int32_t *ticket) {
	return hmrm_render_begin_flags(scene, cam, 0u, ticket);
}

int hmrm_render_begin_flags(const hmrm_scene *scene, const hmrm_camera *cam, uint32_t flags, int32_t *ticket) {
	hmrm_scene *s = const_cast<hmrm_scene *>(scene);
	int rc = check_camera(cam);
	if (rc) return rc;
	if (!s || !ticket) return fail(HMRM_E_ARG, "NULL argument");
	*ticket = -1;
	const size_t W = (size_t)cam->width, H = (size_t)cam->height;
	HIP_TRY(hipSetDevice(s->device));
	std::lock_guard<std::mutex> lk(s->mu);
	int idx = -1;
	for (size_t i = 0; i < s->ring.size(); ++i)
		if (!s->ring[i]->busy) {
			idx = (int)i;
			break;
		}
	if (idx < 0) {
		if ((int)s->ring.size() >= kMaxRing) return fail(HMRM_E_ARG, "hmrm_render_begin: every frame of the ring is in use (release one)");
		RingFrame *r = new (std::nothrow) RingFrame();
		if (!r) return fail(HMRM_E_ARG, "out of memory");
		// (complete before it joins the ring: a half-made slot would be picked up as free by the next call)
		hipError_t e = hipEventCreateWithFlags(&r->kernel_done, hipEventDisableTiming);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&r->copy_done, hipEventDisableTiming);
		if (e == hipSuccess) e = hipHostMalloc((void **)&r->h_capped, sizeof(unsigned long long), hipHostMallocDefault);
		if (e != hipSuccess) {
			if (r->kernel_done) (void)hipEventDestroy(r->kernel_done);
			if (r->copy_done) (void)hipEventDestroy(r->copy_done);
			if (r->h_capped) (void)hipHostFree(r->h_capped);
			delete r;
			return fail(HMRM_E_DEVICE, std::string("hmrm_render_begin: ") + hipGetErrorString(e));
		}
		s->ring.push_back(r);
		idx = (int)s->ring.size() - 1;
	}
	RingFrame *r = s->ring[(size_t)idx];
	if (W * H > r->px) {
		// (the slot is free: nothing in flight touches its buffers)
		if (r->d_frame) (void)hipFree(r->d_frame);
		if (r->h_frame) (void)hipHostFree(r->h_frame);
		r->d_frame = nullptr;
		r->h_frame = nullptr;
		r->px = 0;
		HIP_TRY(hipMalloc((void **)&r->d_frame, W * H * sizeof(uint32_t)));
		HIP_TRY(hipHostMalloc((void **)&r->h_frame, W * H * 4, hipHostMallocDefault));
		r->px = W * H;
	}
	StreamCtx *c = nullptr;
	if ((rc = next_lane(s, &c))) return rc;
	hmrm::DevFrame f;
	FrameSlot *slot = nullptr;
	if ((rc = prepare_frame(s, c, cam, &f, &slot))) return rc;
	hmrm::RowMap rows{0, cam->height, 0, 0, 1, {}, {}, nullptr};
	if ((rc = launch_frame(s, c, f, slot, rows, r->d_frame, (int64_t)W, nullptr, nullptr, false, (flags & HMRM_NO_PROBE) != 0))) return rc;
	r->ctx = c;
	HIP_TRY(hipEventRecord(r->kernel_done, c->stream));
	HIP_TRY(hipStreamWaitEvent(s->copy_stream, r->kernel_done, 0));
	HIP_TRY(hipMemcpyAsync(r->h_frame, r->d_frame, W * H * 4, hipMemcpyDeviceToHost, s->copy_stream));
	HIP_TRY(hipMemcpyAsync(r->h_capped, c->d_counters + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost,
	                       s->copy_stream));
	HIP_TRY(hipEventRecord(r->copy_done, s->copy_stream));
	r->width = cam->width;
	r->height = cam->height;
	r->busy = true;
	*ticket = idx;
	return HMRM_OK;
}

int hmrm_render_wait(const hmrm_scene *scene, int32_t ticket, const uint8_t **rgba, size_t *stride_bytes) {
	hmrm_scene *s = const_cast<hmrm_scene *>(scene);
	if (!s || !rgba) return fail(HMRM_E_ARG, "NULL argument");
	RingFrame *r = nullptr;
	{
		std::lock_guard<std::mutex> lk(s->mu);
		if (ticket < 0 || ticket >= (int)s->ring.size() || !s->ring[(size_t)ticket]->busy)
			return fail(HMRM_E_ARG, "hmrm_render_wait: no such frame in flight");
		r = s->ring[(size_t)ticket];
	}
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipEventSynchronize(r->copy_done)); // (outside the lock: other threads may begin frames meanwhile)
	*rgba = r->h_frame;
	if (stride_bytes) *stride_bytes = (size_t)r->width * 4;
	std::lock_guard<std::mutex> lk(s->mu);
	StreamCtx *c = r->ctx;
	// the counter is cumulative over the lane's launches: later frames may already be in it, so a
	// capped ray is reported with the first frame of the lane waited for after it happened
	if (*r->h_capped > c->capped_seen) {
		const unsigned long long n = *r->h_capped - c->capped_seen;
		c->capped_seen = *r->h_capped;
		return noterm(n);
	}
	return HMRM_OK;
}

void hmrm_render_release(const hmrm_scene *scene, int32_t ticket) {
	hmrm_scene *s = const_cast<hmrm_scene *>(scene);
	if (!s) return;
	std::lock_guard<std::mutex> lk(s->mu);
	if (ticket < 0 || ticket >= (int)s->ring.size()) return;
	RingFrame *r = s->ring[(size_t)ticket];
	if (!r->busy) return;
	(void)hipSetDevice(s->device);
	(void)hipEventSynchronize(r->copy_done); // never hand a slot back while the copy engine writes it
	r->busy = false;
}

// ---- frames into device memory through the launch lanes ----
int hmrm_render_device_begin(const hmrm_scene *scene, const hmrm_camera *cam, void *d_rgba, size_t stride_bytes, int32_t *ticket) {
	return hmrm_render_device_begin_flags(scene, cam, d_rgba, stride_bytes, 0u, ticket);
}

int hmrm_render_device_begin_flags(const hmrm_scene *scene, const hmrm_camera *cam, void *d_rgba, size_t stride_bytes, uint32_t flags,
                                   int32_t *ticket) {
	hmrm_scene *s = const_cast<hmrm_scene *>(scene);
	int rc = check_camera(cam);
	if (rc) return rc;
	if (!s || !d_rgba || !ticket) return fail(HMRM_E_ARG, "NULL argument");
	*ticket = -1;
	if (stride_bytes < (size_t)cam->width * 4 || (stride_bytes & 3) || stride_bytes / 4 > 0x7fffffffu)
		return fail(HMRM_E_ARG, "stride_bytes must be >= width*4, a multiple of 4 and below 2^33");
	HIP_TRY(hipSetDevice(s->device));
	std::lock_guard<std::mutex> lk(s->mu);
	int idx = -1;
	for (size_t i = 0; i < s->dev_tickets.size(); ++i)
		if (!s->dev_tickets[i]->busy) {
			idx = (int)i;
			break;
		}
	if (idx < 0) {
		if ((int)s->dev_tickets.size() >= kMaxRing) return fail(HMRM_E_ARG, "hmrm_render_device_begin: 64 frames in flight (wait for one)");
		DevTicket *t = new (std::nothrow) DevTicket();
		if (!t) return fail(HMRM_E_ARG, "out of memory");
		hipError_t e = hipEventCreateWithFlags(&t->done, hipEventDisableTiming);
		if (e == hipSuccess) e = hipHostMalloc((void **)&t->h_capped, sizeof(unsigned long long), hipHostMallocDefault);
		if (e != hipSuccess) {
			if (t->done) (void)hipEventDestroy(t->done);
			delete t;
			return fail(HMRM_E_DEVICE, std::string("hmrm_render_device_begin: ") + hipGetErrorString(e));
		}
		s->dev_tickets.push_back(t);
		idx = (int)s->dev_tickets.size() - 1;
	}
	DevTicket *t = s->dev_tickets[(size_t)idx];
	StreamCtx *c = nullptr;
	if ((rc = next_lane(s, &c))) return rc;
	hmrm::DevFrame f;
	FrameSlot *slot = nullptr;
	if ((rc = prepare_frame(s, c, cam, &f, &slot))) return rc;
	hmrm::RowMap rows{0, cam->height, 0, 0, 1, {}, {}, nullptr};
	if ((rc = launch_frame(s, c, f, slot, rows, (uint32_t *)d_rgba, (int64_t)(stride_bytes / 4), nullptr, nullptr, false, (flags & HMRM_NO_PROBE) != 0))) return rc;
	HIP_TRY(hipMemcpyAsync(t->h_capped, c->d_counters + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipEventRecord(t->done, c->stream));
	t->ctx = c;
	t->busy = true;
	*ticket = idx;
	return HMRM_OK;
}

int hmrm_render_device_wait(const hmrm_scene *scene, int32_t ticket) {
	hmrm_scene *s = const_cast<hmrm_scene *>(scene);
	if (!s) return fail(HMRM_E_ARG, "NULL argument");
	DevTicket *t = nullptr;
	{
		std::lock_guard<std::mutex> lk(s->mu);
		if (ticket < 0 || ticket >= (int)s->dev_tickets.size() || !s->dev_tickets[(size_t)ticket]->busy)
			return fail(HMRM_E_ARG, "hmrm_render_device_wait: no such frame in flight");
		t = s->dev_tickets[(size_t)ticket];
	}
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipEventSynchronize(t->done)); // (outside the lock: other threads may begin frames meanwhile)
	std::lock_guard<std::mutex> lk(s->mu);
	t->busy = false;
	StreamCtx *c = t->ctx;
	if (*t->h_capped > c->capped_seen) {
		const unsigned long long n = *t->h_capped - c->capped_seen;
		c->capped_seen = *t->h_capped;
		return noterm(n);
	}
	return HMRM_OK;
}

double hmrm_last_kernel_ms(void) { return g_last_kernel_ms; }

double hmrm_bench_kernel_ms(const hmrm_scene *scene, const hmrm_camera *cam, int32_t iters) {
	hmrm_scene *s = const_cast<hmrm_scene *>(scene);
	if (check_camera(cam) || !s || iters <= 0) {
		if (g_error.empty()) g_error = "bad argument";
		return -1.0;
	}
	auto body = [&]() -> int {
		const size_t W = (size_t)cam->width, H = (size_t)cam->height;
		HIP_TRY(hipSetDevice(s->device));
		std::lock_guard<std::mutex> lk(s->mu);
		int rc = ensure_frame(s, W * H);
		if (rc) return rc;
		StreamCtx *c = nullptr;
		if ((rc = ctx_for(s, s->stream, &c))) return rc;
		hmrm::DevFrame f;
		FrameSlot *slot = nullptr;
		if ((rc = prepare_frame(s, c, cam, &f, &slot))) return rc;
		hmrm::RowMap rows{0, cam->height, 0, 0, 1, {}, {}, nullptr};
		HIP_TRY(hipEventRecord(s->ev0, s->stream));
		for (int i = 0; i < iters; ++i)
			if ((rc = launch_frame(s, c, f, slot, rows, s->d_frame, (int64_t)W, nullptr, nullptr, false))) return rc;
		HIP_TRY(hipEventRecord(s->ev1, s->stream));
		HIP_TRY(hipStreamSynchronize(s->stream));
		float ms = 0.f;
		HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
		g_last_kernel_ms = ms / iters;
		return HMRM_OK;
	};
	if (body() != HMRM_OK) return -1.0;
	return g_last_kernel_ms;
}

// ------------------------------------------------------------------ config --
hmrm_config *hmrm_config_create(void) { return new (std::nothrow) hmrm_config(); }
void hmrm_config_destroy(hmrm_config *c) { delete c; }

int hmrm_config_consume_file(hmrm_config *c, const char *path) {
	if (!c || !path) return fail(HMRM_E_ARG, "NULL argument");
	std::ifstream in(path);
	if (!in.is_open()) return fail(HMRM_E_IO, std::string("Failed to open input file: ") + path); // hmap.cpp:537-540
	std::string fatal;
	if (!c->cfg.consume(in, &fatal)) {
		bool img = fatal.compare(0, 14, "Failed to load") == 0;
		return fail(img ? HMRM_E_IMAGE : HMRM_E_CONFIG, fatal);
	}
	return HMRM_OK;
}

int hmrm_config_consume_string(hmrm_config *c, const char *text) {
	if (!c || !text) return fail(HMRM_E_ARG, "NULL argument");
	std::istringstream in(text);
	std::string fatal;
	if (!c->cfg.consume(in, &fatal)) {
		bool img = fatal.compare(0, 14, "Failed to load") == 0;
		return fail(img ? HMRM_E_IMAGE : HMRM_E_CONFIG, fatal);
	}
	return HMRM_OK;
}

const char *hmrm_config_log(const hmrm_config *c) {
	hmrm_config *m = const_cast<hmrm_config *>(c);
	m->log_cache = m->cfg.log.str();
	return m->log_cache.c_str();
}
const char *hmrm_config_warnings(const hmrm_config *c) {
	hmrm_config *m = const_cast<hmrm_config *>(c);
	m->warn_cache = m->cfg.warn.str();
	return m->warn_cache.c_str();
}

void hmrm_config_get_camera(const hmrm_config *c, hmrm_camera *out) {
	const hmrm::Config &g = c->cfg;
	memset(out, 0, sizeof *out);
	out->width = g.screen_width;
	out->height = g.screen_height;
	out->projection = g.image_plane;
	out->bg_r = g.bg_r;
	out->bg_g = g.bg_g;
	out->bg_b = g.bg_b;
	out->sampling = (uint8_t)g.sampling;
	out->hfov = g.hfov;
	out->hang = g.hang;
	out->vang = g.vang;
	out->pos[0] = g.cam_pos[0];
	out->pos[1] = g.cam_pos[1];
	out->pos[2] = g.cam_pos[2];
	out->ortho_width = g.ortho_width;
	out->step_dist = g.step_dist;
}

void hmrm_config_get_scene_params(const hmrm_config *c, hmrm_scene_params *out) {
	const hmrm::Config &g = c->cfg;
	out->min_height = g.min_height;
	out->max_height = g.max_height;
	out->lum_r = g.lum_r;
	out->lum_g = g.lum_g;
	out->lum_b = g.lum_b;
	out->grid_width = g.grid_width;
}

int32_t hmrm_config_cycle(const hmrm_config *c) { return c->cfg.cycle_period; }
int32_t hmrm_config_recording_frame_count(const hmrm_config *c) { return c->cfg.recording_frame_count; }
const char *hmrm_config_heightmap_path(const hmrm_config *c) { return c->cfg.heightmap_path.c_str(); }
const char *hmrm_config_colormap_path(const hmrm_config *c) { return c->cfg.colormap_path.c_str(); }
const char *hmrm_config_output_path(const hmrm_config *c) { return c->cfg.output_path.c_str(); }
int32_t hmrm_config_record_mode(const hmrm_config *c) { return c->cfg.record_mode; }
int32_t hmrm_config_devices(const hmrm_config *c) { return c->cfg.devices; }

const uint8_t *hmrm_config_height_rgb(const hmrm_config *c, int32_t *w, int32_t *h) {
	if (!c->cfg.have_heightmap) return nullptr;
	if (w) *w = c->cfg.heightmap.w;
	if (h) *h = c->cfg.heightmap.h;
	return c->cfg.heightmap.px.data();
}
const uint8_t *hmrm_config_color_rgba(const hmrm_config *c, int32_t *w, int32_t *h) {
	if (!c->cfg.have_colormap) return nullptr;
	if (w) *w = c->cfg.colormap.w;
	if (h) *h = c->cfg.colormap.h;
	return c->cfg.colormap.px.data();
}
int hmrm_config_take_heightmap_dirty(hmrm_config *c) {
	int d = c->cfg.heightmap_dirty ? 1 : 0;
	c->cfg.heightmap_dirty = false;
	return d;
}
int hmrm_config_create_scene(const hmrm_config *c, hmrm_scene **out) {
	if (!c || !out) return fail(HMRM_E_ARG, "NULL argument");
	if (!c->cfg.have_heightmap) return fail(HMRM_E_CONFIG, "Must specify heightmap in config");
	if (!c->cfg.have_colormap) return fail(HMRM_E_CONFIG, "Must specify colormap in config");
	hmrm_scene_params p;
	hmrm_config_get_scene_params(c, &p);
	return hmrm_scene_create(c->cfg.heightmap.px.data(), c->cfg.colormap.px.data(), c->cfg.heightmap.w,
	                         c->cfg.heightmap.h, &p, out);
}

// ---------------------------------------------------------------- image IO --
static int hand_out(hmrm::Image &img, uint8_t **out, int32_t *w, int32_t *h, int32_t *n) {
	uint8_t *p = (uint8_t *)malloc(img.px.size() ? img.px.size() : 1);
	if (!p) return fail(HMRM_E_ARG, "out of memory");
	memcpy(p, img.px.data(), img.px.size());
	*out = p;
	if (w) *w = img.w;
	if (h) *h = img.h;
	if (n) *n = img.comp_in_file;
	return HMRM_OK;
}

int hmrm_image_load(const char *path, int32_t req_comp, uint8_t **out, int32_t *w, int32_t *h,
                    int32_t *comp_in_file) {
	if (!path || !out) return fail(HMRM_E_ARG, "NULL argument");
	hmrm::Image img;
	std::string err;
	if (!hmrm::load_image_file(path, req_comp, &img, &err))
		return fail(err == "can't fopen" ? HMRM_E_IO : HMRM_E_IMAGE, err);
	return hand_out(img, out, w, h, comp_in_file);
}

int hmrm_image_load_memory(const uint8_t *bytes, size_t len, int32_t req_comp, uint8_t **out,
                           int32_t *w, int32_t *h, int32_t *comp_in_file) {
	if (!bytes || !out) return fail(HMRM_E_ARG, "NULL argument");
	hmrm::Image img;
	std::string err;
	if (!hmrm::decode_image(bytes, len, req_comp, &img, &err)) return fail(HMRM_E_IMAGE, err);
	return hand_out(img, out, w, h, comp_in_file);
}

void hmrm_image_free(void *p) { free(p); }

int hmrm_write_png_memory(int32_t w, int32_t h, int32_t comp, const uint8_t *data, size_t stride_bytes,
                          uint8_t **out, size_t *out_len) {
	if (!out || !out_len) return fail(HMRM_E_ARG, "NULL argument");
	std::vector<uint8_t> png;
	if (!hmrm::encode_png(w, h, comp, data, stride_bytes, &png)) return fail(HMRM_E_ARG, "bad image arguments");
	uint8_t *p = (uint8_t *)malloc(png.size());
	if (!p) return fail(HMRM_E_ARG, "out of memory");
	memcpy(p, png.data(), png.size());
	*out = p;
	*out_len = png.size();
	return HMRM_OK;
}

int hmrm_write_png(const char *path, int32_t w, int32_t h, int32_t comp, const uint8_t *data,
                   size_t stride_bytes) {
	if (!path) return fail(HMRM_E_ARG, "NULL argument");
	std::vector<uint8_t> png;
	if (!hmrm::encode_png(w, h, comp, data, stride_bytes, &png)) return fail(HMRM_E_ARG, "bad image arguments");
	if (!hmrm::write_file(path, png.data(), png.size()))
		return fail(HMRM_E_IO, std::string("Failed to write screenshot to ") + path); // hmap.cpp:162-164
	return HMRM_OK;
}

int hmrm_write_ppm(const char *path, int32_t w, int32_t h, int32_t comp, const uint8_t *data,
                   size_t stride_bytes) {
	if (!path) return fail(HMRM_E_ARG, "NULL argument");
	std::vector<uint8_t> pnm;
	if (!hmrm::encode_pnm(w, h, comp, data, stride_bytes, &pnm)) return fail(HMRM_E_ARG, "bad image arguments");
	if (!hmrm::write_file(path, pnm.data(), pnm.size()))
		return fail(HMRM_E_IO, std::string("Failed to write image to ") + path);
	return HMRM_OK;
}

// Device-side GetRay + distance() for one pixel (test hook).
int hmrm_debug_ray(const hmrm_scene *scene, const hmrm_camera *cam, int32_t px, int32_t py,
                   double pos[3], double dir[3], double *entry_d) {
	hmrm_scene *s = const_cast<hmrm_scene *>(scene);
	int rc = check_camera(cam);
	if (rc) return rc;
	if (!s || !pos || !dir || !entry_d) return fail(HMRM_E_ARG, "NULL argument");
	if (px < 0 || py < 0 || px >= cam->width || py >= cam->height) return fail(HMRM_E_ARG, "pixel out of range");
	HIP_TRY(hipSetDevice(s->device));
	std::lock_guard<std::mutex> lk(s->mu);
	if ((rc = ensure_stats(s, 8))) return rc;
	StreamCtx *c = nullptr;
	if ((rc = ctx_for(s, s->stream, &c))) return rc;
	hmrm::DevFrame f;
	if ((rc = prepare_frame(s, c, cam, &f, nullptr))) return rc;
	HIP_TRY(hmrm::launch_probe(f, px, py, s->d_entry, s->stream));
	double host[7];
	HIP_TRY(hipMemcpyAsync(host, s->d_entry, sizeof host, hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	for (int i = 0; i < 3; ++i) { pos[i] = host[i]; dir[i] = host[3 + i]; }
	*entry_d = host[6];
	return HMRM_OK;
}

// Host-only test hook for the launch-order calibration: the plan api.cpp derives from the records of a measured launch
// (tile_rows x {start of the row's first workgroup, longest wave of the row}, 10 ns ticks, measured under the plain
// rotation by `rot`), and the grid-row -> tile-row map the kernel would apply for it.
int hmrm_debug_plan_order(const uint64_t *records, int32_t tile_rows, int32_t rot, int32_t pieces_begin[3],
                          int32_t pieces_count[3], int32_t *tile_row_of_grid_row) {
	if (!records || !pieces_begin || !pieces_count || tile_rows <= 0 || rot < 0 || rot >= tile_rows) return fail(HMRM_E_ARG, "bad argument");
	int b[3] = {0, 0, 0}, c[3] = {0, 0, 0};
	const int n = hmrm::plan_order_from_measurement((const unsigned long long *)records, tile_rows, rot, b, c);
	for (int k = 0; k < 3; ++k) {
		pieces_begin[k] = k < n ? b[k] : 0;
		pieces_count[k] = k < n ? c[k] : 0;
	}
	if (tile_row_of_grid_row) {
		hmrm::RowMap r{};
		hmrm::set_tile_order(&r, tile_rows, rot, n, b, c);
		for (int gy = 0; gy < tile_rows; ++gy) { // device_common.hpp pixel_of_lane, restated
			int delta = r.seg_delta[0];
			for (int k = 1; k < hmrm::kOrderSegs; ++k) delta = (unsigned)gy >= (unsigned)r.seg_first[k - 1] ? r.seg_delta[k] : delta;
			unsigned ty = (unsigned)gy + (unsigned)delta;
			if (ty >= (unsigned)tile_rows) ty -= (unsigned)tile_rows;
			tile_row_of_grid_row[gy] = (int32_t)ty;
		}
	}
	return n;
}

// v_rcp_f64 accuracy on this device (render.hip k_rcp_error): the premise of slab_classify's margins.
int hmrm_debug_rcp_error(int32_t mode, uint64_t count, uint64_t seed, int32_t exp_lo, int32_t exp_hi,
                         double *max_rel_err, uint64_t *hist64) {
	if (!max_rel_err) return fail(HMRM_E_ARG, "NULL argument");
	if (mode < 0 || mode > 2 || count == 0 || count > ((uint64_t)1 << 40) || exp_lo > exp_hi || exp_lo < -1000 || exp_hi > 1000)
		return fail(HMRM_E_ARG, "hmrm_debug_rcp_error: bad mode, count or exponent range");
	unsigned long long *d = nullptr, host[65] = {};
	HIP_TRY(hipMalloc((void **)&d, sizeof host));
	hipError_t e = hipMemset(d, 0, sizeof host);
	if (e == hipSuccess) e = hmrm::launch_rcp_error(mode, count, seed, exp_lo, exp_hi, d, nullptr);
	if (e == hipSuccess) e = hipMemcpy(host, d, sizeof host, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(HMRM_E_DEVICE, std::string("rcp probe: ") + hipGetErrorString(e));
	memcpy(max_rel_err, &host[0], sizeof(double));
	if (hist64)
		for (int i = 0; i < 64; ++i) hist64[i] = host[1 + i];
	return HMRM_OK;
}

int hmrm_debug_frame(const hmrm_camera *cam, const hmrm_scene_params *params, int32_t map_w,
                     int32_t map_h, double *out25, double *tables) {
	int rc = check_camera(cam);
	if (rc) return rc;
	if (!params || !out25) return fail(HMRM_E_ARG, "NULL argument");
	if (cam->projection == HMRM_SPHERICAL && !tables) return fail(HMRM_E_ARG, "tables required for spherical");
	hmrm::HostCamera hc{};
	hc.width = cam->width; hc.height = cam->height; hc.projection = cam->projection;
	hc.bg_r = cam->bg_r; hc.bg_g = cam->bg_g; hc.bg_b = cam->bg_b; hc.sampling = cam->sampling;
	hc.hfov = cam->hfov; hc.hang = cam->hang; hc.vang = cam->vang;
	hc.pos[0] = cam->pos[0]; hc.pos[1] = cam->pos[1]; hc.pos[2] = cam->pos[2];
	hc.ortho_width = cam->ortho_width; hc.step_dist = cam->step_dist;
	const size_t W = (size_t)cam->width, H = (size_t)cam->height;
	hmrm::DevFrame f;
	hmrm::build_frame(hc, map_w, map_h, params->min_height, params->max_height, params->grid_width, &f, nullptr, nullptr,
	                  nullptr, nullptr);
	if (cam->projection == HMRM_SPHERICAL) {
		// the tables exactly as a launch fills them: in pieces, on the host pool (prepare_frame)
		double *cc = tables, *cs = tables + W, *rs = tables + 2 * W, *rc2 = tables + 2 * W + H;
		const int nc = cam->width, nr = cam->height;
		hmrm::parallel_ranges(nc + nr, 1024, [&](int b, int e) {
			if (b < nc) hmrm::fill_col_tables(hc, b, std::min(e, nc), cc, cs);
			if (e > nc) hmrm::fill_row_tables(hc, std::max(b, nc) - nc, e - nc, rs, rc2);
		});
	}
	double *o = out25;
	for (int i = 0; i < 3; ++i) *o++ = f.cam[i];
	for (int i = 0; i < 3; ++i) *o++ = f.upper_left[i];
	for (int i = 0; i < 3; ++i) *o++ = f.plane_right[i];
	for (int i = 0; i < 3; ++i) *o++ = f.plane_down[i];
	for (int i = 0; i < 3; ++i) *o++ = f.look[i];
	for (int i = 0; i < 3; ++i) *o++ = f.c0[i];
	for (int i = 0; i < 3; ++i) *o++ = f.c1[i];
	*o++ = f.nudge;
	*o++ = f.step_dist;
	*o++ = (double)f.grid_pow2;
	*o++ = f.inv_grid_width;
	return HMRM_OK;
}

// Test hook (no GPU): the pyramid layout hmrm_scene_create would choose for a map, and whether 32-bit BYTE offsets
// would cover it (0: only the kernel's 64-bit offsets do -- informational since round 5).
int hmrm_debug_mip_layout(int32_t map_w, int32_t map_h, int32_t *mip_row, int32_t *plane_shift, int32_t *levels) {
	if (map_w <= 0 || map_h <= 0 || (int64_t)map_w * map_h > ((int64_t)1 << 31) / 4) return fail(HMRM_E_ARG, "bad map dimensions");
	int32_t w[hmrm::kMipLevels], h[hmrm::kMipLevels], row = 0, shift = 0;
	const bool fits = mip_layout(map_w, map_h, w, h, &row, &shift);
	if (mip_row) *mip_row = row;
	if (plane_shift) *plane_shift = shift;
	if (levels) *levels = hmrm::kMipLevels;
	return fits ? 1 : 0;
}

// Test hook (no GPU): launch_order.hpp pick_fast_kernel for a scene whose probe state is (scene_verdict, scene_with_records).
int hmrm_debug_pick_kernel(int32_t forced, int32_t use_other, int32_t records_ok, int32_t scene_verdict, int32_t scene_with_records,
                           int32_t *with_records_after) {
	hmrm::KernelChoice choice;
	choice.use_group = scene_verdict != 0;
	choice.probed = choice.use_group;
	choice.with_records = scene_with_records != 0;
	const int k = hmrm::pick_fast_kernel(forced, use_other != 0, records_ok != 0, choice);
	if (with_records_after) *with_records_after = choice.with_records ? 1 : 0;
	return k;
}

// Test hook: the scene's window records (frame.hpp WindowRecord, 32 bytes each, rec_row(map_w) x ceil(map_h / 4)) and the
// threshold table they were built from (map_w x map_h doubles), copied to the host.  Either pointer may be NULL.
int hmrm_debug_read_records(const hmrm_scene *cs, void *records_out, double *thr_out) {
	hmrm_scene *s = const_cast<hmrm_scene *>(cs);
	if (!s) return fail(HMRM_E_ARG, "NULL argument");
	std::lock_guard<std::mutex> lk(s->mu);
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	if (records_out) {
		if (!s->d_records) return fail(HMRM_E_ARG, "this scene has no window records (map too tall)");
		const int rc_r = ensure_records(s);
		if (rc_r) return rc_r;
		HIP_TRY(hipMemcpy(records_out, s->d_records, (size_t)hmrm::rec_row(s->map_w) * (size_t)((s->map_h + 3) / 4) * sizeof(hmrm::WindowRecord),
		                  hipMemcpyDeviceToHost));
	}
	if (thr_out) HIP_TRY(hipMemcpy(thr_out, s->d_thr, (size_t)s->map_w * (size_t)s->map_h * sizeof(double), hipMemcpyDeviceToHost));
	return HMRM_OK;
}

// Test hook (no GPU): the calibration's state machine (launch_order.hpp OrderCalibration / KernelChoice) driven through a
// sequence of full-frame launches of one camera; the measurement of a launch arrives before the next launch.
int hmrm_debug_calibrate(const uint64_t *records, int32_t launches, int32_t tile_rows, int32_t rot, int32_t may_probe,
                         int32_t scene_already_probed, const uint8_t *can_measure, int32_t *trial_used, int32_t *measured,
                         int32_t *group_kernel, int32_t *n_trials, int32_t *best, int32_t *scene_use_group, int32_t *settled_at_launch) {
	if (!records || launches <= 0 || tile_rows <= 0 || tile_rows > kMaxMeasRows || rot < 0 || rot >= tile_rows)
		return fail(HMRM_E_ARG, "hmrm_debug_calibrate: bad argument");
	hmrm::OrderCalibration cal;
	hmrm::KernelChoice choice;
	choice.probed = scene_already_probed != 0;
	int settled_at = -1;
	for (int32_t i = 0; i < launches; ++i) {
		const hmrm::LaunchPlan p = cal.plan(!can_measure || can_measure[i] != 0, choice);
		if (trial_used) trial_used[i] = p.trial;
		if (measured) measured[i] = p.measure ? 1 : 0;
		if (group_kernel) group_kernel[i] = p.use_group ? 1 : 0;
		if (p.measure && cal.on_measured((const unsigned long long *)records + (size_t)i * 2 * (size_t)tile_rows, tile_rows, rot, may_probe != 0, choice))
			settled_at = i;
	}
	if (n_trials) *n_trials = cal.n_trials;
	if (best) *best = cal.best;
	if (scene_use_group) *scene_use_group = choice.use_group ? 1 : 0;
	if (settled_at_launch) *settled_at_launch = settled_at;
	return HMRM_OK;
}

int32_t hmrm_band_local_rows(int32_t height, int32_t band_rows, int32_t band_index, int32_t band_count) {
	if (height <= 0 || band_rows <= 0 || band_count <= 0 || band_index < 0 || band_index >= band_count) return 0;
	int64_t local = 0;
	for (int64_t b = band_index; b * band_rows < height; b += band_count) local += band_rows;
	return (int32_t)local;
}

} // extern "C"
