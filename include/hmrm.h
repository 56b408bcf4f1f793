/*
 * hmrm.h -- C ABI of libhmrm.so, the MI355X-native (gfx950) heightmap ray
 * marcher.  This is the drop-in boundary for the reference's hot path.
 *
 * The reference (Costava/heightmap-ray-marcher) has no FFI/plugin boundary: the
 * path is inline in main() and talks through file-scope globals
 * (main/hmap.cpp:28-112).  The contract a replacement has to honour is
 *     config text in  ->  RGBA8 framebuffer / PNG out.
 * Each entry point below names the reference interface it replaces.  All
 * pointers are plain host pointers unless the name says "device"; no torch, HIP
 * or C++ types appear in any signature.  All functions returning int return
 * HMRM_OK (0) or a negative HMRM_E_* code; hmrm_last_error() then holds the
 * message the reference would have printed to stderr before exit(1).
 *
 * There is NO CPU fallback: entry points that render fail with HMRM_E_DEVICE
 * when no gfx950 device / HIP runtime is usable.
 *
 * Threading (the reference's loop is an OpenMP region over read-only globals, hmap.cpp:978): a scene may
 * be used from several host threads.  Everything a launch mutates -- spherical tables, counters, the
 * cache of per-frame records -- is kept per HIP stream, so hmrm_render_rows_device calls on different
 * streams run concurrently on the device (a scene keeps that state for the 32 most recently used streams;
 * driving it from more makes every launch wait for the stream whose state it takes over); the host-side
 * set-up of a call is serialised per scene.  The
 * entry points that return pixels in host memory (hmrm_render, _stats, _cycle, _multi) use the scene's own
 * stream and scratch frame: one such call at a time per scene.  hmrm_render_begin may be called while
 * other tickets are in flight.  hmrm_scene_update waits for every frame in flight on the scene's OWN streams
 * (tickets of hmrm_render_begin / hmrm_render_device_begin included: they finish with the old heights) before it
 * rewrites the tables; it and hmrm_scene_destroy require that no launch of that scene is in flight on a
 * CALLER's stream (hmrm_render_rows_device).  hmrm_last_error() is per thread.
 */
#ifndef HMRM_H
#define HMRM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMRM_ABI_VERSION 1

enum {
	HMRM_OK          =  0,
	HMRM_E_ARG       = -1,  /* bad argument / inconsistent sizes                       */
	HMRM_E_IO        = -2,  /* file cannot be opened / written (hmap.cpp:537-540,162)   */
	HMRM_E_IMAGE     = -3,  /* image cannot be decoded (hmap.cpp:324-329,345-350)       */
	HMRM_E_CONFIG    = -4,  /* config validation failed (hmap.cpp:493-515)              */
	HMRM_E_DEVICE    = -5,  /* HIP error / no device                                    */
	HMRM_E_NOTERM    = -6   /* a ray hit the step cap: the reference loop would not end */
};

/* main/hmap.cpp:104-106 IMAGEPLANE_* */
enum {
	HMRM_PERSPECTIVE  = 1,
	HMRM_SPHERICAL    = 2,
	HMRM_ORTHOGRAPHIC = 3
};

/* Height / colour sampling.  The reference samples the nearest cell with C truncation
 * (hmap.cpp:1001-1004,1013-1018): HMRM_NEAREST is the bit-exact drop-in.  HMRM_BILINEAR is a
 * build-side quality mode (north_star: "bilinear height/colour sampling"), not in the reference:
 * cell values sit at cell centres, thresholds and R,G,B are interpolated bilinearly in fp64
 * (definition: oracle/hmrm_oracle.c "bilinear quality mode"); additive config key
 * `sampling nearest|bilinear`.  HMRM_NEAREST_F32 (north_star: "float heights") keeps the reference's
 * loop and fp64 positions but compares against (float)(heightmap_buf[i] + min_height): a 4-byte
 * threshold table; a ray's hit step can move where z is within half a float ulp of the threshold
 * (tests bound it); additive config key `heights f64|f32`. */
enum {
	HMRM_NEAREST     = 0,
	HMRM_BILINEAR    = 1,
	HMRM_NEAREST_F32 = 2   /* nearest cell, hit thresholds kept as float (half the table); not parity */
};

/* The globals UpdateHeightmap() and the box corners read:
 * main/hmap.cpp:38-47 (min/max_height, lum_*), :65 (grid_width). */
typedef struct hmrm_scene_params {
	double min_height;   /* default 0.0   */
	double max_height;   /* default 10.0  */
	double lum_r;        /* default 0.299 */
	double lum_g;        /* default 0.587 */
	double lum_b;        /* default 0.114 */
	double grid_width;   /* default 0.05  */
} hmrm_scene_params;

/* The globals the per-frame set-up and the pixel loop read:
 * main/hmap.cpp:31-35 (resolution, hfov), :68 (step_dist), :75-85 (pos, hang,
 * vang), :98 (ortho_width), :107 (image_plane), :110-112 (bg).  Angles are in
 * RADIANS here (the config file holds degrees, hmap.cpp:131-133,367-384). */
typedef struct hmrm_camera {
	int32_t  width;        /* screen_width  */
	int32_t  height;       /* screen_height */
	int32_t  projection;   /* HMRM_PERSPECTIVE | HMRM_SPHERICAL | HMRM_ORTHOGRAPHIC */
	uint8_t  bg_r, bg_g, bg_b;
	uint8_t  sampling;     /* HMRM_NEAREST (0, the reference) | HMRM_BILINEAR (1) | HMRM_NEAREST_F32 (2) */
	double   hfov;
	double   hang;
	double   vang;
	double   pos[3];
	double   ortho_width;
	double   step_dist;
} hmrm_camera;

/* Per-render statistics (build-side addition; BASELINE.md metric definitions). */
typedef struct hmrm_stats {
	uint64_t rays;        /* pixels rendered                                          */
	uint64_t steps;       /* height loads the reference loop executes (hmap.cpp:1013) */
	uint64_t hits;        /* rays that end on terrain (hmap.cpp:1016)                 */
	uint64_t capped;      /* rays stopped by the step cap (reference: endless loop)   */
	/* traversal diagnostics of the production kernel (0 for HMRM_KERNEL=simple)      */
	uint64_t leap_attempts; /* pyramid look-ups tried                                 */
	uint64_t leaps;         /* ... that ended in an exact jump                        */
	uint64_t groups;        /* speculative groups executed (4 positions each; 6 in the plain-groups kernel) */
	uint64_t leaped_steps;  /* ray-steps covered by jumps (part of `steps`)           */
} hmrm_stats;

typedef struct hmrm_scene hmrm_scene;    /* device-resident height + colour maps */
typedef struct hmrm_config hmrm_config;  /* parsed config state                  */

/* ------------------------------------------------------------------ general */
int         hmrm_abi_version(void);
const char *hmrm_last_error(void);              /* thread-local, never NULL */
int         hmrm_device_count(void);            /* <0 on HIP error          */
int         hmrm_set_device(int device);

/* -------------------------------------------------------------------- scene */
/* Replaces the stbi_load results + UpdateHeightmap (hmap.cpp:314-353,171-191):
 * height_rgb is W*H*3 RGB8 (stbi_load req_comp 3), color_rgba W*H*4 RGBA8
 * (req_comp 4), both row-major top-left origin.  Uploads both once; heights are
 * converted on the device with the reference's exact operation order. */
int  hmrm_scene_create(const uint8_t *height_rgb, const uint8_t *color_rgba,
                       int32_t map_w, int32_t map_h,
                       const hmrm_scene_params *params, hmrm_scene **out);
/* Re-run UpdateHeightmap after min/max_height, lum_* (or grid_width) changed
 * (hmap.cpp:401-440,517-519). */
int  hmrm_scene_update(hmrm_scene *scene, const hmrm_scene_params *params);
void hmrm_scene_destroy(hmrm_scene *scene);
/* Copies the device height buffer (heightmap_buf, hmap.cpp:53,187) to `out`
 * (map_w*map_h doubles) -- test hook for UpdateHeightmap parity. */
int  hmrm_scene_read_heights(const hmrm_scene *scene, double *out);

/* ------------------------------------------------------------------- render */
/* Replaces one pass of the pixel loop main/hmap.cpp:978-1058 at `cycle 1` plus
 * the per-frame set-up :661-672,:952-974.  Writes every pixel of the
 * width x height RGBA8 frame (bytes R,G,B,A=255, top-left origin,
 * hmap.cpp:139-154) to host memory; stride_bytes >= width*4. */
int hmrm_render(const hmrm_scene *scene, const hmrm_camera *cam,
                uint8_t *rgba, size_t stride_bytes);

/* The same frame rendered by several scenes at once -- one per GPU, each created after
 * hmrm_set_device(i) with the same maps (BASELINE config C4's sharding of main/hmap.cpp:978-983's
 * independent pixels): scene i renders the cyclic 16-row bands i, i+n, ... and copies them to their
 * rows of `rgba` over its own PCIe link; no exchange between devices.  All kernels are launched before any
 * copy is enqueued and no device waits for another one's copy: into pinned `rgba` (hipHostMalloc /
 * hipHostRegister by the caller) the bands are copied directly, into pageable memory through a pinned
 * staging strip per scene and a host copy.  Same pixels and return codes as hmrm_render. */
int hmrm_render_multi(hmrm_scene *const *scenes, int32_t n_scenes, const hmrm_camera *cam,
                      uint8_t *rgba, size_t stride_bytes);

/* The same frame without blocking: hmrm_render_begin enqueues the kernel and the device-to-host
 * copy into a pinned frame owned by the scene and returns a ticket; hmrm_render_wait blocks until
 * that frame is in host memory and lends it out (*rgba, valid until hmrm_render_release; returns
 * HMRM_E_NOTERM like hmrm_render, the frame is still valid then).  Copies run on their own
 * stream, so with two or more frames in flight kernel k+1 overlaps the PCIe transfer of frame k
 * (replaces the per-frame blit SDL_UpdateTexture, hmap.cpp:1082).  Up to 64 frames in flight per
 * scene; the ring grows on demand and is freed with the scene.
 * Consecutive frames go to three scene-owned launch streams in turn: with two or more tickets in flight the
 * tail of one launch (a few long waves) also overlaps the start of the next (small frames: 1080p over a 1024^2 map
 * 0.064 -> 0.042 ms of GPU time per frame, profiles/r04_lanes.txt) -- the caller manages no stream. */
int  hmrm_render_begin(const hmrm_scene *scene, const hmrm_camera *cam, int32_t *ticket);
/* Per-frame flags of the ticketed entry points.  HMRM_NO_PROBE: never spend this frame on the scene's one-time kernel probe (a scene
 * whose cameras never repeat launches its sixth full frame twice, production kernel and plain groups, to measure which suits
 * its content: a 3-6 ms hiccup on a 4K frame) -- for a caller that counts on every frame's latency.  The probe
 * then waits for a frame without the flag (or the calibration of a repeated camera). */
#define HMRM_NO_PROBE 1u
int  hmrm_render_begin_flags(const hmrm_scene *scene, const hmrm_camera *cam, uint32_t flags, int32_t *ticket);
int  hmrm_render_wait(const hmrm_scene *scene, int32_t ticket, const uint8_t **rgba, size_t *stride_bytes);
void hmrm_render_release(const hmrm_scene *scene, int32_t ticket);

/* The same pass of main/hmap.cpp:978-1058 for a frame that stays on the GPU (a sequence of frames consumed there -- an
 * encoder, a compositor, a collective -- in place of the blit at hmap.cpp:1082): hmrm_render_device_begin launches the frame into the caller's DEVICE memory d_rgba (width x height
 * RGBA8, stride_bytes a multiple of 4) on the next of the scene's launch streams and returns a ticket;
 * hmrm_render_device_wait blocks until that frame is complete (HMRM_E_NOTERM like hmrm_render; the ticket is
 * free again either way).  Frames in flight must not share memory.  Up to 64 in flight per scene. */
int  hmrm_render_device_begin(const hmrm_scene *scene, const hmrm_camera *cam, void *d_rgba, size_t stride_bytes,
                              int32_t *ticket);
int  hmrm_render_device_begin_flags(const hmrm_scene *scene, const hmrm_camera *cam, void *d_rgba, size_t stride_bytes,
                                    uint32_t flags, int32_t *ticket);
int  hmrm_render_device_wait(const hmrm_scene *scene, int32_t ticket);

/* The reference's progressive frame driver (hmap.cpp:976-983, `cycle n` key, default 47):
 * rewrites only pixels p = cycle, cycle + cycle_period, ... (p = x + y*width) of `rgba`
 * and leaves the others as they are; cycle_period consecutive calls with cycle = 0 ..
 * cycle_period-1 and a static camera give the full frame.  (The reference advances
 * `cycle = (cycle + 1) % cycle_period` before each frame, hmap.cpp:976.) */
int hmrm_render_cycle(const hmrm_scene *scene, const hmrm_camera *cam,
                      uint8_t *rgba, size_t stride_bytes, int32_t cycle, int32_t cycle_period);

/* Same pass restricted to rows [row_begin,row_end) of the frame, written to a
 * DEVICE buffer that holds only those rows (row row_begin at d_rgba), enqueued
 * on `hip_stream` (a hipStream_t, NULL = default stream) without a host sync.
 * stride_bytes: a multiple of 4, at least width*4, below 2^33.
 * This is the multi-GPU row-strip entry point. `band_rows`>0 selects cyclic
 * banding: the strip holds bands band_index, band_index+band_count, ... of
 * band_rows rows each, packed back to back (row_begin/row_end then ignored). */
int hmrm_render_rows_device(const hmrm_scene *scene, const hmrm_camera *cam,
                            void *d_rgba, size_t stride_bytes,
                            int32_t row_begin, int32_t row_end,
                            int32_t band_rows, int32_t band_index, int32_t band_count,
                            void *hip_stream);

/* hmrm_render_rows_device does not wait for its launch, so it cannot report rays stopped by the
 * step cap (HMRM_E_NOTERM of hmrm_render; the reference's loop would not terminate for them,
 * hmap.cpp:1000).  This call waits for `hip_stream`, stores in *capped how many rays of the launches
 * enqueued on it through this scene reached the cap since the last call, and returns
 * HMRM_E_NOTERM when that is not zero. */
int hmrm_scene_take_capped(const hmrm_scene *scene, void *hip_stream, uint64_t *capped);

/* Rows the strip buffer of one rank must hold in cyclic-band mode (full bands). */
int32_t hmrm_band_local_rows(int32_t height, int32_t band_rows, int32_t band_index, int32_t band_count);

/* As hmrm_render, plus per-render statistics and optional per-pixel outputs
 * (host pointers, width*height entries each, may be NULL): the step count of
 * each ray and the slab-entry distance returned by distance() (AABB.cpp:49-77).
 * Uses the instrumented kernel variant; pixels are identical. */
int hmrm_render_stats(const hmrm_scene *scene, const hmrm_camera *cam,
                      uint8_t *rgba, size_t stride_bytes,
                      hmrm_stats *stats, uint32_t *steps_per_pixel, double *entry_d);

/* Device-side ImagePlane::GetRay (ImagePlane.hpp:10) and distance()
 * (AABB.hpp:12) for one pixel -- per-ray parity hooks. */
int hmrm_debug_ray(const hmrm_scene *scene, const hmrm_camera *cam,
                   int32_t px, int32_t py, double pos[3], double dir[3], double *entry_d);

/* Host-only (no GPU needed): the per-frame record the kernel receives for this
 * camera -- the reference's per-frame set-up, main/hmap.cpp:661-672,:952-974.
 * out25 = cam[3], upper_left[3], plane_right[3], plane_down[3], look[3], c0[3],
 * c1[3], nudge, step_dist, grid_pow2, inv_grid_width.  `tables` (may be NULL;
 * used for HMRM_SPHERICAL) receives 2*width + 2*height doubles: cos(ha) and
 * sin(ha) per column, then sin(va) and cos(va) per row (Spherical.cpp:18-25). */
int hmrm_debug_frame(const hmrm_camera *cam, const hmrm_scene_params *params,
                     int32_t map_w, int32_t map_h, double *out25, double *tables);

/* Host-only test hook (no GPU needed, no reference counterpart: the reference's OpenMP loop has no launch order).
 * The library hands a frame's 16-row tile rows to the GPU in an order calibrated from one measured launch per
 * cached camera: records[2 t] = start of tile row t's first workgroup, records[2 t + 1] = duration of its longest
 * wave (10 ns ticks), measured under the plain rotation that starts at tile row `rot`.  Returns the number of
 * contiguous pieces (0..3; 0 = the rotation stays) the plan starts first, in order, in pieces_begin / pieces_count,
 * and, when tile_row_of_grid_row is not NULL (tile_rows entries), the resulting permutation: which tile row the
 * grid's row j renders.  Scheduling only -- no order changes a pixel.  Negative = HMRM_E_ARG. */
int hmrm_debug_plan_order(const uint64_t *records, int32_t tile_rows, int32_t rot, int32_t pieces_begin[3],
                          int32_t pieces_count[3], int32_t *tile_row_of_grid_row);

/* Host-only test hook (no GPU, no reference counterpart): the state machine behind that calibration and the scene's kernel
 * probe, driven through `launches` full-frame launches of one camera.  records holds, per launch, tile_rows x {start,
 * longest wave} as above: what a measured launch would report (read only for the launches the machine decides to measure; the
 * report arrives before the next launch).  can_measure[i] (NULL = always): the caller has free record buffers and the scene's
 * other streams are idle.  Out, per launch: the trial whose order it uses (0 = the rotation), whether it is measured, whether it
 * runs the plain-groups kernel; then the number of trials made, the settled one (-1: none yet), the scene's verdict
 * (1 = plain groups) and the launch whose report settled the calibration (-1: not settled). */
int hmrm_debug_calibrate(const uint64_t *records, int32_t launches, int32_t tile_rows, int32_t rot, int32_t may_probe,
                         int32_t scene_already_probed, const uint8_t *can_measure, int32_t *trial_used, int32_t *measured,
                         int32_t *group_kernel, int32_t *n_trials, int32_t *best, int32_t *scene_use_group,
                         int32_t *settled_at_launch);

/* Accuracy of the hardware reciprocal v_rcp_f64 on the current device (test hook; no reference counterpart:
 * the reference divides, AABB.cpp:62-63, and the kernel's one-division shortcut through distance() must prove
 * from approximate quotients which exact quotient is the result).  mode 0: the leading 32 mantissa bits
 * exhaustively for exponent exp_lo (count = 2^32 covers them; trailing 20 bits 0 / all ones / hashed by
 * seed & 3); mode 1: hashed mantissa, sign and exponent in [exp_lo, exp_hi]; mode 2: n * rcp(d) against the
 * correctly rounded n / d for box-like n (exponent in [exp_lo, exp_hi]) and direction-like d (2^-40..1).
 * *max_rel_err = largest relative error seen; hist64 (may be NULL) = 64 counters, [k] = samples with an
 * error in [2^-k, 2^-(k-1)), [63] also holds the exact ones. */
int hmrm_debug_rcp_error(int32_t mode, uint64_t count, uint64_t seed, int32_t exp_lo, int32_t exp_hi,
                         double *max_rel_err, uint64_t *hist64);

/* Test hook, needs no GPU: the window-maximum pyramid layout hmrm_scene_create chooses for a map_w x map_h map
 * (row pitch in windows, log2 of the plane pitch, number of levels).  Returns 1 when 32-bit BYTE offsets would cover
 * every plane, 0 when only 64-bit ones do (very oblong maps near the 2^29-cell limit, e.g. 16385 x 32766).  The
 * production kernel forms the offsets in 64 bits since round 5, so both kinds of map render with it (round 4 sent the
 * second kind through the literal loop, main/hmap.cpp:1000-1038 as written, 85 x slower).  Negative = HMRM_E_ARG. */
int hmrm_debug_mip_layout(int32_t map_w, int32_t map_h, int32_t *mip_row, int32_t *plane_shift, int32_t *levels);

/* Which kernel full frames of this scene are rendered with: 0 = the production kernel (speculative groups + exact
 * leaps over the window-maximum pyramid), 1 = the speculative groups alone, 2 = the literal loop, 3 = the speculative
 * groups with leaps over window records (a 16-cell window's maximum without its 8 highest cells, and where those stand:
 * nearest-sampling frames; frames of the other sampling modes keep the production kernel then).  1 or 3 without
 * HMRM_KERNEL=group / rec means the scene's one-time probe (part of the launch-order calibration of the first camera
 * that is rendered repeatedly, or -- for cameras that never repeat -- the scene's sixth full frame launched twice)
 * measured that kernel at least 3 % faster on this content -- maps on which rays cannot jump over whole windows: needles
 * on a plateau, white noise. */
int hmrm_debug_kernel_choice(const hmrm_scene *scene);

/* Test hook: the scene's window records -- one 32-byte record per 16 x 16-cell window of the map, a window every 4 cells,
 * ceil(map_w / 4) per row and ceil(map_h / 4) rows: { float max2; uint32 spare; uint8 xs[8]; uint8 ys[8]; uint32 spare[2] },
 * max2 = the window's maximum hit threshold without its 8 highest cells (rounded up to float), xs / ys = where the cells
 * strictly above it stand inside the window (255: slot unused) -- and the threshold table they were built from
 * (heightmap_buf[i] + min_height, main/hmap.cpp:1013-1016; map_w x map_h doubles), copied to the host.  Either pointer
 * may be NULL.  The record kernel (hmrm_debug_kernel_choice 3) leaps across a window when the ray stays at or above max2
 * and its path misses the recorded cells; no counterpart in the reference, which samples every step. */
int hmrm_debug_read_records(const hmrm_scene *scene, void *records_out, double *thr_out);

/* Test hook (no GPU): which kernel a frame is launched with -- returns 0 the plain groups, 1 the production kernel, 2 the
 * record kernel -- given HMRM_KERNEL (`forced`: 0 none, 1 group, 3 rec), whether the launch plan or the scene's verdict
 * asks for the other kernel, whether this frame could run the record kernel (nearest sampling), and the scene's probe
 * state (verdict for the other kernel; obtained with the record kernel or with the plain groups).  A verdict holds for
 * the frames that would run what was measured.  *with_records_after: what the scene remembers afterwards (a probe's own
 * launch notes what it measures). */
int hmrm_debug_pick_kernel(int32_t forced, int32_t use_other, int32_t records_ok, int32_t scene_verdict, int32_t scene_with_records,
                           int32_t *with_records_after);

/* The environment knobs (HMRM_KERNEL, HMRM_STEP_CAP, ...) are read once, when a
 * scene is created; this re-reads them for a live scene (tests and tools switch kernel variants).  Launch orders
 * calibrated so far are forgotten (they were measured on the old kernel variant). */
int hmrm_debug_reload_env(hmrm_scene *scene);

/* Time of the most recent render kernel launch on this thread, measured with
 * HIP events on the launch stream (ms); <0 if none. Only valid after
 * hmrm_render / hmrm_render_stats, which synchronise. */
double hmrm_last_kernel_ms(void);

/* Launches the render kernel `iters` times back to back into a device scratch
 * frame (no D2H) and returns the mean kernel duration in ms measured with HIP
 * events on the launch stream; <0 on error.  Bench hook. */
double hmrm_bench_kernel_ms(const hmrm_scene *scene, const hmrm_camera *cam, int32_t iters);

/* ---------------------------------------------------------------- recording */
/* Programmatic animation for the reference's recording mode (hmap.cpp:869-900,
 * :916-926 is an empty "alter this block and recompile" stub, :1131-1144 saves
 * frames).  The sweep is the orbit of BASELINE config C5: camera of frame k of
 * `frames` on a horizontal circle of `radius` around (centre_x, centre_y), looking at
 * the centre: hang_k = hang0 + 2*pi*k/frames, pos_k = centre - radius*(cos, sin)(hang_k);
 * everything else from `base`. */
void hmrm_orbit_camera(const hmrm_camera *base, double centre_x, double centre_y, double radius,
                       double hang0, int32_t frame, int32_t frames, hmrm_camera *out);
/* Renders `frames` orbit frames and saves them as <dir>/hmap_<id>_<n>.png
 * (hmap.cpp:1132-1134) with a pool of PNG encoder threads (0 = one per host core).
 * verbose != 0 prints the reference's "Saved screenshot at ..." / "Done recording." lines. */
int hmrm_record_orbit(const hmrm_scene *scene, const hmrm_camera *base, double centre_x, double centre_y,
                      double radius, double hang0, int32_t frames, const char *dir, long long id,
                      int32_t encoder_threads, int32_t verbose);
/* The same sweep sharded over several scenes -- one per GPU, each created after hmrm_set_device
 * with the same maps (BASELINE config C5): frame k is rendered by scenes[k mod n_scenes]
 * (hmrm_orbit_frame_owner), no exchange between devices, one shared pool of encoder threads.
 * Files and bytes are those of hmrm_record_orbit. */
int hmrm_record_orbit_multi(hmrm_scene *const *scenes, int32_t n_scenes, const hmrm_camera *base,
                            double centre_x, double centre_y, double radius, double hang0, int32_t frames,
                            const char *dir, long long id, int32_t encoder_threads, int32_t verbose);
int32_t hmrm_orbit_frame_owner(int32_t frame, int32_t n_devices);

/* ------------------------------------------------------------------- config */
/* Replaces ConsumeConfigStream (main/hmap.cpp:309-520) and the globals'
 * defaults (:31-112).  Same whitespace token grammar, same 27 keys, same echo of
 * every option to `echo_fd`-style sinks: echo text is appended to an internal
 * log retrievable with hmrm_config_log().  Additive keys (not in the reference,
 * named by north_star): `projection perspective|spherical|orthographic|1|2|3`,
 * `output <path.png|.ppm>`, `record orbit|off`, `devices n`, `sampling nearest|bilinear`, `heights f64|f32`.  Unknown key -> "WARNING: Unknown identifier: k". */
hmrm_config *hmrm_config_create(void);
void         hmrm_config_destroy(hmrm_config *cfg);
/* Consume a whole stream; loads heightmap/colormap images when those keys
 * appear; runs the end-of-stream validation (maps present, equal dimensions). */
int          hmrm_config_consume_file(hmrm_config *cfg, const char *path);
int          hmrm_config_consume_string(hmrm_config *cfg, const char *text);
const char  *hmrm_config_log(const hmrm_config *cfg);      /* stdout echo so far   */
const char  *hmrm_config_warnings(const hmrm_config *cfg); /* stderr text so far   */
void         hmrm_config_get_camera(const hmrm_config *cfg, hmrm_camera *out);
void         hmrm_config_get_scene_params(const hmrm_config *cfg, hmrm_scene_params *out);
int32_t      hmrm_config_cycle(const hmrm_config *cfg);
int32_t      hmrm_config_recording_frame_count(const hmrm_config *cfg);
const char  *hmrm_config_heightmap_path(const hmrm_config *cfg);
const char  *hmrm_config_colormap_path(const hmrm_config *cfg);
const char  *hmrm_config_output_path(const hmrm_config *cfg);
int32_t      hmrm_config_record_mode(const hmrm_config *cfg);   /* additive `record orbit|off`: 1|0 */
int32_t      hmrm_config_devices(const hmrm_config *cfg);       /* additive `devices n`: GPUs for recording, 0 = all */
/* Loaded maps (owned by cfg): RGB8 / RGBA8; NULL until the key was consumed. */
const uint8_t *hmrm_config_height_rgb(const hmrm_config *cfg, int32_t *w, int32_t *h);
const uint8_t *hmrm_config_color_rgba(const hmrm_config *cfg, int32_t *w, int32_t *h);
/* 1 if heights need recomputing since the last call (should_update_heightmap). */
int          hmrm_config_take_heightmap_dirty(hmrm_config *cfg);
/* Scene from the loaded maps + params (= hmrm_scene_create on the above). */
int          hmrm_config_create_scene(const hmrm_config *cfg, hmrm_scene **out);

/* ----------------------------------------------------------------- image IO */
/* Replaces stbi_load(path,&w,&h,&n,req_comp) (hmap.cpp:320-321,341-342) for the formats decoded
 * natively: PNG (all colour types / bit depths, interlaced too), JPEG (baseline, extended and
 * progressive Huffman, 8-bit; grey, YCbCr, RGB, CMYK/YCCK), BMP (1/4/8-bit palette, 16/24/32-bit,
 * bit fields), TGA (types 1/2/3/9/10/11) and binary PNM (P5/P6).  Pixels equal stb_image v2.27's
 * for every req_comp (16-bit -> 8 by >>8; grey -> RGB replicate; missing alpha = 255).  The other
 * formats the reference's stb reads (README.md:75) are decoded too, with stb's pixels: GIF (first frame),
 * PSD (8/16-bit RGB, raw or RLE), Softimage PIC and Radiance HDR (stb's float -> 8-bit tone curve).
 * *out is malloc'ed; free with hmrm_image_free. */
int  hmrm_image_load(const char *path, int32_t req_comp,
                     uint8_t **out, int32_t *w, int32_t *h, int32_t *comp_in_file);
int  hmrm_image_load_memory(const uint8_t *bytes, size_t len, int32_t req_comp,
                            uint8_t **out, int32_t *w, int32_t *h, int32_t *comp_in_file);
void hmrm_image_free(void *p);
/* Replaces SavePNG / stbi_write_png(path,w,h,comp,data,stride)
 * (hmap.cpp:157-160): same filter choice and the same deflate as
 * stb_image_write v1.16, so equal pixels give equal files. */
int  hmrm_write_png(const char *path, int32_t w, int32_t h, int32_t comp,
                    const uint8_t *data, size_t stride_bytes);
int  hmrm_write_png_memory(int32_t w, int32_t h, int32_t comp, const uint8_t *data,
                           size_t stride_bytes, uint8_t **out, size_t *out_len);
/* Binary PPM (P6), alpha dropped -- build-side addition named by north_star. */
int  hmrm_write_ppm(const char *path, int32_t w, int32_t h, int32_t comp,
                    const uint8_t *data, size_t stride_bytes);

#ifdef __cplusplus
}
#endif
#endif /* HMRM_H */
