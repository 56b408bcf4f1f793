// render.hpp -- launch interface of render.hip (device code) for api.cpp.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "frame.hpp"

namespace hmrm {

// UpdateHeightmap on the device.  plain=false writes thr[i] = heightmap_buf[i] + min_h and
// folds max(thr) into *d_max_key (order-preserving key, zero-initialised by the caller);
// plain=true writes heightmap_buf[i] itself (test hook).
hipError_t launch_prepare_heights(const uint8_t *d_rgb, double *d_out, int64_t n, double lum_r,
                                  double lum_g, double lum_b, double min_h, double max_h, bool plain,
                                  unsigned long long *d_max_key, hipStream_t stream);
double max_key_to_double(unsigned long long key);

// One pass of the pixel loop over the rows described by `rows`, into d_out
// (uint32 RGBA per pixel, out_stride_px pixels per local row).
// d_counters: 3 x uint64 {steps, hits, capped}; steps/hits only filled when stats.
hipError_t launch_render(const DevFrame &f, const RowMap &rows, const double *d_thr,
                         const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px,
                         unsigned long long *d_counters, uint32_t *d_steps, double *d_entry,
                         bool stats, hipStream_t stream);

// Production kernel (render_fast.hip): speculative step groups (kPlainGroups), plus exact leaps over empty pyramid
// windows (kLeaps), or over windows of the record level that are empty but for a few recorded cells the ray's path
// misses (kRecords: nearest sampling only, `d_records` from launch_build_records).  Same outputs as launch_render.
// f.sampling == 2 reads the float copy of the table (d_thr32, launch_thr_to_float) instead of d_thr.
enum FastKernel { kPlainGroups = 0, kLeaps = 1, kRecords = 2 };
hipError_t launch_render_fast(const DevFrame &f, const RowMap &rows, const double *d_thr, const float *d_thr32,
                              const uint32_t *d_cmap, uint32_t *d_out, int64_t out_stride_px,
                              unsigned long long *d_counters, uint32_t *d_steps, double *d_entry, bool stats,
                              FastKernel kernel, const WindowRecord *d_records, hipStream_t stream);
// The record table of the thr table: rec_row(map_w) x ceil(map_h / 4) WindowRecords.
hipError_t launch_build_records(const double *d_thr, int map_w, int map_h, WindowRecord *d_dst, hipStream_t stream);
// thr32[i] = (float)thr[i], round to nearest (the "float heights" mode).
hipError_t launch_thr_to_float(const double *d_thr, float *d_thr32, int64_t n, hipStream_t stream);
// 3x3 maximum filter of the thr table (bounds every bilinear interpolation, render_fast.hip).
hipError_t launch_dilate3x3(const double *d_thr, int w, int h, double *d_dst, hipStream_t stream);
// Window-maximum pyramid (see render_fast.hip): level 0 from the thr table, level l+1 from level l.
// `pitch` = row pitch (floats) of every plane of the pyramid buffer (DevFrame::mip_row).
hipError_t launch_build_mip0(const double *d_thr, int map_w, int map_h, float *d_dst, int dst_w, int dst_h,
                             int pitch, hipStream_t stream);
hipError_t launch_build_mip_up(const float *d_src, int src_w, int src_h, float *d_dst, int dst_w, int dst_h,
                               int pitch, int src_level, hipStream_t stream);
// round-up-to-float of a double (the pyramid's rounding), on the host: for the whole-map element
float round_up_to_float_host(double v);

// Pixel tile (= workgroup) shape of launch_render / launch_render_fast.
void render_tile_shape(int *tile_w, int *tile_h);

// GetRay + distance() of pixel (px,py): d_out7 = pos[3], dir[3], d.
hipError_t launch_probe(const DevFrame &f, int px, int py, double *d_out7, hipStream_t stream);

// Launch-order calibration (RowMap::measure): device records of tile_rows x kMeasureStride words, zeroed before a
// measured launch and reduced afterwards to {start of the row's first workgroup, longest wave of the row} per tile row
// in pinned (device-mapped) host memory -- both small kernels of the launch stream.
hipError_t launch_measure_init(unsigned long long *d_rec, int tile_rows, hipStream_t stream);
hipError_t launch_measure_readback(const unsigned long long *d_rec, unsigned long long *h_pinned_dev, int tile_rows, hipStream_t stream);

// n doubles from pinned (device-mapped) host memory into device memory, as a kernel on `stream` (render.hip).
hipError_t launch_upload_tables(const double *h_pinned, double *d_dst, size_t n, hipStream_t stream);

// v_rcp_f64 accuracy probe (render.hip k_rcp_error): d_out65[0] = max relative error as fp64 bits, [1..64] = histogram
// by binary order of magnitude; the caller zeroes d_out65.
hipError_t launch_rcp_error(int mode, uint64_t count, uint64_t seed, int exp_lo, int exp_hi,
                            unsigned long long *d_out65, hipStream_t stream);

} // namespace hmrm
