// launch_order.hpp -- in which order a launch's 16-row tile rows are handed to the GPU (host logic, no device code).
// The reference's OpenMP loop (main/hmap.cpp:978) has no counterpart: its static schedule gives every thread the same
// share of pixels; on the GPU the workgroups start in index order and what is handed out last decides how long a
// launch outlives its dispatch.  Scheduling only -- no order changes a pixel.
#pragma once
#include <vector>

#include "frame.hpp"

namespace hmrm {

constexpr int kCostRows = 16; // screen rows per entry of a row-cost estimate: one sample row per 16-row tile row

// First tile row of the plain rotation: the first one whose estimated march length (row_cost, row_cost.cpp) reaches a
// tenth of the frame's maximum.  0 when disabled, for band launches or without an estimate.
int choose_tile_rot(bool enabled, const std::vector<float> &row_cost, const RowMap &rows, int tile_h);

// RowMap's launch order from up to three contiguous tile-row pieces that start first, in the order given (disjoint,
// together one contiguous range), followed by the rest of the frame from the end of that range on, wrapping around.
// No (or invalid) pieces = the rotation by `rot`.
void set_tile_order(RowMap *r, int tiles_y, int rot, int n, const int *b, const int *c);

// Records of a measured launch: rec[2 t] = start of tile row t's first workgroup, rec[2 t + 1] = its longest wave
// (s_memrealtime ticks); 0 start = no record.
double measured_makespan(const unsigned long long *rec, int tiles_y);
// Candidate orders from the records of a launch measured under the rotation by `rot`: the plan of the delivery-time
// model, and a generic head / tail / middle split of the measured hot range.  -> number of pieces (0: none).
int plan_order_from_measurement(const unsigned long long *rec, int tiles_y, int rot, int *pb, int *pc);
int split_hot_range(const unsigned long long *rec, int tiles_y, int rot, double head_frac, double tail_frac, int *pb, int *pc);

} // namespace hmrm
