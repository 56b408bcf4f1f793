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

// ---- the calibration as a state machine (host logic only: api.cpp owns the HIP events and the record buffers) -----------
// A camera that is rendered again and again is CALIBRATED: a short list of trial orders -- [0] the rotation, then the
// model's plan and a generic head / tail / middle split made from the rotation's first records -- each timed by
// kOrderSamples measured full-frame launches, after which the one with the shortest measured makespan stays (another
// order must beat the rotation by 1 %).  The last candidate of ONE record per scene is not an order but the OTHER KERNEL:
// the plain speculative groups without pyramid leaps (with leaps over window records where the frame's sampling allows
// them: api.cpp launch_kernel), under the rotation.  On content that admits no jumps (needles on a plateau,
// white noise: profiles/r04_content.txt) the leap kernel's attempts are pure overhead; the kernel that measures 3 % faster
// renders the scene from then on.  Same pixels whatever is chosen.
constexpr int kOrderSamples = 2; // measured launches per trial (one launch's makespan wobbles by a few per cent)

struct OrderTrial {
	int n = 0, b[3] = {0, 0, 0}, c[3] = {0, 0, 0}; // pieces (n = 0: the plain rotation)
	bool group = false;                            // the trial runs the plain groups (no leaps), rotation order
	double makespan = 0.0;                         // measured, ticks: the shorter of its samples; 0 = not yet
	int samples = 0;
};

// Which kernel suits a scene's content: probed ONCE per scene and height update, by the first camera that gets calibrated
// or -- for cameras that never repeat -- by the scene's shadow probe (api.cpp); everybody else renders with the verdict.
struct KernelChoice {
	bool probed = false;           // some record's trials hold (or held) the probe, or the shadow probe has been launched
	bool use_group = false;        // the verdict: the other kernel -- the plain groups, or for nearest-sampling frames their
	                               // extension by window records (frame.hpp WindowRecord) -- measured at least 3 % faster
	bool with_records = false;     // which of the two the probe's launches ran: the verdict holds for frames that would run the same
	unsigned unprobed_frames = 0;  // full frames rendered without any probe (the shadow probe's trigger)
	void reset() { *this = KernelChoice(); }
};

struct LaunchPlan {
	int trial = 0;          // whose order to launch with (index into OrderCalibration::trials)
	bool measure = false;   // ... as a measured launch, to be reported with on_measured
	bool use_group = false; // render with the plain groups
};

struct OrderCalibration { // of one cached camera
	OrderTrial trials[4];
	int n_trials = 1;     // known so far (the candidates are made from the rotation's first records)
	int in_flight = -1;   // the trial whose measured launch has not been reported yet
	int best = -1;        // settled: index into trials (-1: still calibrating, the rotation is used)
	unsigned uses = 0;    // full-frame launches planned for this camera

	void reset() { *this = OrderCalibration(); }
	bool probing() const; // this record's trials hold the scene's kernel probe
	// Would the next plan() ask for a measured launch if it may?  (So that the caller only looks whether it may -- event
	// queries on the scene's other streams -- while a calibration is still going on.)
	bool wants_measure() const;
	// Another stream has settled this camera already: adopt its result instead of measuring again.
	void adopt(const OrderTrial &settled);
	// The next full-frame launch of this camera.  `can_measure`: the caller has a free set of record buffers and nothing
	// else of the scene is running that would disturb a measurement.  A plan with `measure` set marks its trial in flight.
	LaunchPlan plan(bool can_measure, const KernelChoice &scene);
	// The measured launch in flight has been read back (rec: measured_makespan's layout).  `may_probe`: the scene allows a
	// kernel probe (production kernel, probe not disabled).  Returns true when the calibration has just settled: trials[best]
	// is the result (its `group` flag the scene's verdict when this record held the probe -- already stored in `scene`).
	bool on_measured(const unsigned long long *rec, int tiles_y, int rot, bool may_probe, KernelChoice &scene);
	// The measured launch in flight is lost (its records were reused, the knobs changed): forget it.
	void drop_in_flight() { in_flight = -1; }
};

// Which kernel a frame is launched with: 0 the plain groups, 1 pyramid leaps (production), 2 window records (the values of
// render.hpp FastKernel).  `forced`: HMRM_KERNEL (0 none, 1 group, 3 rec); `use_other`: the launch plan or the scene's
// verdict asks for the other kernel; `records_ok`: this frame could run the record kernel (nearest sampling, table there).
// The other kernel is the record kernel where it can run, else the plain groups.  A verdict holds for the frames that would
// run what the probe measured: one obtained with the records leaves bilinear / float-heights frames on the production
// kernel, one obtained with the plain groups leaves nearest frames there.  A probe's own launch -- `use_other` while the
// scene has no verdict -- notes in scene.with_records what it measures.
int pick_fast_kernel(int forced, bool use_other, bool records_ok, KernelChoice &scene);

// The shadow probe's verdict: one frame launched twice, production kernel then plain groups, both measured.
void fold_shadow_probe(KernelChoice &scene, const unsigned long long *leap_rec, const unsigned long long *group_rec, int tiles_y);

} // namespace hmrm
