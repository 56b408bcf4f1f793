// launch_order.cpp -- see launch_order.hpp.
#include "launch_order.hpp"

#include <algorithm>

namespace hmrm {

// Launch order.  Workgroups start in index order, and the waves that graze the terrain near the
// horizon run 50-100x longer than the rest: if their tile rows come late in the grid, the whole
// launch waits for them with most of the chip idle (C3: up to +45 %).  So the grid is rotated to
// begin at the first tile row whose estimated march length (row_cost, per frame, host) reaches
// a tenth of the frame's maximum: the long rows start first, the cost then falls off towards
// the bottom of the frame, and the rows above (sky, or short marches) fill the tail.
// Scheduling only; HMRM_TILE_ORDER=0 keeps row-major order for A/B runs.
int choose_tile_rot(bool enabled, const std::vector<float> &row_cost, const RowMap &rows, int tile_h) {
	if (!enabled || rows.band_rows > 0 || row_cost.empty()) return 0;
	const int tiles_y = (rows.local_rows + tile_h - 1) / tile_h;
	if (tiles_y <= 1) return 0;
	auto tile_cost = [&](int t) {
		float c = 0.0f;
		const int r0 = rows.row_begin + t * tile_h, r1 = r0 + tile_h - 1;
		for (int k = r0 / kCostRows; k <= r1 / kCostRows && k < (int)row_cost.size(); ++k)
			if (row_cost[(size_t)k] > c) c = row_cost[(size_t)k];
		return c;
	};
	float top = 0.0f;
	for (int t = 0; t < tiles_y; ++t) top = std::max(top, tile_cost(t));
	if (!(top > 0.0f)) return 0;
	for (int t = 0; t < tiles_y; ++t)
		if (tile_cost(t) >= 0.1f * top) return t;
	return 0;
}

// ---- launch order from measurement ----------------------------------------------------------------------------
// What decides how long a launch outlives its dispatch is which marching tile rows are handed out LAST: their
// longest wave runs on after everything else has drained (tools/timeline.py: on C3 the bottom rows of the frame hold
// steep rays whose waves keep all lanes busy in both blocks of every loop trip and run 60 us, the middle rows 40 us,
// the horizon rows up to 130 us).  An estimate of a row's march length says nothing about that, so it is MEASURED:
// the second full-frame launch of a cached camera runs with RowMap::measure set and every wave folds its start time
// and duration into its tile row's record.  From the records: dispatch time of every tile row under the rotation
// (D[i] = start of the next row - start of this one) and its longest wave q[i].  Handing the rows out in another
// order moves their start times by the D of the rows put in front, and the launch lasts max_i (start'_i + q_i): the
// classic delivery-time problem, best solved by "longest q first".  With the three contiguous pieces the kernel's
// row map offers that means: head [h0, a) first (the horizon rows, as with the rotation), then the tail [b, end),
// then the middle [a, b) -- a and b chosen by exhaustive search over the model.  The third launch runs that
// candidate, measured again, and whichever order had the shorter measured makespan is kept.  Scheduling only.
double measured_makespan(const unsigned long long *rec, int tiles_y) {
	unsigned long long s0 = ~0ull;
	for (int t = 0; t < tiles_y; ++t)
		if (rec[2 * t]) s0 = std::min(s0, rec[2 * t]);
	double m = 0.0;
	for (int t = 0; t < tiles_y; ++t)
		if (rec[2 * t]) m = std::max(m, (double)(rec[2 * t] - s0) + (double)rec[2 * t + 1]);
	return m;
}

// A generic candidate that needs no model: the measured hot range [rot, last row whose longest wave is >= 5 % of the
// frame's longest) cut into head, middle and tail by fractions; order head, tail, middle.  -> pieces (0: none).
int split_hot_range(const unsigned long long *rec, int tiles_y, int rot, double head_frac, double tail_frac, int *pb, int *pc) {
	if (tiles_y < 12 || rot < 0 || rot >= tiles_y) return 0;
	unsigned long long qmax = 0;
	for (int t = 0; t < tiles_y; ++t) qmax = std::max(qmax, rec[2 * t + 1]);
	int n = 0;
	for (int t = rot; t < tiles_y; ++t)
		if ((double)rec[2 * t + 1] >= 0.05 * (double)qmax) n = t - rot + 1;
	const int head = (int)(n * head_frac), tail = (int)(n * tail_frac);
	if (n < 12 || head < 1 || tail < 1 || head + tail >= n) return 0;
	pb[0] = rot;            pc[0] = head;
	pb[1] = rot + n - tail; pc[1] = tail;
	pb[2] = rot + head;     pc[2] = n - head - tail;
	return 3;
}

// -> number of pieces (0: keep the rotation); records measured under the rotation by `rot`.
int plan_order_from_measurement(const unsigned long long *rec, int tiles_y, int rot, int *pb, int *pc) {
	if (tiles_y < 12 || rot < 0 || rot >= tiles_y) return 0;
	unsigned long long s0 = ~0ull;
	for (int t = 0; t < tiles_y; ++t) {
		if (rec[2 * t] == 0ull) return 0; // a tile row without a record: not a launch this plan understands
		s0 = std::min(s0, rec[2 * t]);
	}
	const int N = tiles_y;
	std::vector<double> st((size_t)N), q((size_t)N), D((size_t)N);
	double qmax = 0.0;
	for (int i = 0; i < N; ++i) {
		const int t = (rot + i) % N;
		st[(size_t)i] = (double)(rec[2 * t] - s0);
		if (i > 0) st[(size_t)i] = std::max(st[(size_t)i], st[(size_t)i - 1]); // (rows start in order; noise aside)
		q[(size_t)i] = (double)rec[2 * t + 1];
		qmax = std::max(qmax, q[(size_t)i]);
	}
	for (int i = 0; i + 1 < N; ++i) D[(size_t)i] = st[(size_t)i + 1] - st[(size_t)i];
	D[(size_t)N - 1] = N > 1 ? D[(size_t)N - 2] : 0.0;
	// hot range in dispatch order: up to the last row (before the frame's end) whose longest wave matters
	int n = 0;
	for (int i = 0; i < N - rot; ++i)
		if (q[(size_t)i] >= 0.05 * qmax) n = i + 1;
	if (n < 12) return 0;
	// Two objectives.  PRIMARY: the modelled makespan max_i (start'_i + q_i).  Rows whose longest wave alone nearly
	// fills the launch (the horizon rows: q >= 0.85 of the measured makespan) bound it from below whatever the order, and
	// in practice even they finish earlier when less is left running beside them at the end -- which the model, with
	// its fixed q, cannot see.  So the search minimises the SECONDARY objective, the same maximum over all other rows,
	// among the orders that do not make the primary one worse; the measured third launch has the last word.
	std::vector<double> cum((size_t)n + 1, 0.0), g((size_t)n), g2((size_t)n), pm((size_t)n + 1, 0.0), sm((size_t)n + 1, -1e300),
	    pm2((size_t)n + 1, 0.0), sm2((size_t)n + 1, -1e300);
	double floor_ms = 0.0; // everything after the hot range starts after it whatever its internal order
	for (int i = n; i < N; ++i) floor_ms = std::max(floor_ms, st[(size_t)i] + q[(size_t)i]);
	double span = floor_ms;
	for (int i = 0; i < n; ++i) span = std::max(span, st[(size_t)i] + q[(size_t)i]);
	for (int i = 0; i < n; ++i) {
		cum[(size_t)i + 1] = cum[(size_t)i] + D[(size_t)i];
		g[(size_t)i] = cum[(size_t)i] + q[(size_t)i];
		g2[(size_t)i] = q[(size_t)i] >= 0.85 * span ? -1e300 : g[(size_t)i];
		pm[(size_t)i + 1] = std::max(pm[(size_t)i], g[(size_t)i]);
		pm2[(size_t)i + 1] = std::max(pm2[(size_t)i], g2[(size_t)i]);
	}
	for (int i = n - 1; i >= 0; --i) {
		sm[(size_t)i] = std::max(sm[(size_t)i + 1], g[(size_t)i]);
		sm2[(size_t)i] = std::max(sm2[(size_t)i + 1], g2[(size_t)i]);
	}
	const double base = std::max(pm[(size_t)n], floor_ms), base2 = std::max(pm2[(size_t)n], 0.0);
	double best2 = base2;
	int best_a = 0, best_b = 0;
	const int step = std::max(1, n / 96);
	for (int a = 0; a < n; a += step) {
		double mid = -1e300, mid2 = -1e300; // max of g / g2 over [a, b)
		int scanned = a;
		for (int b2 = a + step; b2 < n; b2 += step) {
			for (; scanned < b2; ++scanned) {
				mid = std::max(mid, g[(size_t)scanned]);
				mid2 = std::max(mid2, g2[(size_t)scanned]);
			}
			const double shift_tail = cum[(size_t)a] - cum[(size_t)b2], shift_mid = cum[(size_t)n] - cum[(size_t)b2];
			const double m1 = std::max(std::max(pm[(size_t)a], sm[(size_t)b2] + shift_tail), std::max(mid + shift_mid, floor_ms));
			if (m1 > 1.001 * base) continue;
			const double m2 = std::max(pm2[(size_t)a], std::max(sm2[(size_t)b2] + shift_tail, mid2 + shift_mid));
			if (m2 < best2) {
				best2 = m2;
				best_a = a;
				best_b = b2;
			}
		}
	}
	if (!(best2 < 0.95 * base2) || best_b <= best_a) return 0;
	int k = 0;
	if (best_a > 0) { pb[k] = rot; pc[k] = best_a; ++k; }
	pb[k] = rot + best_b; pc[k] = n - best_b; ++k;
	pb[k] = rot + best_a; pc[k] = best_b - best_a; ++k;
	return k;
}

// RowMap's launch order from up to three contiguous tile-row pieces that start first, in the order given (they must
// be disjoint and form one contiguous range of tile rows), followed by the rest of the frame from the end of that
// range onwards, wrapping around.  No pieces = plain rotation by `rot`.
void set_tile_order(RowMap *r, int tiles_y, int rot, int n, const int *b, const int *c) {
	for (int k = 0; k < 3; ++k) r->seg_first[k] = 0x7fffffff;
	for (int k = 0; k < 4; ++k) r->seg_delta[k] = 0;
	r->seg_delta[0] = rot;
	if (n <= 0 || tiles_y >= 32768) return;
	int lo = tiles_y, hi = 0, total = 0;
	for (int k = 0; k < n; ++k) {
		if (b[k] < 0 || c[k] <= 0 || b[k] + c[k] > tiles_y) return;
		for (int j = 0; j < k; ++j)
			if (b[k] < b[j] + c[j] && b[j] < b[k] + c[k]) return; // overlap
		lo = std::min(lo, b[k]);
		hi = std::max(hi, b[k] + c[k]);
		total += c[k];
	}
	if (hi - lo != total) return; // not one contiguous range
	int first = 0;
	for (int k = 0; k < n; ++k) {
		if (k > 0) r->seg_first[k - 1] = first;
		r->seg_delta[k] = b[k] - first;
		first += c[k];
	}
	if (total < tiles_y) { // the rest: from `hi` on, wrapping
		r->seg_first[n - 1] = first;
		r->seg_delta[n] = hi - first;
	}
}

// ---- the calibration's state machine --------------------------------------------------------------------------------
bool OrderCalibration::probing() const {
	for (int k = 1; k < n_trials; ++k)
		if (trials[k].group) return true;
	return false;
}

bool OrderCalibration::wants_measure() const {
	if (best >= 0 || in_flight >= 0 || uses + 1 < 2) return false;
	for (int k = 0; k < n_trials; ++k)
		if (trials[k].samples < kOrderSamples) return true;
	return false;
}

void OrderCalibration::adopt(const OrderTrial &settled) {
	trials[1] = settled;
	trials[1].makespan = 0.0;
	trials[1].samples = 0;
	n_trials = 2;
	best = 1;
	in_flight = -1;
}

LaunchPlan OrderCalibration::plan(bool can_measure, const KernelChoice &scene) {
	++uses;
	LaunchPlan p;
	p.trial = best >= 0 ? best : 0;
	// (the first launch of a camera is never measured: a camera that is rendered once costs nothing)
	if (best < 0 && in_flight < 0 && uses >= 2 && can_measure)
		for (int k = 0; k < n_trials; ++k)
			if (trials[k].samples < kOrderSamples) {
				p.trial = k;
				p.measure = true;
				in_flight = k;
				break;
			}
	// the probing record times both kernels; everybody else renders with the one the scene's probe chose
	p.use_group = (probing() && (best >= 0 || p.measure)) ? trials[p.trial].group : scene.use_group;
	return p;
}

bool OrderCalibration::on_measured(const unsigned long long *rec, int tiles_y, int rot, bool may_probe, KernelChoice &scene) {
	if (in_flight < 0 || in_flight >= n_trials) return false;
	OrderTrial &t = trials[in_flight];
	const double span = std::max(1.0, measured_makespan(rec, tiles_y));
	t.makespan = t.samples == 0 ? span : std::min(t.makespan, span);
	++t.samples;
	if (in_flight == 0 && t.samples == 1) { // the rotation's records: make the candidates
		OrderTrial &model = trials[n_trials];
		model = OrderTrial();
		model.n = plan_order_from_measurement(rec, tiles_y, rot, model.b, model.c);
		if (model.n > 0) ++n_trials;
		OrderTrial &split = trials[n_trials];
		split = OrderTrial();
		split.n = split_hot_range(rec, tiles_y, rot, 0.45, 0.25, split.b, split.c);
		bool same = n_trials > 1 && split.n == model.n;
		for (int k = 0; same && k < 3; ++k) same = split.b[k] == model.b[k] && split.c[k] == model.c[k];
		if (split.n > 0 && !same) ++n_trials;
		if (may_probe && !scene.probed) { // the other kernel, under the rotation
			scene.probed = true;          // (this record's trials hold the probe: one camera per scene)
			OrderTrial &g = trials[n_trials];
			g = OrderTrial();
			g.group = true;
			++n_trials;
		}
	}
	in_flight = -1;
	for (int k = 0; k < n_trials; ++k)
		if (trials[k].samples < kOrderSamples) return false; // more to time
	// all timed: the shortest stays; another order must beat the rotation by 1 %, the other kernel the best order by 3 %
	int b = 0;
	for (int k = 1; k < n_trials; ++k)
		if (!trials[k].group && trials[k].makespan < 0.99 * trials[0].makespan && trials[k].makespan < trials[b].makespan) b = k;
	for (int k = 1; k < n_trials; ++k)
		if (trials[k].group && trials[k].makespan < 0.97 * trials[b].makespan) b = k;
	best = b;
	if (probing()) scene.use_group = trials[b].group; // (the probing record decides)
	return true;
}

int pick_fast_kernel(int forced, bool use_other, bool records_ok, KernelChoice &scene) {
	const int other = records_ok ? 2 : 0;
	if (forced == 1) return 0;
	if (forced == 3) return other;
	if (!use_other) return 1;
	if (scene.use_group) return scene.with_records == records_ok ? other : 1;
	scene.with_records = records_ok; // (a probe's launch)
	return other;
}

void fold_shadow_probe(KernelChoice &scene, const unsigned long long *leap_rec, const unsigned long long *group_rec, int tiles_y) {
	const double leap_span = std::max(1.0, measured_makespan(leap_rec, tiles_y));
	const double group_span = std::max(1.0, measured_makespan(group_rec, tiles_y));
	scene.use_group = group_span < 0.97 * leap_span;
}

} // namespace hmrm
