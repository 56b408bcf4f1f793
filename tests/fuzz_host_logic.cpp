// ASan / UBSan fuzz of the host-only logic on the render path (test infrastructure; built and run by
// tests/test_host_logic_fuzz.py with g++ -fsanitize=address,undefined, no GPU, no HIP):
//
//   config.cpp        ConsumeConfigStream's grammar (main/hmap.cpp:309-520) fed mutated, truncated and hostile streams;
//   camera.cpp        build_frame / fill_*_tables (src/{Perspective,Spherical,Orthographic}.cpp constructors) and
//   row_cost.cpp      estimate_row_costs for degenerate and non-finite cameras (W = 1, H = 1, NaN angles, huge positions);
//   launch_order.cpp  orders from measured records -- which the DEVICE writes, so nothing about them may be trusted:
//                     every RowMap they lead to must hand every tile row to exactly one grid row, whatever the records
//                     hold -- and the calibration's state machine under random event sequences.
//
// usage: fuzz_host_logic <rounds> ; prints "host logic ok: ..." and returns 0, or aborts under the sanitizers.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "config.hpp"
#include "frame.hpp"
#include "launch_order.hpp"

namespace {

uint64_t g_state = 0x243f6a8885a308d3ull;
uint64_t rnd() {
	g_state += 0x9e3779b97f4a7c15ull;
	uint64_t z = g_state;
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return z ^ (z >> 31);
}
int rnd_int(int lo, int hi) { return lo + (int)(rnd() % (uint64_t)(hi - lo + 1)); } // inclusive
double rnd_unit() { return (double)(rnd() >> 11) * 0x1p-53; }

double hostile_double() {
	static const double k[] = {0.0, -0.0, 1.0, -1.0, 1e-320, 1e308, -1e308, 0x1p-1074, 3.141592653589793, 1.5707963267948966,
	                           std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity(),
	                           std::numeric_limits<double>::quiet_NaN(), 1e18, -1e18, 0.05, 0.01, 255.0, 2147483648.0};
	if (rnd() % 3 == 0) return k[rnd() % (sizeof(k) / sizeof(k[0]))];
	return (rnd_unit() - 0.5) * std::ldexp(1.0, rnd_int(-30, 40));
}

// ---- config -----------------------------------------------------------------------------------------------------------
const char *kKeys[] = {"heightmap", "colormap", "print", "resolution", "hfov", "hang", "vang", "pos", "pos_x", "pos_y", "pos_z",
                       "min_height", "max_height", "lum", "lum_norm", "lum_r", "lum_g", "lum_b", "grid_width", "ortho_width", "step_dist",
                       "bg_color", "cycle", "mouse_sens", "scroll_sens", "move", "recording_frame_count", "projection", "output",
                       "sampling", "heights", "devices", "record"};
const char *kTokens[] = {
    "perspective", "spherical", "orthographic", "orbit", "off", "nearest", "bilinear", "f32", "f64", "bits", "unknown_key",
    "0", "1", "2", "3", "-1", "800", "600", "1e999", "-1e999", "nan", "inf", "-inf", "0x10", "1e-400", "4294967296", "-2147483649",
    "99999999999999999999999999", "3.14.15", "--5", "+", "-", ".", "e", "\"quoted path\"", "/nonexistent/map.png",
    "/dev/null", "#", "=", "resolution=3", "\x01\x02", "\xff\xfe", "0.05", "90", "180", "360", "47", "255", "256"};

std::string g_map_dir; // argv[2]: where the harness may write three tiny PPM maps (4x3, 4x3 and 5x3)

void write_ppm(const std::string &path, int w, int h) {
	FILE *fp = fopen(path.c_str(), "wb");
	if (!fp) return;
	fprintf(fp, "P6\n%d %d\n255\n", w, h);
	for (int i = 0; i < 3 * w * h; ++i) fputc((int)(rnd() & 0xff), fp);
	fclose(fp);
}

std::string random_config() {
	std::string s;
	const int n = rnd_int(0, 60);
	for (int i = 0; i < n; ++i) {
		const unsigned r = (unsigned)(rnd() % 10);
		if (!g_map_dir.empty() && rnd() % 8 == 0) {
			static const char *maps[] = {"/h.ppm", "/c.ppm", "/d.ppm"};
			s += rnd() % 2 ? "heightmap " : "colormap ";
			s += g_map_dir + maps[rnd() % 8 == 0 ? 2 : rnd() % 2];
		} else if (r < 3) s += kKeys[rnd() % (sizeof(kKeys) / sizeof(kKeys[0]))];
		else if (r < 7) s += kTokens[rnd() % (sizeof(kTokens) / sizeof(kTokens[0]))];
		else if (r < 9) {
			char buf[64];
			snprintf(buf, sizeof(buf), "%.17g", hostile_double());
			s += buf;
		} else {
			const int len = rnd_int(0, 12);
			for (int j = 0; j < len; ++j) s += (char)(rnd() & 0xff);
		}
		static const char *seps[] = {" ", "\n", "\t", "  ", "\r\n", ""};
		s += seps[rnd() % 6];
	}
	return s;
}

int fuzz_config(int rounds) {
	int accepted = 0;
	for (int i = 0; i < rounds; ++i) {
		hmrm::Config cfg;
		const int streams = rnd_int(1, 3); // (the reference consumes several files into the same globals)
		bool ok = true;
		for (int k = 0; k < streams && ok; ++k) {
			std::istringstream in(random_config());
			std::string fatal;
			ok = cfg.consume(in, &fatal);
			if (!ok && fatal.empty()) {
				fprintf(stderr, "config: refused without a message\n");
				return -1;
			}
		}
		accepted += ok ? 1 : 0;
	}
	return accepted;
}

// ---- cameras ----------------------------------------------------------------------------------------------------------
int fuzz_cameras(int rounds) {
	int frames = 0;
	for (int i = 0; i < rounds; ++i) {
		hmrm::HostCamera cam;
		std::memset(&cam, 0, sizeof(cam));
		static const int dims[] = {1, 2, 3, 7, 8, 9, 16, 63, 64, 65, 100, 333, 1024};
		cam.width = dims[rnd() % 13];
		cam.height = dims[rnd() % 13];
		cam.projection = rnd_int(1, 3);
		cam.bg_r = (uint8_t)rnd(); cam.bg_g = (uint8_t)rnd(); cam.bg_b = (uint8_t)rnd();
		cam.sampling = (uint8_t)rnd_int(0, 2);
		const bool hostile = rnd() % 2 == 0;
		cam.hfov = hostile ? hostile_double() : rnd_unit() * 3.2;
		cam.hang = hostile ? hostile_double() : (rnd_unit() - 0.5) * 7.0;
		cam.vang = hostile ? hostile_double() : rnd_unit() * 3.2;
		for (int k = 0; k < 3; ++k) cam.pos[k] = hostile ? hostile_double() : (rnd_unit() - 0.5) * 400.0;
		cam.ortho_width = hostile ? hostile_double() : rnd_unit();
		cam.step_dist = hostile ? hostile_double() : rnd_unit();
		const int map_w = rnd_int(1, 5000), map_h = rnd_int(1, 5000);
		const double gw = rnd() % 4 ? 1.0 : (hostile ? hostile_double() : rnd_unit() + 1e-3);
		const double lo = hostile ? hostile_double() : (rnd_unit() - 0.5) * 10.0;
		const double hi = hostile ? hostile_double() : lo + rnd_unit() * 50.0;
		std::vector<double> cc((size_t)cam.width), cs((size_t)cam.width), rs((size_t)cam.height), rc((size_t)cam.height);
		hmrm::DevFrame f;
		std::memset(&f, 0, sizeof(f));
		const bool tables = cam.projection == 2;
		hmrm::build_frame(cam, map_w, map_h, lo, hi, gw, &f, tables ? cc.data() : nullptr, tables ? cs.data() : nullptr,
		                  tables ? rs.data() : nullptr, tables ? rc.data() : nullptr);
		if (f.screen_w != cam.width || f.screen_h != cam.height || f.map_w != map_w || f.map_h != map_h) {
			fprintf(stderr, "camera: frame record does not carry the sizes it was given\n");
			return -1;
		}
		if (tables) { // the pool fills sub-ranges: any split must stay inside the arrays
			const int a = rnd_int(0, cam.width), b = rnd_int(0, cam.height);
			hmrm::fill_col_tables(cam, 0, a, cc.data(), cs.data());
			hmrm::fill_col_tables(cam, a, cam.width, cc.data(), cs.data());
			hmrm::fill_row_tables(cam, 0, b, rs.data(), rc.data());
			hmrm::fill_row_tables(cam, b, cam.height, rs.data(), rc.data());
		}
		f.thr_max = hostile ? hostile_double() : hi + lo;
		const int per = hmrm::kCostRows;
		std::vector<float> cost((size_t)((cam.height + per - 1) / per) + 1, -7.0f);
		hmrm::estimate_row_costs(f, cc.data(), cs.data(), rs.data(), rc.data(), per, cost.data());
		if (cost.back() != -7.0f) {
			fprintf(stderr, "camera: estimate_row_costs wrote past ceil(h / rows_per_sample) entries\n");
			return -1;
		}
		cost.pop_back();
		hmrm::RowMap rows;
		std::memset(&rows, 0, sizeof(rows));
		rows.local_rows = cam.height;
		const int tile_h = rnd() % 2 ? 16 : 8;
		const int tiles_y = (cam.height + tile_h - 1) / tile_h;
		const int rot = hmrm::choose_tile_rot(true, cost, rows, tile_h);
		if (rot < 0 || rot >= tiles_y) {
			fprintf(stderr, "camera: rotation %d outside [0, %d)\n", rot, tiles_y);
			return -1;
		}
		++frames;
	}
	return frames;
}

// ---- launch orders ----------------------------------------------------------------------------------------------------
// The kernel's reading of a RowMap (frame.hpp): grid row j belongs to the last piece k with j >= seg_first[k - 1] and renders
// tile row (j + seg_delta[k]) mod tiles_y.
bool order_is_a_permutation(const hmrm::RowMap &r, int tiles_y) {
	std::vector<int> seen((size_t)tiles_y, 0);
	for (int j = 0; j < tiles_y; ++j) {
		int k = 0;
		for (int p = 0; p < 3; ++p)
			if (j >= r.seg_first[p]) k = p + 1;
		const long long t = ((long long)j + r.seg_delta[k]) % tiles_y;
		if (t < 0 || t >= tiles_y || seen[(size_t)t]++) return false;
	}
	return true;
}

void random_records(std::vector<unsigned long long> &rec, int tiles_y) {
	const unsigned kind = (unsigned)(rnd() % 6);
	const unsigned long long base = kind == 5 ? ~0ull - 1000 : rnd() % (1ull << 40);
	for (int t = 0; t < tiles_y; ++t) {
		unsigned long long start, longest;
		switch (kind) {
		case 0: start = base + (unsigned long long)t * 13; longest = 100 + rnd() % 20000; break; // plausible
		case 1: start = rnd() % 3 ? base + rnd() % 100000 : 0; longest = rnd() % 50000; break;   // unordered, holes
		case 2: start = 0; longest = 0; break;                                                   // nothing recorded
		case 3: start = rnd(); longest = rnd(); break;                                           // garbage
		case 4: start = base; longest = t == tiles_y / 2 ? 1000000 : 1; break;                   // one spike, equal starts
		default: start = base + (unsigned long long)t; longest = ~0ull - rnd() % 1000; break;    // near the top of the range
		}
		rec[(size_t)(2 * t)] = start;
		rec[(size_t)(2 * t + 1)] = longest;
	}
}

int fuzz_orders(int rounds) {
	int orders = 0;
	for (int i = 0; i < rounds; ++i) {
		const int tiles_y = rnd() % 8 == 0 ? rnd_int(1, 11) : rnd_int(12, 512);
		std::vector<unsigned long long> rec((size_t)(2 * tiles_y));
		random_records(rec, tiles_y);
		const int rot = rnd() % 16 == 0 ? rnd_int(-3, tiles_y + 3) : rnd_int(0, tiles_y - 1);
		const double span = hmrm::measured_makespan(rec.data(), tiles_y);
		if (!(span >= 0.0)) {
			fprintf(stderr, "orders: makespan %g\n", span);
			return -1;
		}
		int b[3] = {0, 0, 0}, c[3] = {0, 0, 0};
		for (int which = 0; which < 3; ++which) {
			int n;
			if (which == 0) n = hmrm::plan_order_from_measurement(rec.data(), tiles_y, rot, b, c);
			else if (which == 1) n = hmrm::split_hot_range(rec.data(), tiles_y, rot, rnd_unit(), rnd_unit(), b, c);
			else { // pieces nobody validated: set_tile_order has to
				n = rnd_int(-1, 3);
				for (int k = 0; k < 3; ++k) { b[k] = rnd_int(-2, tiles_y + 2); c[k] = rnd_int(-2, tiles_y + 2); }
			}
			if (n < 0 || n > 3) {
				if (which < 2) { fprintf(stderr, "orders: %d pieces\n", n); return -1; }
				n = 0;
			}
			hmrm::RowMap r;
			std::memset(&r, 0, sizeof(r));
			const int safe_rot = rot < 0 || rot >= tiles_y ? 0 : rot; // (api.cpp passes choose_tile_rot's result: inside the frame)
			hmrm::set_tile_order(&r, tiles_y, safe_rot, n, b, c);
			if (!order_is_a_permutation(r, tiles_y)) {
				fprintf(stderr, "orders: tiles_y %d rot %d source %d pieces %d [%d+%d %d+%d %d+%d]: not a permutation\n", tiles_y, rot,
				        which, n, b[0], c[0], b[1], c[1], b[2], c[2]);
				return -1;
			}
			++orders;
		}
	}
	return orders;
}

int fuzz_calibration(int rounds) {
	int settled = 0;
	for (int i = 0; i < rounds; ++i) {
		hmrm::KernelChoice scene;
		hmrm::OrderCalibration cal[2]; // two cameras of one scene
		const int tiles_y = rnd_int(12, 300);
		std::vector<unsigned long long> rec((size_t)(2 * tiles_y));
		const int rot = rnd_int(0, tiles_y - 1);
		const int events = rnd_int(1, 80);
		for (int e = 0; e < events; ++e) {
			hmrm::OrderCalibration &c = cal[rnd() % 2];
			const unsigned what = (unsigned)(rnd() % 16);
			if (what == 0) c.drop_in_flight();
			else if (what == 1) c.reset();
			else if (what == 2) scene.reset();
			else if (what == 3 && cal[0].best >= 0) cal[1].adopt(cal[0].trials[cal[0].best]);
			else if (c.in_flight >= 0 && what < 10) {
				random_records(rec, tiles_y);
				const int before = c.in_flight;
				if (before >= c.n_trials) { fprintf(stderr, "calibration: trial %d of %d in flight\n", before, c.n_trials); return -1; }
				if (c.on_measured(rec.data(), tiles_y, rot, rnd() % 2 == 0, scene)) {
					if (c.best < 0 || c.best >= c.n_trials) { fprintf(stderr, "calibration: settled on trial %d of %d\n", c.best, c.n_trials); return -1; }
					++settled;
				}
				if (c.in_flight >= 0) { fprintf(stderr, "calibration: still in flight after its report\n"); return -1; }
			} else {
				(void)c.wants_measure();
				(void)c.probing();
				const hmrm::LaunchPlan p = c.plan(rnd() % 4 != 0, scene);
				if (p.trial < 0 || p.trial >= c.n_trials || c.n_trials > 4) { fprintf(stderr, "calibration: plan for trial %d of %d\n", p.trial, c.n_trials); return -1; }
				if (p.measure != (c.in_flight == p.trial && p.measure)) { fprintf(stderr, "calibration: a measured plan is not in flight\n"); return -1; }
				hmrm::RowMap r;
				std::memset(&r, 0, sizeof(r));
				const hmrm::OrderTrial &t = c.trials[p.trial];
				hmrm::set_tile_order(&r, tiles_y, rot, t.n, t.b, t.c);
				if (!order_is_a_permutation(r, tiles_y)) { fprintf(stderr, "calibration: a trial's order is not a permutation\n"); return -1; }
				int with = 0;
				hmrm::KernelChoice probe = scene;
				const int k = hmrm::pick_fast_kernel((int)(rnd() % 4), p.use_group, rnd() % 2 == 0, probe);
				if (k < 0 || k > 2) { fprintf(stderr, "calibration: kernel %d\n", k); return -1; }
				(void)with;
			}
		}
		if (rnd() % 2) { // the shadow probe's verdict from two sets of records
			std::vector<unsigned long long> other((size_t)(2 * tiles_y));
			random_records(rec, tiles_y);
			random_records(other, tiles_y);
			hmrm::fold_shadow_probe(scene, rec.data(), other.data(), tiles_y);
		}
	}
	return settled;
}

} // namespace

int main(int argc, char **argv) {
	const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
	if (argc > 2) {
		g_map_dir = argv[2];
		write_ppm(g_map_dir + "/h.ppm", 4, 3);
		write_ppm(g_map_dir + "/c.ppm", 4, 3);
		write_ppm(g_map_dir + "/d.ppm", 5, 3);
	}
	const int a = fuzz_config(rounds);
	if (a < 0) return 1;
	const int f = fuzz_cameras(rounds);
	if (f < 0) return 1;
	const int o = fuzz_orders(rounds * 2);
	if (o < 0) return 1;
	const int s = fuzz_calibration(rounds);
	if (s < 0) return 1;
	printf("host logic ok: %d config streams (%d accepted), %d cameras, %d launch orders, %d calibration runs (%d settled)\n", rounds, a, f,
	       o, rounds, s);
	return 0;
}
