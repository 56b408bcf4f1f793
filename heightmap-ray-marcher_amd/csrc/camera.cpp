// camera.cpp -- per-frame host set-up: camera basis, image plane, box corners.
//
// Mirrors what the reference does on the host once per frame
// (main/hmap.cpp:661-672 look/up; :952-965 ImagePlane construction =
// src/Perspective.cpp:3-23, src/Spherical.cpp:3-15, src/Orthographic.cpp:3-17;
// :968-974 hmap_c0/hmap_c1).  All libm calls (sin, cos, tan) stay on the host so
// they are glibc's, as in the reference; the device never evaluates a
// transcendental.  Must be compiled with -ffp-contract=off -fno-builtin and
// without -ffast-math: the value of every expression below is part of the
// pixel-exactness contract.
//
// glm semantics (glm 0.9.9.8, not vendored by the reference) are restated in
// Vec3's operators: component-wise +,-; s*v; cross; dot = (x+y)+z;
// normalize = v * (1/sqrt(dot(v,v))).
#include "frame.hpp"

#include <cmath>
#include <cstring>

namespace hmrm {
namespace {

struct Vec3 {
	double x, y, z;
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return Vec3{-a.x, -a.y, -a.z}; }
inline Vec3 operator*(double s, Vec3 v) { return Vec3{s * v.x, s * v.y, s * v.z}; }
inline Vec3 cross(Vec3 a, Vec3 b) {
	return Vec3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline double dot(Vec3 a, Vec3 b) {
	const double px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z;
	return px + py + pz;
}
inline Vec3 normalize(Vec3 v) {
	const double inv = 1.0 / std::sqrt(dot(v, v));
	return Vec3{v.x * inv, v.y * inv, v.z * inv};
}
// Orthographic's constructor takes look/up as glm::vec3 (float) by value
// (src/Orthographic.cpp:3) although the caller holds dvec3 (hmap.cpp:963).
inline Vec3 through_float(Vec3 v) {
	return Vec3{(double)(float)v.x, (double)(float)v.y, (double)(float)v.z};
}
inline void store(double dst[3], Vec3 v) {
	dst[0] = v.x;
	dst[1] = v.y;
	dst[2] = v.z;
}

} // namespace

// Spherical.cpp:18-25 is separable: ha (and so cos ha, sin ha) depends on the column only, va on the row only.
// Columns [begin, end) of the W-entry tables; glibc's cos / sin, one call per value as in the reference.
void fill_col_tables(const HostCamera &cam, int32_t begin, int32_t end, double *col_cos_ha, double *col_sin_ha) {
	const double hfov = cam.hfov;
	const double ul_hang = cam.hang + (hfov / 2.0);            // Spherical.cpp:12
	for (int32_t px = begin; px < end; ++px) {
		const double w = (double)px / (cam.width - 1);          // hmap.cpp:986
		const double ha = ul_hang - w * hfov;                   // Spherical.cpp:18
		col_cos_ha[px] = std::cos(ha);
		col_sin_ha[px] = std::sin(ha);
	}
}

// Rows [begin, end) of the H-entry tables.
void fill_row_tables(const HostCamera &cam, int32_t begin, int32_t end, double *row_sin_va, double *row_cos_va) {
	const double aspect = (double)cam.width / cam.height;      // hmap.cpp:955-956
	const double vfov = cam.hfov / aspect;                     // Spherical.cpp:11
	const double ul_vang = cam.vang - (vfov / 2.0);            // Spherical.cpp:13
	for (int32_t py = begin; py < end; ++py) {
		const double h = (double)py / (cam.height - 1);         // hmap.cpp:987
		const double va = ul_vang + h * vfov;                   // Spherical.cpp:19
		row_sin_va[py] = std::sin(va);
		row_cos_va[py] = std::cos(va);
	}
}

void build_frame(const HostCamera &cam, int32_t map_w, int32_t map_h, double min_height,
                 double max_height, double grid_width, DevFrame *f, double *col_cos_ha,
                 double *col_sin_ha, double *row_sin_va, double *row_cos_va) {
	*f = DevFrame();
	f->screen_w = cam.width;
	f->screen_h = cam.height;
	f->projection = cam.projection;
	f->map_w = map_w;
	f->map_h = map_h;
	f->bg[0] = cam.bg_r;
	f->bg[1] = cam.bg_g;
	f->bg[2] = cam.bg_b;
	f->bg[3] = 255;
	f->sampling = cam.sampling;

	const Vec3 pos{cam.pos[0], cam.pos[1], cam.pos[2]};
	store(f->cam, pos);

	// hmap.cpp:661-672
	const double hang = cam.hang, vang = cam.vang;
	const Vec3 look{std::sin(vang) * std::cos(hang), std::sin(vang) * std::sin(hang), std::cos(vang)};
	const double up_vang = vang - (M_PI / 2.0);
	const Vec3 up{std::sin(up_vang) * std::cos(hang), std::sin(up_vang) * std::sin(hang),
	              std::cos(up_vang)};

	const double aspect = (double)cam.width / cam.height; // hmap.cpp:955-956

	if (cam.projection == 1) {
		// src/Perspective.cpp:10-22
		const double half_w = std::tan(cam.hfov / 2.0);
		const double half_h = half_w / aspect;
		const Vec3 right = normalize(cross(look, up));
		const Vec3 centre = pos + look;
		const Vec3 ul = (centre + half_h * up) - half_w * right;
		const Vec3 ll = (centre - half_h * up) - half_w * right;
		const Vec3 ur = (centre + half_h * up) + half_w * right;
		store(f->upper_left, ul);
		store(f->plane_right, ur - ul);
		store(f->plane_down, ll - ul);
	} else if (cam.projection == 2) {
		// src/Spherical.cpp:11-14, then the separable halves of :18-25 (callers that fill the tables
		// in pieces -- api.cpp, on several host threads -- pass null here and call the two fills themselves)
		if (col_cos_ha && col_sin_ha) fill_col_tables(cam, 0, cam.width, col_cos_ha, col_sin_ha);
		if (row_sin_va && row_cos_va) fill_row_tables(cam, 0, cam.height, row_sin_va, row_cos_va);
	} else {
		// src/Orthographic.cpp:5-16
		const Vec3 lk = through_float(look);
		const Vec3 u = through_float(up);
		const Vec3 right = cross(lk, u);
		const double ow = cam.ortho_width;
		const Vec3 ul = (pos - ((cam.width / 2.0) * ow) * right) + ((cam.height / 2.0) * ow) * u;
		store(f->look, lk);
		store(f->upper_left, ul);
		store(f->plane_right, (cam.width * ow) * right);
		store(f->plane_down, (cam.height * ow) * (-u));
	}

	// hmap.cpp:968-974
	f->c0[0] = 0.0;
	f->c0[1] = 0.0;
	f->c0[2] = min_height;
	f->c1[0] = f->c0[0] + map_w * grid_width;
	f->c1[1] = f->c0[1] - map_h * grid_width;
	f->c1[2] = max_height;

	for (int i = 0; i < 3; ++i) {
		auto high = [](double v) {
			uint64_t b;
			memcpy(&b, &v, sizeof b);
			return (uint32_t)(b >> 32);
		};
		const uint32_t h0 = high(f->c0[i] - cam.pos[i]), h1 = high(f->c1[i] - cam.pos[i]);
		const bool moderate = (((h0 >> 20) & 0x7ffu) - 523u) < 1000u && (((h1 >> 20) & 0x7ffu) - 523u) < 1000u;
		f->box_side_known[i] = (moderate && ((h0 ^ h1) >> 31) == 0u) ? 1 : 0;
		f->box_side[i] = f->box_side_known[i] ? h0 : 0u;
	}

	f->grid_width = grid_width;
	f->nudge = grid_width * 0.01; // hmap.cpp:998
	f->step_dist = cam.step_dist;

	// x / 2^k == x * 2^-k bit for bit (both are the correctly rounded value of the
	// same real number) as long as 2^-k is representable: normal power of two.
	int e = 0;
	const double m = std::frexp(grid_width, &e);
	f->grid_pow2 = (std::isfinite(grid_width) && m == 0.5 && e > -1000 && e < 1000) ? 1 : 0;
	f->inv_grid_width = 1.0 / grid_width;
	f->grid_mode = grid_width == 1.0 ? 0 : (f->grid_pow2 ? 1 : 2);

	// Scheduling hints only (never change a pixel).  The finest pyramid level has 4-cell windows
	// placed every 2 cells, i.e. 2..4 cells of room; with steps longer than about a third of a
	// cell that is too few steps for a jump to pay for its bookkeeping, and the 16-cell level is
	// the finest one used (measured: C3 at 0.25 cells/step wants the 4-cell level, C2/C4/C5 at 0.5
	// run 5-7 % faster without it).  After a refused attempt at the finest level the ray marches
	// real steps before it looks again: one group when windows are 4 cells (16 steps) wide, seven
	// when they are 16 cells (32 steps) wide (tools/min_level_exp.py, round 3: a pause of 6 extra groups
	// against 3 is C5 -1.9 %, C2 -1.1 %, C4 -0.4 %; 8 the same).
	const double cells_per_step = std::fabs(cam.step_dist / grid_width);
	f->min_window = cells_per_step > 0.35 ? 16 : 4;
	f->finest_pause = cells_per_step > 0.35 ? 6 : 0;
	f->min_level = 0; // (api.cpp turns min_window into a level of the pyramid it built)
}

} // namespace hmrm
