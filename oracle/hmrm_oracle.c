/*
 * hmrm_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or
 * called from the product path (heightmap-ray-marcher_amd/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and there
 * only as the checker / the reported CPU baseline.
 *
 * What it is: a plain-C, CPU restatement of the reference's per-pixel hot path
 * (ray generate -> AABB slab intersect -> fixed-step march -> shade) and of the
 * per-frame set-up and UpdateHeightmap that feed it.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference).
 *
 * PARITY UNPINNED (render path).  The reference holds no tests, golden images
 * or fixtures for this path, and it cannot be built in this image: main/hmap.cpp
 * needs SDL2 + SDL2_ttf, and main/hmap.cpp and every src/ file need glm
 * ("tested with 0.9.9.8", README.md:109), none of which is installed; writing
 * stand-in headers to force a build is not allowed.  The vector arithmetic that
 * lives in glm is therefore restated here from glm 0.9.9.8's published
 * formulas (see the g_* helpers below), anchored on the reference's own call
 * sites.  The image-decode / PNG-encode rows ARE pinned, separately, against
 * the reference's vendored stb sources built as oracle/_ref (see Makefile).
 *
 * Arithmetic contract (reference build: g++ -std=c++98, no -O, no -march,
 * makefile:20-23): IEEE-754 binary64, SSE2, no FMA contraction, libm = glibc.
 * Build this file with -ffp-contract=off and without -ffast-math / -march=native;
 * any -O level then gives the same bits.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double x, y, z; } v3;

/* ---- glm 0.9.9.8 semantics (third-party, not under /root/reference) --------
 * operator+/-: component-wise.  scalar*vec: (s*v.x, s*v.y, s*v.z).
 * dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z  (left-to-right).
 * cross: (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y).
 * normalize: v * inversesqrt(dot(v,v)),  inversesqrt(x) = 1/sqrt(x).
 * Call sites: src/Perspective.cpp:13-14,27; src/Orthographic.cpp:11. */
static v3 g_add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static v3 g_sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static v3 g_neg(v3 a)       { v3 r = { -a.x, -a.y, -a.z }; return r; }
static v3 g_smul(double s, v3 v) { v3 r = { s * v.x, s * v.y, s * v.z }; return r; }
/* Sensitivity switch (tests/glm_sensitivity.py only; 0 everywhere else): glm is not under
 * /root/reference, so the two formulas below are restated from its published source and nothing
 * the reference holds pins them.  The alternatives a different glm release or build could have
 * used are kept behind this switch to MEASURE how many pixels would move if the assumption were
 * wrong: bit 0 = normalize as v / sqrt(dot) instead of v * (1 / sqrt(dot)); bit 1 = dot summed as
 * x + (y + z) instead of (x + y) + z. */
static int g_glm_variant = 0;
void oracle_set_glm_variant(int v) { g_glm_variant = v; }
static double g_dot(v3 a, v3 b) {
	v3 t = { a.x * b.x, a.y * b.y, a.z * b.z };
	if (g_glm_variant & 2) return t.x + (t.y + t.z);
	return t.x + t.y + t.z;
}
static v3 g_cross(v3 x, v3 y) {
	v3 r = { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
	return r;
}
static v3 g_normalize(v3 v) {
	double inv = 1.0 / sqrt(g_dot(v, v));
	v3 r = { v.x * inv, v.y * inv, v.z * inv };
	if (g_glm_variant & 1) {
		const double len = sqrt(g_dot(v, v));
		v3 q = { v.x / len, v.y / len, v.z / len };
		return q;
	}
	return r;
}
/* dvec3 -> glm::vec3 (float) -> dvec3, per component, as the Orthographic ctor's
 * by-value `glm::vec3 lk, u` parameters do (src/Orthographic.cpp:3,5-6). */
static v3 g_float_roundtrip(v3 v) {
	v3 r = { (double)(float)v.x, (double)(float)v.y, (double)(float)v.z };
	return r;
}

/* ---- configuration the hot path reads (main/hmap.cpp:28-112 globals) ------ */
typedef struct {
	int32_t screen_width, screen_height;   /* :31-32 */
	int32_t image_plane;                   /* :104-107  1 persp, 2 spherical, 3 ortho */
	int32_t heightmap_width, heightmap_height; /* :54-55 (== colormap dims, :503-515) */
	double hfov, hang, vang;               /* radians, :35,80,85 */
	double cam_pos[3];                     /* :75 */
	double min_height, max_height;         /* :38-39 */
	double grid_width, step_dist, ortho_width; /* :65,68,98 */
	uint8_t bg_r, bg_g, bg_b;              /* :110-112 */
	uint8_t sampling;   /* 0 = nearest cell (the reference, :1001-1018); 1 = bilinear quality mode;
	                       2 = nearest cell with the hit thresholds rounded to float (neither is in the
	                       reference: build-side additions, defined below) */
	int64_t step_cap;   /* oracle-only guard: the reference loop is unbounded (:1000) */
} oracle_cfg;

typedef struct { v3 pos, dir; } ray_t;   /* src/Ray.hpp:6-9 */

/* One image plane, all three kinds (src/ImagePlane.hpp:6-15). */
typedef struct {
	int kind;
	v3 cam_pos;
	/* Perspective / Orthographic */
	v3 look, upper_left, plane_right, plane_down;
	/* Spherical */
	double hfov, vfov, ul_hang, ul_vang;
} plane_t;

/* src/Perspective.cpp:3-23 */
static void plane_perspective(plane_t *p, v3 pos, v3 look, v3 up, double hf, double ar) {
	double half_plane_width = tan(hf / 2.0);
	double half_plane_height = half_plane_width / ar;
	v3 right = g_normalize(g_cross(look, up));
	v3 fwd = g_add(pos, look);
	v3 upper_left  = g_sub(g_add(fwd, g_smul(half_plane_height, up)), g_smul(half_plane_width, right));
	v3 lower_left  = g_sub(g_sub(fwd, g_smul(half_plane_height, up)), g_smul(half_plane_width, right));
	v3 upper_right = g_add(g_add(fwd, g_smul(half_plane_height, up)), g_smul(half_plane_width, right));
	p->kind = 1;
	p->cam_pos = pos;
	p->upper_left = upper_left;
	p->plane_right = g_sub(upper_right, upper_left);
	p->plane_down = g_sub(lower_left, upper_left);
}

/* src/Spherical.cpp:3-15 */
static void plane_spherical(plane_t *p, v3 pos, double ha, double va, double hf, double ar) {
	p->kind = 2;
	p->cam_pos = pos;
	p->hfov = hf;
	p->vfov = hf / ar;
	p->ul_hang = ha + (hf / 2.0);
	p->ul_vang = va - (p->vfov / 2.0);
}

/* src/Orthographic.cpp:3-17 */
static void plane_orthographic(plane_t *p, v3 pos, v3 lk, v3 u, double ow, int sw, int sh) {
	v3 look = g_float_roundtrip(lk);
	v3 up = g_float_roundtrip(u);
	v3 right = g_cross(look, up);
	p->kind = 3;
	p->cam_pos = pos;
	p->look = look;
	p->upper_left = g_add(g_sub(pos, g_smul((sw / 2.0) * ow, right)), g_smul((sh / 2.0) * ow, up));
	p->plane_right = g_smul(sw * ow, right);
	p->plane_down = g_smul(sh * ow, g_neg(up));
}

/* src/Perspective.cpp:25-32, src/Spherical.cpp:17-31, src/Orthographic.cpp:19-25 */
static ray_t plane_get_ray(const plane_t *p, double w, double h) {
	ray_t r;
	if (p->kind == 1) {
		v3 on_plane = g_add(g_add(p->upper_left, g_smul(w, p->plane_right)), g_smul(h, p->plane_down));
		r.pos = p->cam_pos;
		r.dir = g_normalize(g_sub(on_plane, p->cam_pos));
	} else if (p->kind == 2) {
		double ha = p->ul_hang - w * p->hfov;
		double va = p->ul_vang + h * p->vfov;
		r.pos = p->cam_pos;
		r.dir.x = sin(va) * cos(ha);
		r.dir.y = sin(va) * sin(ha);
		r.dir.z = cos(va);
	} else {
		r.pos = g_add(g_add(p->upper_left, g_smul(w, p->plane_right)), g_smul(h, p->plane_down));
		r.dir = p->look;
	}
	return r;
}

/* src/AABB.cpp:49-77 */
static double aabb_distance(ray_t ray, v3 c0, v3 c1) {
	double lo = -INFINITY, hi = +INFINITY;
	double ro[3] = { ray.pos.x, ray.pos.y, ray.pos.z };
	double rd[3] = { ray.dir.x, ray.dir.y, ray.dir.z };
	double b0[3] = { c0.x, c0.y, c0.z };
	double b1[3] = { c1.x, c1.y, c1.z };
	int i;
	for (i = 0; i < 3; ++i) {
		double dim_lo = (b0[i] - ro[i]) / rd[i];
		double dim_hi = (b1[i] - ro[i]) / rd[i];
		if (dim_lo > dim_hi) { double t = dim_lo; dim_lo = dim_hi; dim_hi = t; }
		if (dim_hi < lo || dim_lo > hi) return INFINITY;
		if (dim_lo > lo) lo = dim_lo;
		if (dim_hi < hi) hi = dim_hi;
	}
	return (lo > hi) ? INFINITY : lo;
}

/* src/AABB.cpp:30-47 */
static int aabb_intersection(v3 *out, double *d_out, ray_t ray, v3 c0, v3 c1) {
	double d = aabb_distance(ray, c0, c1);
	*d_out = d;
	if (d == INFINITY) return 0;
	if (d < 0.0) return 0;
	out->x = ray.pos.x + d * ray.dir.x;
	out->y = ray.pos.y + d * ray.dir.y;
	out->z = ray.pos.z + d * ray.dir.z;
	return 1;
}

/* main/hmap.cpp:118-124 */
static double clampd(double v, double lo, double hi) {
	if (v < lo) return lo;
	if (v > hi) return hi;
	return v;
}

/* (int)double as the reference's x86-64 build does it (cvttsd2si): values that
 * do not fit, and NaN, give INT_MIN.  Written out so the oracle has no UB. */
static int trunc_to_int_x86(double v) {
	if (!(v > -2147483649.0 && v < 2147483648.0)) return (-2147483647 - 1);
	return (int)v;
}

/* main/hmap.cpp:171-191 UpdateHeightmap */
void oracle_update_heightmap(const uint8_t *base_rgb, int64_t num_pixels,
                             double lum_r, double lum_g, double lum_b,
                             double min_height, double max_height, double *heightmap_buf) {
	int64_t p;
	for (p = 0; p < num_pixels; ++p) {
		const uint8_t r = base_rgb[3 * p + 0];
		const uint8_t g = base_rgb[3 * p + 1];
		const uint8_t b = base_rgb[3 * p + 2];
		const double value = clampd((lum_r * r) + (lum_g * g) + (lum_b * b), 0.0, 255.0);
		heightmap_buf[p] = (value / 255.0) * (max_height - min_height) + min_height;
	}
}

/* main/hmap.cpp:131-133 */
double oracle_degrees_to_rads(double degrees) { return (degrees / 180.0) * M_PI; }

/* Per-frame set-up: main/hmap.cpp:661-672 (look/up), :952-965 (plane), :968-974 (box). */
static void frame_setup(const oracle_cfg *c, plane_t *pl, v3 *c0, v3 *c1) {
	v3 cam = { c->cam_pos[0], c->cam_pos[1], c->cam_pos[2] };
	v3 look = { sin(c->vang) * cos(c->hang), sin(c->vang) * sin(c->hang), cos(c->vang) };
	double up_vang = c->vang - (M_PI / 2.0);
	v3 up = { sin(up_vang) * cos(c->hang), sin(up_vang) * sin(c->hang), cos(up_vang) };
	memset(pl, 0, sizeof *pl);
	if (c->image_plane == 1)
		plane_perspective(pl, cam, look, up, c->hfov, (double)c->screen_width / c->screen_height);
	else if (c->image_plane == 2)
		plane_spherical(pl, cam, c->hang, c->vang, c->hfov, (double)c->screen_width / c->screen_height);
	else
		plane_orthographic(pl, cam, look, up, c->ortho_width, c->screen_width, c->screen_height);
	c0->x = 0.0; c0->y = 0.0; c0->z = c->min_height;
	c1->x = c0->x + c->heightmap_width * c->grid_width;
	c1->y = c0->y - c->heightmap_height * c->grid_width;
	c1->z = c->max_height;
}

/* ---- bilinear quality mode (build-side addition, no counterpart in the reference) ----------
 * The reference samples the nearest cell.  north_star asks for bilinear height / colour
 * sampling as an option; it is DEFINED here (and implemented identically on the GPU):
 *   cell (i,j) carries its value at its centre (i+0.5, j+0.5) in cell units;
 *   at cell coordinates (qx,qy):  u = qx-0.5, v = qy-0.5, tx = u-floor(u), ty = v-floor(v),
 *   neighbours i0 = clamp(floor(u)), i1 = clamp(floor(u)+1) (same for j), and
 *   f = a + ty*(b - a),  a = f00 + tx*(f10 - f00),  b = f01 + tx*(f11 - f01)   (fp64, no FMA).
 * Heights interpolate the thresholds heightmap_buf + c0.z; a ray hits when z < f.  Colours
 * interpolate each of R,G,B and round with floor(f + 0.5); the alpha-0 rule (hmap.cpp:1020)
 * keeps looking at the nearest cell.  Range test, stepping and everything else as the reference. */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

typedef struct { int i0, i1, j0, j1; double tx, ty; } bil_t;

static bil_t bil_setup(double qx, double qy, int w, int h) {
	bil_t b;
	double u = qx - 0.5, v = qy - 0.5;
	double fu = floor(u), fv = floor(v);
	b.tx = u - fu;
	b.ty = v - fv;
	b.i0 = clampi((int)fu, 0, w - 1);
	b.i1 = clampi((int)fu + 1, 0, w - 1);
	b.j0 = clampi((int)fv, 0, h - 1);
	b.j1 = clampi((int)fv + 1, 0, h - 1);
	return b;
}

static double bil_mix(const bil_t *b, double f00, double f10, double f01, double f11) {
	double a = f00 + b->tx * (f10 - f00);
	double c = f01 + b->tx * (f11 - f01);
	return a + b->ty * (c - a);
}

/* One pixel of main/hmap.cpp:982-1057.  Returns the number of height loads
 * (:1013-1014) executed = the "ray-steps" of BASELINE.md; -1 - steps when the
 * oracle-only step cap fired (the reference would not have terminated). */
static int64_t shade_pixel(const oracle_cfg *c, const plane_t *pl, v3 c0, v3 c1,
                           const double *heightmap_buf, const uint8_t *colormap_buf,
                           int w, int h, uint8_t rgba[4], double *entry_d, ray_t *ray_out) {
	int64_t steps = 0;
	int capped = 0;
	ray_t ray = plane_get_ray(pl, (double)w / (c->screen_width - 1), (double)h / (c->screen_height - 1));
	v3 int_point = { 0, 0, 0 };
	double d;
	int hit = aabb_intersection(&int_point, &d, ray, c0, c1);
	int real_hit = 0;
	if (entry_d) *entry_d = d;
	if (ray_out) *ray_out = ray;

	if (hit) {
		int_point = g_add(int_point, g_smul(c->grid_width * 0.01, ray.dir));
		for (;;) {
			int gridx = trunc_to_int_x86((int_point.x - c0.x) / c->grid_width);
			int gridy = trunc_to_int_x86(-(int_point.y - c0.y) / c->grid_width);
			double heightmap_z;
			if (gridx < 0 || gridy < 0 || gridx >= c->heightmap_width || gridy >= c->heightmap_height)
				break;
			if (steps >= c->step_cap) { capped = 1; break; }
			heightmap_z = heightmap_buf[gridx + (int64_t)gridy * c->heightmap_width];
			steps += 1;
			if (c->sampling == 1) {
				/* bilinear quality mode (see above); qx,qy are the doubles the reference truncates */
				const int W = c->heightmap_width, H = c->heightmap_height;
				const double qx = (int_point.x - c0.x) / c->grid_width, qy = -(int_point.y - c0.y) / c->grid_width;
				const bil_t b = bil_setup(qx, qy, W, H);
				const double t = bil_mix(&b, heightmap_buf[b.i0 + (int64_t)b.j0 * W] + c0.z, heightmap_buf[b.i1 + (int64_t)b.j0 * W] + c0.z,
				                         heightmap_buf[b.i0 + (int64_t)b.j1 * W] + c0.z, heightmap_buf[b.i1 + (int64_t)b.j1 * W] + c0.z);
				if (int_point.z < t) {
					int64_t red_index = (gridx + (int64_t)gridy * W) * 4;
					if (colormap_buf[red_index + 3] == 0) {
						rgba[0] = c->bg_r; rgba[1] = c->bg_g; rgba[2] = c->bg_b;
					} else {
						int k;
						for (k = 0; k < 3; ++k) {
							const double f = bil_mix(&b, colormap_buf[(b.i0 + (int64_t)b.j0 * W) * 4 + k], colormap_buf[(b.i1 + (int64_t)b.j0 * W) * 4 + k],
							                         colormap_buf[(b.i0 + (int64_t)b.j1 * W) * 4 + k], colormap_buf[(b.i1 + (int64_t)b.j1 * W) * 4 + k]);
							rgba[k] = (uint8_t)floor(clampd(f + 0.5, 0.0, 255.0));
						}
					}
					rgba[3] = 255;
					real_hit = 1;
					break;
				}
			} else if (c->sampling == 2 ? int_point.z < (double)(float)(heightmap_z + c0.z)
			                            : int_point.z < heightmap_z + c0.z) {
				/* sampling 2, "float heights" (north_star): the same loop, positions still fp64, but the
				 * threshold of a cell is (float)(heightmap_buf[i] + hmap_c0.z), round to nearest */
				int64_t red_index = (gridx + (int64_t)gridy * c->heightmap_width) * 4;
				if (colormap_buf[red_index + 3] == 0) {
					rgba[0] = c->bg_r; rgba[1] = c->bg_g; rgba[2] = c->bg_b;
				} else {
					rgba[0] = colormap_buf[red_index + 0];
					rgba[1] = colormap_buf[red_index + 1];
					rgba[2] = colormap_buf[red_index + 2];
				}
				rgba[3] = 255;
				real_hit = 1;
				break;
			}
			int_point = g_add(int_point, g_smul(c->step_dist, ray.dir));
		}
	}
	if (!real_hit) {
		if (ray.dir.z > 0.0) {
			/* std::pow(double,int) under -std=c++98 -> __builtin_powi -> z*z (:1044-1046) */
			const double r_ = 220.0 * (ray.dir.z * ray.dir.z) + c->bg_r;
			const double g_ = 240.0 * (ray.dir.z * ray.dir.z) + c->bg_g;
			const double b_ = 255.0 * ray.dir.z + c->bg_b;
			rgba[0] = (uint8_t)floor(clampd(r_, 0.0, 255.0));
			rgba[1] = (uint8_t)floor(clampd(g_, 0.0, 255.0));
			rgba[2] = (uint8_t)floor(clampd(b_, 0.0, 255.0));
		} else {
			rgba[0] = c->bg_r; rgba[1] = c->bg_g; rgba[2] = c->bg_b;
		}
		rgba[3] = 255;
	}
	return capped ? (-1 - steps) : steps;
}

/*
 * Render rows row_begin, row_begin+row_stride, ... < row_end of a full frame
 * (= `cycle 1`, main/hmap.cpp:976-983) into framebuf (RGBA8, stride W*4,
 * main/hmap.cpp:139-154).  Rows not visited are left untouched.
 * steps_out (optional, W*H int64): per-pixel height loads, or -1-steps if capped.
 * entry_d_out (optional, W*H doubles): distance() result per pixel.
 * Returns total height loads over the visited rows; *capped_out = number of
 * pixels where the oracle-only step cap fired.
 */
int64_t oracle_render(const oracle_cfg *c, const double *heightmap_buf, const uint8_t *colormap_buf,
                      uint8_t *framebuf, int64_t *steps_out, double *entry_d_out,
                      int row_begin, int row_end, int row_stride, int num_threads,
                      int64_t *capped_out) {
	plane_t pl;
	v3 c0, c1;
	int64_t total = 0, capped = 0;
	int h;
	frame_setup(c, &pl, &c0, &c1);
	if (row_stride < 1) row_stride = 1;
#ifdef _OPENMP
	if (num_threads > 0) omp_set_num_threads(num_threads);
#else
	(void)num_threads;
#endif
	#pragma omp parallel for schedule(dynamic, 1) reduction(+:total, capped)
	for (h = row_begin; h < row_end; h += row_stride) {
		int w;
		for (w = 0; w < c->screen_width; ++w) {
			const int64_t p = w + (int64_t)h * c->screen_width;
			double d;
			int64_t s = shade_pixel(c, &pl, c0, c1, heightmap_buf, colormap_buf, w, h,
			                        framebuf + p * 4, &d, NULL);
			if (steps_out) steps_out[p] = s;
			if (entry_d_out) entry_d_out[p] = d;
			if (s < 0) { capped += 1; s = -1 - s; }
			total += s;
		}
	}
	if (capped_out) *capped_out = capped;
	return total;
}

/* Per-ray probe for the debug hooks: ray (pos, dir) and distance() for one pixel. */
void oracle_probe_ray(const oracle_cfg *c, int w, int h, double pos[3], double dir[3], double *entry_d) {
	plane_t pl;
	v3 c0, c1;
	ray_t ray;
	frame_setup(c, &pl, &c0, &c1);
	ray = plane_get_ray(&pl, (double)w / (c->screen_width - 1), (double)h / (c->screen_height - 1));
	pos[0] = ray.pos.x; pos[1] = ray.pos.y; pos[2] = ray.pos.z;
	dir[0] = ray.dir.x; dir[1] = ray.dir.y; dir[2] = ray.dir.z;
	*entry_d = aabb_distance(ray, c0, c1);
}

int oracle_max_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* The per-frame record (hmap.cpp:661-672,:952-974) in the layout the product's
 * hmrm_debug_frame test hook uses: cam[3], upper_left[3], plane_right[3],
 * plane_down[3], look[3], c0[3], c1[3], nudge.  Fields a projection does not
 * have are 0.  For spherical also: out[22..25] = hfov, vfov, ul_hang, ul_vang. */
void oracle_frame_record(const oracle_cfg *c, double out[26]) {
	plane_t pl;
	v3 c0, c1;
	int i = 0;
	frame_setup(c, &pl, &c0, &c1);
	out[i++] = pl.cam_pos.x; out[i++] = pl.cam_pos.y; out[i++] = pl.cam_pos.z;
	out[i++] = pl.upper_left.x; out[i++] = pl.upper_left.y; out[i++] = pl.upper_left.z;
	out[i++] = pl.plane_right.x; out[i++] = pl.plane_right.y; out[i++] = pl.plane_right.z;
	out[i++] = pl.plane_down.x; out[i++] = pl.plane_down.y; out[i++] = pl.plane_down.z;
	out[i++] = pl.look.x; out[i++] = pl.look.y; out[i++] = pl.look.z;
	out[i++] = c0.x; out[i++] = c0.y; out[i++] = c0.z;
	out[i++] = c1.x; out[i++] = c1.y; out[i++] = c1.z;
	out[i++] = c->grid_width * 0.01;
	out[i++] = pl.hfov; out[i++] = pl.vfov; out[i++] = pl.ul_hang; out[i++] = pl.ul_vang;
}
