// row_cost.cpp -- scheduling aid for the launch order (api.cpp choose_tile_rot): per block of screen rows the
// longest in-box ray, in steps, over a few sample columns.  Approximate arithmetic on purpose -- nothing here
// reaches a pixel -- so this file is compiled with the compiler's builtins (camera.cpp is not: -fno-builtin keeps
// its libm calls literal, and made fmin / fmax / sqrt here 150 us of function calls per 4K camera).
#include "frame.hpp"

#include <cmath>

namespace hmrm {

void estimate_row_costs(const DevFrame &f, const double *col_cos_ha, const double *col_sin_ha,
                        const double *row_sin_va, const double *row_cos_va, int rows_per_sample, float *out) {
	const int W = f.screen_w, H = f.screen_h;
	const int n = (H + rows_per_sample - 1) / rows_per_sample;
	constexpr int kCols = 9;
	int col[kCols];
	for (int c = 0; c < kCols; ++c) col[c] = (int)((int64_t)(W - 1) * c / (kCols - 1));
	const double inv_step = 1.0 / std::fabs(f.step_dist);
	for (int k = 0; k < n; ++k) {
		int py = k * rows_per_sample + rows_per_sample / 2;
		if (py > H - 1) py = H - 1;
		const double h = H > 1 ? (double)py / (H - 1) : 0.0;
		double best = 0.0;
		for (int c = 0; c < kCols; ++c) {
			const int px = col[c];
			double o[3], d[3];
			if (f.projection == 2) {
				o[0] = f.cam[0]; o[1] = f.cam[1]; o[2] = f.cam[2];
				d[0] = row_sin_va[py] * col_cos_ha[px];
				d[1] = row_sin_va[py] * col_sin_ha[px];
				d[2] = row_cos_va[py];
			} else {
				const double w = W > 1 ? (double)px / (W - 1) : 0.0;
				double p[3];
				for (int i = 0; i < 3; ++i) p[i] = f.upper_left[i] + w * f.plane_right[i] + h * f.plane_down[i];
				if (f.projection == 1) {
					for (int i = 0; i < 3; ++i) { o[i] = f.cam[i]; d[i] = p[i] - f.cam[i]; }
					const double inv = 1.0 / std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
					for (int i = 0; i < 3; ++i) d[i] *= inv;
				} else {
					for (int i = 0; i < 3; ++i) { o[i] = p[i]; d[i] = f.look[i]; }
				}
			}
			double lo = 0.0, hi = HUGE_VAL; // the part of the ray in front of its origin
			for (int i = 0; i < 3; ++i) {
				const double inv = 1.0 / d[i];
				const double t0 = (f.c0[i] - o[i]) * inv, t1 = (f.c1[i] - o[i]) * inv;
				// (plain comparisons: gcc keeps fmin / fmax as libm calls without -ffinite-math-only; a NaN parameter
				// -- 0 * inf on an axis the ray is parallel to -- just leaves lo / hi as they were)
				const double a = t0 < t1 ? t0 : t1, b = t0 < t1 ? t1 : t0;
				lo = a > lo ? a : lo;
				hi = b < hi ? b : hi;
			}
			const double steps = (hi - lo) * inv_step;
			if (steps > best && std::isfinite(steps)) best = steps;
		}
		out[k] = (float)best;
	}
}

} // namespace hmrm
