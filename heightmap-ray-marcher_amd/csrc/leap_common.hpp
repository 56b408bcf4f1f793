// leap_common.hpp -- the exact-leap arithmetic of the production kernel (render_fast.hip): cell
// coordinates, the saturating (int) cast and the per-axis "same binade => p_k = p_0 + k*delta exactly"
// state.  The argument is in render_fast.hip's header;
// tests/test_leap_math.py checks the arithmetic model by brute force.  Include only from .hip files
// compiled with -ffp-contract=off.
#pragma once
#include "device_common.hpp"

#pragma clang fp contract(off)

namespace hmrm {

constexpr int kTopLevel = kMipLevels; // whole-map level: the one element of the pyramid's top plane

__device__ __forceinline__ uint32_t hi32(double v) { return (uint32_t)((unsigned long long)__double_as_longlong(v) >> 32); }
__device__ __forceinline__ uint32_t lo32(double v) { return (uint32_t)(unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ double f64_from_hi(uint32_t hi) { return __longlong_as_double((long long)((unsigned long long)hi << 32)); }

// Cell coordinate q with trunc(q) == trunc(v / grid_width); v is x - c0.x or -(y - c0.y) with
// c0.x = c0.y = 0.0 (hmap.cpp:968: v - 0.0 == v for every v, so the subtraction is elided).
//   GWM 0: grid_width == 1.0   -> q = v
//   GWM 1: grid_width = 2^k    -> q = v * 2^-k                 (exact reciprocal)
//   GWM 2: any grid_width      -> q' = v * fl(1/gw), which differs from the correctly rounded
//          v/gw by < 2^-50 relative.  Unless q' lies within 2^-20 of an integer both truncate to
//          the same cell (for |q'| >= 2^28 both are far outside any map, whatever they truncate
//          to); `near` collects that rare case and the caller then divides for real.  What is returned is
//          q' + 2^-20 (one fused multiply-add: this is an approximation anyway), so that the test is one
//          v_fract and one compare -- near  <=>  fract(q' + 2^-20) < 2^-19  <=>  q' in [k - 2^-20, k + 2^-20) --
//          and off that neighbourhood trunc(q' + 2^-20) == trunc(q') on either side of zero.  Callers only
//          truncate the value (or replace it by the true quotient).
template <int GWM>
__device__ __forceinline__ double cell_coord_fast(double v, const DevFrame &f, bool &near) {
	if (GWM == 0) return v;
	if (GWM == 2) {
		const double q = __builtin_fma(v, f.inv_grid_width, 0x1p-20);
		near = near || !(__builtin_amdgcn_fract(q) >= 0x1p-19); // (NaN: near)
		return q;
	}
	return v * f.inv_grid_width;
}

// (int)q exactly as the reference's x86 build evaluates it for the range test of hmap.cpp:1001-
// 1011, without C++'s undefined behaviour for out-of-range values: v_cvt_i32_f64 truncates
// toward zero and saturates, so q in (-1,0) gives 0 (inside, as on the CPU), q <= -1 a negative
// index, |q| >= 2^31 and +-inf give INT_MIN / INT_MAX (outside; cvttsd2si gives INT_MIN: also
// outside).  Only NaN would differ (0 here, INT_MIN there), and a position cannot be NaN inside
// the loop: the entry point is checked once, and sums of finite or infinite steps of one sign
// never produce NaN.
__device__ __forceinline__ int cvt_i32_sat(double q) {
	int r;
	asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(q));
	return r;
}
// (int)(-q): the negation rides on the instruction's source modifier (exact, like the reference's
// unary minus before the cast, hmap.cpp:1003)
__device__ __forceinline__ int cvt_i32_sat_neg(double q) {
	int r;
	asm("v_cvt_i32_f64_e64 %0, -%1" : "=v"(r) : "v"(q));
	return r;
}
// row * width + col for 0 <= row, col, width < 2^24 with the product below 2^32 (cell and window
// indices of a map: api.cpp refuses larger dimensions for this kernel): one full-rate v_mad_u32_u24
// instead of the quarter-rate 32-bit multiply.  Garbage in, garbage out for lanes outside the grid
// (their index is replaced by 0 before use).
__device__ __forceinline__ int index_2d(int row, int width, int col) {
	return (int)(__umul24((unsigned)row, (unsigned)width) + (unsigned)col);
}

// HMRM_STEPS_LEFT (default 1): how long a coordinate stays inside its binade is counted, not re-estimated and
// re-verified at every landing.  At a refresh the number of further steps that certainly stay strictly inside the
// binade is estimated from the reciprocal, shortened by two and VERIFIED once (the position that many steps ahead
// has the same sign and exponent and is off the boundary; every position before it then is too, the motion being
// monotonic).  From there on every position of the ray in this binade is p_0 + j * delta exactly, j = steps taken
// (leaped or real), so `left` just counts down; a jump of n <= left steps needs no exponent / mantissa test at its
// landing point, and left < 0 says the binade has been left: refresh.  0 = round 2's scheme (key compare, three
// binade rooms and three landing tests per attempt), kept for A/B runs.
#ifndef HMRM_STEPS_LEFT
#define HMRM_STEPS_LEFT 1
#endif
constexpr bool kStepsLeft = HMRM_STEPS_LEFT != 0;
// HMRM_CROSS (default 1; needs HMRM_STEPS_LEFT): a jump ends with ONE real step p + s after its multiplied ones.  When a
// binade's end cut the jump short that step is the one that crosses it, so no group of real steps has to be marched just
// to get a coordinate into its next binade.  The count below then is taken as tight as the estimate allows (no two
// steps of allowance -- its far end is verified anyway), so that the step after the last counted one does cross
// (tests/test_leap_math.py: in 99.99 % of random cases).  0 = jumps of multiplied steps only, kept for A/B runs.
#ifndef HMRM_CROSS
#define HMRM_CROSS 1
#endif
constexpr bool kCross = kStepsLeft && HMRM_CROSS != 0;

// Exact-stepping state of one coordinate inside its current binade.
struct Axis {
	double delta;  // p_{k+1} - p_k for every p of the binade (valid iff key matches)
	double lim;    // binade boundary the coordinate is moving towards
	double rdel;   // ~1/delta (signed); (lim - p) * rdel estimates the steps left
	uint32_t key;  // sign+exponent bits (hi32 >> 20) the above was measured for
	int left;      // kStepsLeft: further steps certain to stay strictly inside the binade; < 0: refresh needed
};

// Measure delta at p (see file header) from TWO real steps.  Off a rounding tie the
// increment is the same for every p of the binade.  On an exact tie (s = q*u + u/2)
// round-to-even makes every result an even multiple of u, so from the first step on
// the increment is constant as well (q or q+1 by the parity of q); only a start value
// of the wrong parity steps differently once -- which shows as two unequal increments
// and is rejected here (the next group of real steps lands on the steady parity).
// Invalid (key = ~0) also when the two steps leave the binade or change sign, or p is
// tiny / non-finite.
__device__ __forceinline__ void axis_refresh(Axis &a, double p, double s) {
	// straight-line on purpose (bitwise tests, selects): a wave runs this whenever any lane
	// crosses a binade, so exec-mask branches here would cost every lane of the wave
	const double p1 = p + s, p2 = p1 + s;
	const uint32_t hp = hi32(p), hp1 = hi32(p1), hp2 = hi32(p2);
	const uint32_t e = (hp >> 20) & 0x7ffu;
	const double d = p1 - p;                  // exact: multiples of u, |d| < 2^53 u
	const int ok = (int)((((hp ^ hp1) | (hp ^ hp2)) >> 20) == 0u) & (int)(e - 128u <= 1772u) & (int)((p2 - p1) == d);
	a.key = ok ? (hp >> 20) : 0xffffffffu;
	a.delta = d;
	// |p| grows (d has p's sign): the limit is 2^(E+1), else 2^E; either way with p's sign.
	// One integer add on the high word (E <= 1900, no overflow into the sign).
	const uint32_t away = (((hi32(d) ^ hp) >> 31) ^ 1u) << 20;
	const double lim = f64_from_hi((hp & 0xfff00000u) + away);
	const bool still = d == 0.0;              // the coordinate never moves (s == 0 or absorbed): unlimited room
	a.lim = still ? p + 1.0 : lim;
	a.rdel = still ? 0x1p40 : __builtin_amdgcn_rcp(d);
	if (kStepsLeft) {
		// steps that stay inside: the estimate (the reciprocal is good to 2^-24: shortened by 2^-22 of itself, so it is
		// below the true quotient and its integer part is at most the count wanted; without kCross two steps fewer),
		// at most 2^30, verified at its far end
		int k = cvt_i32_sat((lim - p) * a.rdel * (1.0 - 0x1p-22)) - (kCross ? 0 : 2);
		k = k < 0 ? 0 : (k > (1 << 30) ? (1 << 30) : k);
		const double pk = p + (double)k * d; // (exact: a multiple of u below 2^53 u)
		const uint32_t hk = hi32(pk);
		const int inside = (int)((hk >> 20) == (hp >> 20)) & (int)(((hk & 0xfffffu) | lo32(pk)) != 0u);
		a.left = ok ? (still ? (1 << 30) : (inside ? k : 0)) : -1;
	}
}

// p_n = p + n*delta is trustworthy iff it is still in p's binade with the same sign,
// and -- when the coordinate moves -- not exactly on a binade boundary (moving towards
// zero, the step that produced it could have rounded on the finer grid below 2^E).
__device__ __forceinline__ bool axis_landing_ok(const Axis &a, double pn) {
	// (bitwise on purpose: one straight-line expression instead of a chain of exec-mask branches)
	const uint32_t h = hi32(pn);
	const int same_binade = (h >> 20) == a.key ? 1 : 0;
	const int off_boundary = ((h & 0xfffffu) | lo32(pn)) != 0u ? 1 : 0;
	const int still = a.delta == 0.0 ? 1 : 0;
	return (same_binade & (still | off_boundary)) != 0;
}


} // namespace hmrm
