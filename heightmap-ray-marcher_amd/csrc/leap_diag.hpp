// leap_diag.hpp -- the instrumented view of k_render_fast's loop (tools only: tools/attempt_diag.py,
// iter_map.py, wave_time.py, refresh_diag.py select what is counted with HMRM_DIAG_ITERS -> DevFrame::diag_mode).
// LoopDiag<false> is empty: nothing of this reaches the production instantiations.
#pragma once
#include "device_common.hpp"

namespace hmrm {

template <bool STATS>
struct LoopDiag;

template <>
struct LoopDiag<false> {
	__device__ __forceinline__ void start() {}
	__device__ __forceinline__ void begin_trip() {}
	__device__ __forceinline__ void on_attempt() {}
	__device__ __forceinline__ void on_refresh_check(const DevFrame &, bool, bool, bool) {}
	__device__ __forceinline__ void on_landing_refused(const DevFrame &, bool, bool, bool, bool) {}
	__device__ __forceinline__ void on_attempt_done(const DevFrame &, bool, bool, bool, bool, bool, bool, bool, int, int) {}
	__device__ __forceinline__ void on_trip(const DevFrame &, bool, bool) {}
	__device__ __forceinline__ void on_bounds(bool, bool) {}
	__device__ __forceinline__ void on_group() {}
	__device__ __forceinline__ void load_begin(const DevFrame &, int) {}
	__device__ __forceinline__ void load_end(const DevFrame &, int, float &) {}
	__device__ __forceinline__ void load_end(const DevFrame &, int, double &, double &, double &, double &) {}
	__device__ __forceinline__ uint32_t pixel_value(const DevFrame &, unsigned long long) { return 0u; }
	__device__ __forceinline__ void publish(const StatsOut &, const DevFrame &) {}
};

template <>
struct LoopDiag<true> {
	uint32_t attempts = 0, leaps = 0, groups = 0;
	unsigned long long leaped = 0;
	unsigned long long x0 = 0, x1 = 0, x2 = 0, x3 = 0; // meaning depends on diag_mode
	unsigned long long t_start = 0, t_load = 0;
	bool attempted = false; // this trip ran the attempt block
	bool jumped = false, at_binade = false; // this trip's attempt: succeeded / was limited by a binade's end

	__device__ __forceinline__ void start() { t_start = __builtin_amdgcn_s_memtime(); }
	__device__ __forceinline__ void begin_trip() { attempted = jumped = at_binade = false; }
	__device__ __forceinline__ void on_bounds(bool ok, bool binade_bound) {
		jumped = ok;
		at_binade = binade_bound;
	}
	__device__ __forceinline__ void on_attempt() {
		++attempts;
		attempted = true;
	}
	__device__ __forceinline__ static bool wave_leader(unsigned long long active) {
		return (int)(threadIdx.x & 63u) == __ffsll((long long)active) - 1;
	}
	// mode 16: how often does a WAVE run a refresh block, and for how many lanes?
	__device__ __forceinline__ void on_refresh_check(const DevFrame &f, bool need_x, bool need_y, bool need_z) {
		if (f.diag_mode != 16) return;
		const unsigned long long act = __ballot(true);
		const unsigned long long bx = __ballot(need_x), by = __ballot(need_y), bz = __ballot(need_z);
		if (wave_leader(act)) {
			x0 += 1u;                                                     // attempt blocks run by waves
			x1 += (bx ? 1u : 0u) + (by ? 1u : 0u) + (bz ? 1u : 0u);       // refresh blocks run by waves
			x2 += (unsigned)(__popcll(bx) + __popcll(by) + __popcll(bz)); // lanes that needed them
			x3 += (unsigned)__popcll(act);                                // lanes in the attempt blocks
		}
	}
	// mode 10: which landing test refused a jump that the estimates allowed
	__device__ __forceinline__ void on_landing_refused(const DevFrame &f, bool refused, bool in_grid, bool in_window, bool high) {
		if (f.diag_mode != 10 || !refused) return;
		x0 += !in_grid ? 1u : 0u;
		x1 += (in_grid && !in_window) ? 1u : 0u;
		x2 += (in_grid && in_window && !high) ? 1u : 0u;
		x3 += (in_grid && in_window && high) ? 1u : 0u; // binade / boundary tests
	}
	// modes 4-7, 9, 11: why attempts fail and at which level; always: jumps and the steps they cover
	__device__ __forceinline__ void on_attempt_done(const DevFrame &f, bool inb0, bool exact, bool above, bool short_jump,
	                                                bool z_bound, bool can, bool ok, int n, int lev) {
		const bool ie = inb0 && exact;
		auto by_level = [&](bool c, unsigned w) {
			x0 += (c && lev == 0) ? w : 0u;
			x1 += (c && lev == 1) ? w : 0u;
			x2 += (c && lev == 2) ? w : 0u;
			x3 += (c && lev >= 3) ? w : 0u;
		};
		if (f.diag_mode == 4) {
			x0 += (ie && !above) ? 1u : 0u;
			x1 += (ie && above && short_jump && z_bound) ? 1u : 0u;
			x2 += (ie && above && short_jump && !z_bound) ? 1u : 0u;
			x3 += (can && !ok) ? 1u : 0u;
		}
		if (f.diag_mode == 9) by_level(ie && above && short_jump && !z_bound, 1u);
		if (f.diag_mode == 11) by_level(ie && !above && lev <= 3, 1u);
		if (f.diag_mode == 5) by_level(lev <= 3, 1u);
		if (f.diag_mode == 6) by_level(ok && lev <= 3, 1u);
		if (f.diag_mode == 7) by_level(ok && lev <= 3, (unsigned)n);
		if (f.diag_mode == 21) { // what limited a successful jump: the window's maximum (z room) or something else; and their steps
			x0 += (ok && z_bound) ? 1u : 0u;
			x1 += (ok && !z_bound) ? 1u : 0u;
			x2 += (ok && z_bound) ? (unsigned)n : 0u;
			x3 += (ok && !z_bound) ? (unsigned)n : 0u;
		}
		if (f.diag_mode == 22) by_level(ok && z_bound, 1u); // z-limited jumps by level
		if (f.diag_mode == 23) { // attempts and jumps of the levels above 3
			x0 += (lev >= 3 && lev <= 4) ? 1u : 0u;
			x1 += (lev >= 5 && lev <= 6) ? 1u : 0u;
			x2 += (lev >= 7) ? 1u : 0u;
			x3 += (ok && lev >= 7) ? (unsigned)n : 0u;
		}
		leaped += ok ? (unsigned)n : 0u;
		leaps += ok ? 1u : 0u;
	}
	// modes 12-15: wave-level view of the loop -- who runs which block, with how many useful lanes
	__device__ __forceinline__ void on_trip(const DevFrame &f, bool leap_enabled, bool skip_group) {
		if (f.diag_mode == 20 && !skip_group) { // mode 20: what came before a group (per lane)
			x0 += (attempted && jumped && at_binade) ? 1u : 0u;   // a jump that stopped at a binade's end
			x1 += (attempted && !jumped && at_binade) ? 1u : 0u;  // an attempt with no binade room for a jump
			x2 += !attempted ? 1u : 0u;                           // no attempt (pause after failures)
			x3 += (attempted && !at_binade) ? 1u : 0u;            // any other attempt
		}
		if (f.diag_mode < 12 || f.diag_mode > 15) return;
		const unsigned long long act = __ballot(true);
		const unsigned long long att = __ballot(leap_enabled && attempted);
		const unsigned long long grp = __ballot(!skip_group);
		if (!wave_leader(act)) return;
		const int nact = __popcll(act);
		auto by_share = [&](int k) { // share of the active lanes taking part: < 1/8, < 1/4, < 1/2, >= 1/2
			x0 += (8 * k < nact) ? 1u : 0u;
			x1 += (8 * k >= nact && 4 * k < nact) ? 1u : 0u;
			x2 += (4 * k >= nact && 2 * k < nact) ? 1u : 0u;
			x3 += (2 * k >= nact) ? 1u : 0u;
		};
		if (f.diag_mode == 12) {
			x0 += 1u;
			x1 += att ? 1u : 0u;
			x2 += grp ? 1u : 0u;
			x3 += (unsigned)nact;
		} else if (f.diag_mode == 13) {
			x0 += att ? 64u : 0u;               // lane slots spent in attempt blocks
			x1 += (unsigned)__popcll(att);      // ... of which useful
			x2 += grp ? 64u : 0u;               // lane slots spent in group blocks
			x3 += (unsigned)__popcll(grp);      // ... of which useful
		} else if (f.diag_mode == 14) {
			if (att) by_share(__popcll(att));
		} else if (grp) {
			by_share(__popcll(grp));
		}
	}
	__device__ __forceinline__ void on_group() { ++groups; }
	// modes 17 (pyramid look-up) / 18 (the group's height loads): cycles a wave waits for the data, per pixel
	// (the wait is forced right after the load: the instrumented kernel then overlaps nothing with it)
	__device__ __forceinline__ void load_begin(const DevFrame &f, int mode) {
		if (f.diag_mode == mode) t_load = __builtin_amdgcn_s_memtime();
	}
	__device__ __forceinline__ void load_end(const DevFrame &f, int mode, float &v) {
		if (f.diag_mode != mode) return;
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
		x0 += __builtin_amdgcn_s_memtime() - t_load;
	}
	__device__ __forceinline__ void load_end(const DevFrame &f, int mode, double &a, double &b, double &c, double &d) {
		if (f.diag_mode != mode) return;
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
		x0 += __builtin_amdgcn_s_memtime() - t_load;
	}
	// what the per-pixel output holds instead of the step count (modes 1-3)
	__device__ __forceinline__ uint32_t pixel_value(const DevFrame &f, unsigned long long steps) {
		if (f.diag_mode == 1) return ((attempts > 0xffffu ? 0xffffu : attempts) << 16) | (groups > 0xffffu ? 0xffffu : groups);
		if (f.diag_mode == 2) return (uint32_t)(__builtin_amdgcn_s_memtime() - t_start); // wave cycles
		if (f.diag_mode == 3) return (uint32_t)t_start;
		if (f.diag_mode == 17 || f.diag_mode == 18) return (uint32_t)x0;
		if (f.diag_mode == 19) return ((attempts > 0xffffu ? 0xffffu : attempts) << 16) | (leaps > 0xffffu ? 0xffffu : leaps);
		return steps > 0xffffffffull ? 0xffffffffu : (uint32_t)steps;
	}
	__device__ __forceinline__ void publish(const StatsOut &st, const DevFrame &f) {
		unsigned long long a = attempts, l = leaps, g = groups, s = leaped;
		if (f.diag_mode >= 4) { a = x0; l = x1; g = x2; s = x3; }
		for (int off = 32; off > 0; off >>= 1) {
			a += __shfl_xor(a, off);
			l += __shfl_xor(l, off);
			g += __shfl_xor(g, off);
			s += __shfl_xor(s, off);
		}
		if ((threadIdx.x & 63) == 0) {
			if (a) atomicAdd(&st.counters[4], a);
			if (l) atomicAdd(&st.counters[5], l);
			if (g) atomicAdd(&st.counters[6], g);
			if (s) atomicAdd(&st.counters[7], s);
		}
	}
};

} // namespace hmrm
