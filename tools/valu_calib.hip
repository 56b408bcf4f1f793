// valu_calib.hip -- calibration of the SQ VALU counters on gfx950 (tools only, not part of the product).
//
// bench.py's roofline figure for the render kernel is "VALU issue": the share of SIMD cycles in
// which the vector pipe is busy.  rocprofv3 offers SQ_ACTIVE_INST_VALU and SQ_INSTS_VALU[_*]; what
// one unit of each means for a given instruction class has to be measured on a kernel whose pipe
// occupancy is known.  Every kernel below is a long chain of ONE instruction class on 8 independent
// accumulators per lane (no memory traffic in the loop), launched so that every SIMD holds `waves`
// waves; the pipe is then saturated and
//     cycles per wave-instruction (pipe) = SIMDs * duration * clock / wave-instructions issued.
// Run under rocprofv3 (kernel trace for the durations, then PMC passes), e.g. tools/profile_round.sh.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/valu_calib tools/valu_calib.hip
//   /tmp/valu_calib [waves_per_simd=8] [iters=20000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#pragma clang fp contract(off)

#define CK(e)                                                                       \
	do {                                                                            \
		hipError_t e_ = (e);                                                        \
		if (e_ != hipSuccess) {                                                     \
			fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                 \
			return 1;                                                               \
		}                                                                           \
	} while (0)

constexpr int kAcc = 8;
constexpr int kUnroll = 16; // instructions per accumulator per loop trip

// 0: v_add_f64   1: v_mul_f64   2: v_fma_f64   3: v_add_f32   4: v_add_u32   5: v_cmp_lt_f64 + v_cndmask (x2)
// 6: v_cvt_i32_f64   7: v_rcp_f64   8..19: see main()
template <int MODE>
__global__ __launch_bounds__(256) void k_calib(double *out, int iters, double seed) {
	double a[kAcc];
	float b[kAcc];
	unsigned c[kAcc];
	for (int i = 0; i < kAcc; ++i) {
		a[i] = seed + (double)(threadIdx.x + i) * 0x1p-20;
		b[i] = (float)a[i];
		c[i] = threadIdx.x * 2654435761u + (unsigned)i;
	}
	const double s = 1.0 + seed * 0x1p-30;
	const float sf = (float)s;
	const unsigned inv = (unsigned)(seed * 77.0) + threadIdx.x;
	if (MODE == 20) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(inv), "v"((unsigned)iters) : "vcc");
	if (MODE == 21 || MODE == 28 || MODE == 29) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(inv), "v"((unsigned)iters) : "s20", "s21");
	if (MODE == 27) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(inv), "v"((unsigned)iters) : "vcc");
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
			for (int i = 0; i < kAcc; ++i) {
				// (inline asm: exactly one instruction of the named class per accumulator step, nothing folded or packed)
				if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
				if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
				if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
				if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(sf));
				if (MODE == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 5) {
					unsigned lo = c[i], hi = c[i] ^ 1u;
					asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc"
					             : "+v"(lo), "+v"(hi) : "v"(a[i]), "v"(s), "v"(c[(i + 1) % kAcc]) : "vcc");
					c[i] = lo ^ hi;
				}
				if (MODE == 6) {
					int r;
					asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(a[i]));
					c[i] ^= (unsigned)r;
				}
				if (MODE == 7) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
				if (MODE == 8) asm volatile("v_cmp_lt_f64 s[20:21], %0, %1" : : "v"(a[i]), "v"(s) : "s20", "s21");
				if (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 11) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
				if (MODE == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 13) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));
				if (MODE == 14) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(c[i]));
				if (MODE == 15) asm volatile("v_fract_f64 %0, %0" : "+v"(a[i]));
				if (MODE == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 17) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(c[i]), "v"(c[(i + 1) % kAcc]) : "s20", "s21");
				if (MODE == 18) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i]));
				if (MODE == 19) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s) : "vcc");
				// selects: condition in vcc (VOP2) or in an SGPR pair (VOP3), second source a loop invariant
				if (MODE == 20) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[i]) : "v"(inv));
				if (MODE == 21) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(c[i]) : "v"(inv));
				if (MODE == 22) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]), "v"(inv) : "vcc");
				if (MODE == 23) asm volatile("v_cmp_lt_u32 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]), "v"(inv) : "s20", "s21");
				if (MODE == 25) asm volatile("v_cmp_lt_u32 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(c[i]), "+v"(c[(i + 3) % kAcc]) : "v"(c[(i + 1) % kAcc]), "v"(inv) : "vcc");
				if (MODE == 26) asm volatile("v_cmp_lt_f64 s[20:21], %2, %3\n\tv_cndmask_b32_e64 %0, %0, %4, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %4, s[20:21]" : "+v"(c[i]), "+v"(c[(i + 3) % kAcc]) : "v"(a[i]), "v"(s), "v"(inv) : "s20", "s21");
				if (MODE == 27) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(c[i]) : "v"(inv));
				if (MODE == 28) asm volatile("s_and_b64 vcc, s[20:21], exec\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[i]) : "v"(inv) : "vcc", "scc");
				if (MODE == 29) asm volatile("s_and_b64 s[22:23], s[20:21], exec\n\tv_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(c[i]) : "v"(inv) : "s22", "s23", "scc");
				if (MODE == 30) asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_add_u32 %1, %1, %4\n\tv_cndmask_b32 %1, %1, %4, vcc" : "+v"(c[i]), "+v"(c[(i + 3) % kAcc]) : "v"(a[i]), "v"(s), "v"(inv) : "vcc");
				// 32-bit integer forms the render loop uses besides plain VOP2 add / logic (tools/isa_cost.py's classes)
				if (MODE == 31) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 32) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 33) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 34) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 35) asm volatile("v_max_i32 %0, %0, %1" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 36) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(c[i]));
				if (MODE == 37) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
				if (MODE == 38) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(a[i]));
				if (MODE == 39) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(c[i]), "v"(c[(i + 1) % kAcc]) : "vcc");
				if (MODE == 40) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(s));
				if (MODE == 41) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c[i]) : "v"(c[(i + 1) % kAcc]));
				if (MODE == 42) asm volatile("v_cvt_i32_f64_e64 %0, -%1" : "=v"(c[i]) : "v"(a[i]));
				if (MODE == 24) asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc" : "+v"(c[i]), "+v"(c[(i + 3) % kAcc]) : "v"(a[i]), "v"(s), "v"(inv) : "vcc");
			}
		}
	}
	double t = 0.0;
	for (int i = 0; i < kAcc; ++i) t += a[i] + (double)b[i] + (double)c[i];
	if (t == 123.456) out[0] = t; // keeps the chains alive without a store in practice
}

template <int MODE>
static int run(const char *what, double *d_out, int blocks, int iters, double insts_per_acc_step) {
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k_calib<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters / 10, 1.0); // warm-up
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL(k_calib<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0);
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms = 0.f;
	CK(hipEventElapsedTime(&ms, e0, e1));
	const double waves = (double)blocks * 4.0;
	const double winsts = waves * (double)iters * kUnroll * kAcc * insts_per_acc_step;
	// SIMD-cycles per wave-instruction at an assumed 2.4 GHz: SIMDs * t * f / instructions
	const double cyc = 1024.0 * (ms * 1e-3) * 2.4e9 / winsts;
	printf("%-28s blocks %5d  %8.3f ms  %.3e wave-insts  %.2f SIMD-cycles/inst at 2.4 GHz\n", what, blocks, ms, winsts,
	       cyc);
	return 0;
}

int main(int argc, char **argv) {
	const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
	const int iters = argc > 2 ? atoi(argv[2]) : 20000;
	const int blocks = 256 * waves_per_simd; // 256 CUs, one 4-wave block per SIMD quartet and wave slot
	double *d_out = nullptr;
	CK(hipMalloc((void **)&d_out, 64));
	printf("valu_calib: %d waves per SIMD, %d loop trips of %d instructions per accumulator\n", waves_per_simd, iters,
	       kUnroll);
	if (run<0>("v_add_f64", d_out, blocks, iters, 1)) return 1;
	if (run<1>("v_mul_f64", d_out, blocks, iters, 1)) return 1;
	if (run<2>("v_fma_f64", d_out, blocks, iters, 1)) return 1;
	if (run<3>("v_add_f32", d_out, blocks, iters, 1)) return 1;
	if (run<4>("v_add_u32", d_out, blocks, iters, 1)) return 1;
	if (run<5>("v_cmp_f64+2 v_cndmask+2 v_xor", d_out, blocks, iters / 2, 5)) return 1;
	if (run<6>("v_cvt_i32_f64 (+v_xor)", d_out, blocks, iters, 2)) return 1;
	if (run<7>("v_rcp_f64", d_out, blocks, iters / 2, 1)) return 1;
	if (run<8>("v_cmp_lt_f64 -> sgpr", d_out, blocks, iters, 1)) return 1;
	if (run<9>("v_cndmask_b32", d_out, blocks, iters, 1)) return 1;
	if (run<10>("v_and_b32", d_out, blocks, iters, 1)) return 1;
	if (run<11>("v_min_f64", d_out, blocks, iters, 1)) return 1;
	if (run<12>("v_mul_lo_u32", d_out, blocks, iters / 2, 1)) return 1;
	if (run<13>("v_lshlrev_b64", d_out, blocks, iters, 1)) return 1;
	if (run<14>("v_cvt_f64_i32", d_out, blocks, iters, 1)) return 1;
	if (run<15>("v_fract_f64", d_out, blocks, iters, 1)) return 1;
	if (run<16>("v_mov_b32", d_out, blocks, iters, 1)) return 1;
	if (run<17>("v_cmp_lt_u32 -> sgpr", d_out, blocks, iters, 1)) return 1;
	if (run<18>("v_sqrt_f64", d_out, blocks, iters / 2, 1)) return 1;
	if (run<19>("v_div_fmas_f64", d_out, blocks, iters, 1)) return 1;
	if (run<20>("v_cndmask_b32 (vcc, const)", d_out, blocks, iters, 1)) return 1;
	if (run<21>("v_cndmask_b32 (sgpr pair)", d_out, blocks, iters, 1)) return 1;
	if (run<22>("v_cmp_u32->vcc + cndmask", d_out, blocks, iters, 2)) return 1;
	if (run<23>("v_cmp_u32->sgpr + cndmask", d_out, blocks, iters, 2)) return 1;
	if (run<24>("v_cmp_f64->vcc + 2 cndmask", d_out, blocks, iters / 2, 3)) return 1;
	if (run<25>("v_cmp_u32->vcc + 2 cndmask", d_out, blocks, iters / 2, 3)) return 1;
	if (run<26>("v_cmp_f64->sgpr + 2 cndmask_e64", d_out, blocks, iters / 2, 3)) return 1;
	if (run<27>("v_cndmask_b32_e64 (vcc, const)", d_out, blocks, iters, 1)) return 1;
	if (run<28>("s_and vcc + cndmask e32", d_out, blocks, iters, 1)) return 1;
	if (run<29>("s_and sgpr + cndmask_e64", d_out, blocks, iters, 1)) return 1;
	if (run<30>("cmp_f64->vcc, cnd, add, cnd", d_out, blocks, iters / 2, 4)) return 1;
	if (run<31>("v_add3_u32", d_out, blocks, iters, 1)) return 1;
	if (run<32>("v_lshl_add_u32", d_out, blocks, iters, 1)) return 1;
	if (run<33>("v_mad_u32_u24", d_out, blocks, iters, 1)) return 1;
	if (run<34>("v_and_or_b32", d_out, blocks, iters, 1)) return 1;
	if (run<35>("v_max_i32", d_out, blocks, iters, 1)) return 1;
	if (run<36>("v_ashrrev_i32", d_out, blocks, iters, 1)) return 1;
	if (run<37>("v_cvt_f64_f32", d_out, blocks, iters, 1)) return 1;
	if (run<38>("v_lshl_add_u64", d_out, blocks, iters, 1)) return 1;
	if (run<39>("v_cmp_eq_u32 -> vcc", d_out, blocks, iters, 1)) return 1;
	if (run<40>("v_mov_b64", d_out, blocks, iters, 1)) return 1;
	if (run<41>("v_xor_b32", d_out, blocks, iters, 1)) return 1;
	if (run<42>("v_cvt_i32_f64 (-src)", d_out, blocks, iters, 1)) return 1;
	CK(hipDeviceSynchronize());
	return 0;
}
