/*
 * nmf_oracle.c -- CPU restatement of nmfgpu's factorisation path.
 *
 * STATUS: TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker / the timed
 * CPU baseline -- never as a compute path of the shipped engine.  The engine under
 * nmfgpu_amd/ neither links nor loads it and fails loudly without its HIP code objects.
 *
 * PARITY UNPINNED (by the reference's arithmetic): nmfgpu v0.2.3 has no tests, no golden
 * vectors and no CPU path, and its arithmetic lives in closed CUDA libraries (cuBLAS gemm/
 * syrk/symm/trsm, cuSOLVER geqrf/ormqr, cuRAND, cuSPARSE; "CUDA >= 7.0", CMakeLists.txt:57-59)
 * that cannot be built or run here.  What this oracle restates is the
 * reference's operation ORDER and SEMANTICS, file:line cited at every function; what pins it:
 *   - the two reference translation units that do build here (source/nmf/Algorithm.cpp,
 *     source/nmf/Summary.cpp, compiled by oracle/Makefile into oracle/_ref/) for the seed
 *     stream and the best-run bookkeeping, with their outputs committed under tests/golden/;
 *   - known-answer properties of the mathematics (tests/test_oracle.py);
 *   - the reference example's own host-side check (example/main.cpp:133-146).
 *
 * Plain C99 + OpenMP; float and double instances are generated from nmf_oracle_impl.h.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__AVX2__) && defined(__FMA__) && !defined(ORACLE_NO_FAST_SGEMM)
#include "sgemm_avx2.h"
#define ORACLE_FAST_SGEMM 1
#endif

#define T float
#define SFX f32
#define EPS_T FLT_EPSILON
#include "nmf_oracle_impl.h"
#undef T
#undef SFX
#undef EPS_T
#undef ORACLE_FAST_SGEMM

#define T double
#define SFX f64
#define EPS_T DBL_EPSILON
#include "nmf_oracle_impl.h"
#undef T
#undef SFX
#undef EPS_T

/* Per-run seed stream.  Restates IAlgorithm::IAlgorithm / generateRandomNumber
 * (source/nmf/Algorithm.cpp:26-31): uniform_int_distribution<unsigned>(0, UINT_MAX) bound to
 * std::mt19937(seed); over the full 32-bit range libstdc++ returns the raw MT19937 outputs.
 * MT19937 itself is the published Matsumoto-Nishimura generator.  Writes the first `count`
 * draws; run k of a compute() call sees description.seed = out[k-1]. */
void oracle_seed_stream(uint32_t seed, int count, uint32_t* out) {
	uint32_t mt[624];
	mt[0] = seed;
	for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
	int idx = 624;
	for (int c = 0; c < count; ++c) {
		if (idx >= 624) {
			for (int i = 0; i < 624; ++i) {
				uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
				mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
			}
			idx = 0;
		}
		uint32_t y = mt[idx++];
		y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
		out[c] = y;
	}
}

/* Best-run bookkeeping.  Restates Summary::insert / reset (source/nmf/Summary.cpp:47-60):
 * a record becomes the best run only if its frobenius is STRICTLY below every earlier one.
 * Given the frobenius values in insertion order, returns the index bestRun() would report. */
unsigned oracle_summary_best_run(const double* frobenius, int count) {
	unsigned best = 0;
	for (int k = 0; k < count; ++k) {
		int better = 1;
		for (int i = 0; i < k; ++i) if (frobenius[i] <= frobenius[k]) { better = 0; break; }
		if (better) best = (unsigned)k;
	}
	return best;
}

/* Direct residual ||V - W H||_F in double, as example/main.cpp:133-146 computes it. */
double oracle_direct_frobenius_f64(int m, int n, int r, const double* V, int ldv, const double* W, int ldw, const double* H, int ldh) {
	double acc = 0.0;
#pragma omp parallel for reduction(+:acc) schedule(static)
	for (int j = 0; j < n; ++j)
		for (int i = 0; i < m; ++i) {
			double s = 0.0;
			for (int k = 0; k < r; ++k) s += W[(size_t)k * ldw + i] * H[(size_t)j * ldh + k];
			double d = V[(size_t)j * ldv + i] - s;
			acc += d * d;
		}
	return sqrt(acc);
}

double oracle_direct_frobenius_f32(int m, int n, int r, const float* V, int ldv, const float* W, int ldw, const float* H, int ldh) {
	double acc = 0.0;
#pragma omp parallel for reduction(+:acc) schedule(static)
	for (int j = 0; j < n; ++j)
		for (int i = 0; i < m; ++i) {
			double s = 0.0;
			for (int k = 0; k < r; ++k) s += (double)W[(size_t)k * ldw + i] * (double)H[(size_t)j * ldh + k];
			double d = (double)V[(size_t)j * ldv + i] - s;
			acc += d * d;
		}
	return sqrt(acc);
}

#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* The GPU box exposes far more hardware threads than its CPU share allows; the Python front end
 * sizes the team to the share (and to 1 for small problems, where fork/join dominates). */
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
	omp_set_num_threads(n > 0 ? n : 1);
#else
	(void)n;
#endif
}

/* Bit-exact restatement of the engine's fp32 MFMA factor product (nmfgpu_amd/csrc/kernels.hip,
 * k_factor_product_f32): OUT(c, x) = sum_y F(c, y) A(x, y), with the engine's summation order:
 *   - the reduction range is cut into `splits` workgroup slices of 8 wave pieces each, piece
 *     boundaries at floor(steps_total * i / (8 * splits)) K-steps of two y;
 *   - inside a piece the v_mfma_f32_32x32x2_f32 accumulator is a y-ordered fmaf chain
 *     (/opt/skills/guides/cdna_hip_programming.md, "FP32-input MFMA": D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)));
 *   - the 8 pieces of a slice are added in wave order, the slices in slice order.
 * A is X x Y (lda), F is r x Y (ldf), OUT is r x X (ldo), all column-major. */
void oracle_emulate_factor_product_f32(int X, int Y, int r, const float* A, int lda, const float* F, int ldf,
                                       int splits, float* OUT, int ldo) {
	const int steps_total = (Y + 1) / 2;
	const int nw = splits * 8;
#pragma omp parallel for schedule(static)
	for (int x = 0; x < X; ++x)
		for (int c = 0; c < r; ++c) {
			float total = 0.f;
			for (int sp = 0; sp < splits; ++sp) {
				float slab = 0.f;
				for (int w = 0; w < 8; ++w) {
					const int widx = sp * 8 + w;
					const int s0 = (int)(((long)steps_total * widx) / nw);
					const int s1 = (int)(((long)steps_total * (widx + 1)) / nw);
					float acc = 0.f;
					for (int st = s0; st < s1; ++st)
						for (int k = 0; k < 2; ++k) {
							const int y = 2 * st + k;
							const float a = y < Y ? A[(size_t)y * lda + x] : 0.f;
							const float f = y < Y ? F[(size_t)y * ldf + c] : 0.f;
							acc = fmaf(a, f, acc);
						}
					slab = w == 0 ? acc : slab + acc;
				}
				total = sp == 0 ? slab : total + slab;
			}
			OUT[(size_t)x * ldo + c] = total;
		}
}
