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

#define T float
#define SFX f32
#define EPS_T FLT_EPSILON
#include "nmf_oracle_impl.h"
#undef T
#undef SFX
#undef EPS_T

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

int oracle_num_threads(void) {
#ifdef _OPENMP
	extern int omp_get_max_threads(void);
	return omp_get_max_threads();
#else
	return 1;
#endif
}
