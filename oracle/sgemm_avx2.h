/*
 * sgemm_avx2.h -- register-blocked fp32 products for the CPU oracle (TEST INFRASTRUCTURE, float instantiation only).
 *
 * Why: bench.py's `cpu_baseline` times this port on the GPU box's host cores.  With compiler-vectorised loops the two
 * large products of an iteration (W^T V and V H^T, AlgorithmMultiplicativeFrobenius.h:176-178,240-241) ran at 15-20 %
 * of the cores' FMA rate and the figure swung with the machine; a packed 16 x 6 micro-kernel (12 ymm accumulators,
 * two panel loads and six broadcasts per twelve FMAs) puts the port where a BLAS puts the reference's host-side
 * equivalents.  Same arithmetic as nmf_oracle_impl.h: every product term is one fp32 FMA, accumulation in fp32; only the
 * order of the additions differs (reduction range in chunks, chunk sums added in chunk order) and it does not depend on
 * the thread count.
 *
 * Column-major everywhere, explicit leading dimensions, like the rest of the oracle.
 */
#ifndef ORACLE_SGEMM_AVX2_H
#define ORACLE_SGEMM_AVX2_H

#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

enum { SG_MR = 16, SG_NR = 6 };

/* T(16 x NR) = sum_k a[k*16 .. +16] * b[k*bsk + q*bsn]; `a` is a packed 16-row panel (zero padded), b is addressed by strides */
#define SG_STEP(q) if (nr_ > q) { const __m256 bq = _mm256_broadcast_ss(bk + (long)q * bsn); l##q = _mm256_fmadd_ps(a0, bq, l##q); h##q = _mm256_fmadd_ps(a1, bq, h##q); }
#define SG_OUT(q) if (nr_ > q) { lo[q] = l##q; hi[q] = h##q; }
/* accumulators are named scalars on purpose: as array elements gcc 11 stores every one of them back to the stack each step */
#define SG_UKR(NR)                                                                                                   \
	static inline __attribute__((always_inline)) void sg_ukr##NR(int kc, const float* a, const float* b, long bsk,    \
	                                                             long bsn, __m256* lo, __m256* hi) {                  \
		enum { nr_ = NR };                                                                                            \
		const __m256 z = _mm256_setzero_ps();                                                                         \
		__m256 l0 = z, l1 = z, l2 = z, l3 = z, l4 = z, l5 = z, h0 = z, h1 = z, h2 = z, h3 = z, h4 = z, h5 = z;        \
		for (int k = 0; k < kc; ++k) {                                                                                \
			const __m256 a0 = _mm256_loadu_ps(a + (long)k * 16), a1 = _mm256_loadu_ps(a + (long)k * 16 + 8);          \
			const float* bk = b + (long)k * bsk;                                                                      \
			SG_STEP(0) SG_STEP(1) SG_STEP(2) SG_STEP(3) SG_STEP(4) SG_STEP(5)                                         \
		}                                                                                                             \
		SG_OUT(0) SG_OUT(1) SG_OUT(2) SG_OUT(3) SG_OUT(4) SG_OUT(5)                                                   \
		(void)l1; (void)l2; (void)l3; (void)l4; (void)l5; (void)h1; (void)h2; (void)h3; (void)h4; (void)h5;           \
	}
SG_UKR(1) SG_UKR(2) SG_UKR(3) SG_UKR(4) SG_UKR(5) SG_UKR(6)
#undef SG_UKR
#undef SG_STEP
#undef SG_OUT

/* C(mr x nr tile, ldc) = (first ? 0 : C) + T */
static inline void sg_tile(int kc, const float* a, const float* b, long bsk, long bsn, float* c, long ldc, int mr, int nr, int first) {
	__m256 lo[SG_NR], hi[SG_NR];
	switch (nr) {
	case 6: sg_ukr6(kc, a, b, bsk, bsn, lo, hi); break;
	case 5: sg_ukr5(kc, a, b, bsk, bsn, lo, hi); break;
	case 4: sg_ukr4(kc, a, b, bsk, bsn, lo, hi); break;
	case 3: sg_ukr3(kc, a, b, bsk, bsn, lo, hi); break;
	case 2: sg_ukr2(kc, a, b, bsk, bsn, lo, hi); break;
	default: sg_ukr1(kc, a, b, bsk, bsn, lo, hi); break;
	}
	if (mr == SG_MR) {
		for (int q = 0; q < nr; ++q) {
			float* cq = c + (long)q * ldc;
			if (first) { _mm256_storeu_ps(cq, lo[q]); _mm256_storeu_ps(cq + 8, hi[q]); }
			else { _mm256_storeu_ps(cq, _mm256_add_ps(_mm256_loadu_ps(cq), lo[q])); _mm256_storeu_ps(cq + 8, _mm256_add_ps(_mm256_loadu_ps(cq + 8), hi[q])); }
		}
	} else {
		float t[SG_MR];
		for (int q = 0; q < nr; ++q) {
			float* cq = c + (long)q * ldc;
			_mm256_storeu_ps(t, lo[q]); _mm256_storeu_ps(t + 8, hi[q]);
			for (int i = 0; i < mr; ++i) cq[i] = first ? t[i] : cq[i] + t[i];
		}
	}
}

/* C(ka x kb) = A^T B, A is m x ka, B is m x kb.  A^T is packed once into 16-row panels P[panel][i][16]; a task owns a run of
 * 6-column blocks of B and walks the reduction range in chunks of KC rows: panel chunk (16 KB) in L1, the task's chunk of B in L2,
 * B itself streamed from memory exactly once. */
static int sgemm_tn_avx2(int m, int ka, int kb, const float* A, int lda, const float* B, int ldb, float* C, int ldc) {
	enum { KC = 256 };
	if (m <= 0) { for (int j = 0; j < kb; ++j) for (int a = 0; a < ka; ++a) C[(size_t)j * ldc + a] = 0; return 1; }
	const int panels = (ka + SG_MR - 1) / SG_MR;
	float* P = (float*)malloc(sizeof(float) * (size_t)panels * (size_t)m * SG_MR);
	if (!P) return 0;
#pragma omp parallel for schedule(static) collapse(2)
	for (int p = 0; p < panels; ++p)
		for (int i0 = 0; i0 < m; i0 += 1024) {
			const int i1 = i0 + 1024 < m ? i0 + 1024 : m;
			float* dst = P + ((size_t)p * m) * SG_MR;
			for (int c = 0; c < SG_MR; ++c) {
				const int col = p * SG_MR + c;
				if (col < ka) { const float* src = A + (size_t)col * lda; for (int i = i0; i < i1; ++i) dst[(size_t)i * SG_MR + c] = src[i]; }
				else for (int i = i0; i < i1; ++i) dst[(size_t)i * SG_MR + c] = 0.0f;
			}
		}
	const int nblk = (kb + SG_NR - 1) / SG_NR;
	int threads = 1;
#ifdef _OPENMP
	threads = omp_get_max_threads();
#endif
	int per = (nblk + 4 * threads - 1) / (4 * threads);
	if (per < 1) per = 1;
	if (per > 16) per = 16;
	const int tasks = (nblk + per - 1) / per;
#pragma omp parallel for schedule(dynamic, 1)
	for (int t = 0; t < tasks; ++t) {
		const int b0 = t * per, b1 = b0 + per < nblk ? b0 + per : nblk;
		for (int k0 = 0; k0 < m; k0 += KC) {
			const int kc = k0 + KC < m ? KC : m - k0;
			for (int p = 0; p < panels; ++p) {
				const float* a = P + ((size_t)p * m + k0) * SG_MR;
				const int mr = ka - p * SG_MR < SG_MR ? ka - p * SG_MR : SG_MR;
				for (int b = b0; b < b1; ++b) {
					const int j = b * SG_NR, nr = kb - j < SG_NR ? kb - j : SG_NR;
					sg_tile(kc, a, B + (size_t)j * ldb + k0, 1, ldb, C + (size_t)j * ldc + p * SG_MR, ldc, mr, nr, k0 == 0);
				}
			}
		}
	}
	free(P);
	return 1;
}

/* rows [i0, i1) of  sum_{j in [j0, j1)} A(:, j) B(:, j)^T  into D (ldd), D(0, 0) standing for row i0; Bp is B packed by sg_pack_b */
static void sg_nt_block(int i0, int i1, int j0, int j1, int kb, const float* A, int lda, const float* Bp, int n, float* D, long ldd, float* pack) {
	enum { KC = 128 };
	const int subs = (i1 - i0 + SG_MR - 1) / SG_MR, nblk = (kb + SG_NR - 1) / SG_NR;
	for (int c0 = j0; c0 < j1; c0 += KC) {       /* j0 is a multiple of KC: chunk boundaries are the packing's */
		const int kc = c0 + KC < j1 ? KC : j1 - c0;
		for (int j = 0; j < kc; ++j) {           /* one 512-byte run of a column of A at a time */
			const float* src = A + (size_t)(c0 + j) * lda + i0;
			for (int s = 0; s < subs; ++s) {
				float* dst = pack + ((size_t)s * KC + j) * SG_MR;
				const int mr = (i1 - i0) - s * SG_MR;
				if (mr >= SG_MR) { _mm256_storeu_ps(dst, _mm256_loadu_ps(src + s * SG_MR)); _mm256_storeu_ps(dst + 8, _mm256_loadu_ps(src + s * SG_MR + 8)); }
				else for (int i = 0; i < SG_MR; ++i) dst[i] = i < mr ? src[s * SG_MR + i] : 0.0f;
			}
		}
		const float* bchunk = Bp + (size_t)(c0 / KC) * nblk * KC * SG_NR;
		for (int b = 0; b < nblk; ++b) {
			const int k = b * SG_NR, nr = kb - k < SG_NR ? kb - k : SG_NR;
			const float* bb = bchunk + (size_t)b * KC * SG_NR;
			for (int s = 0; s < subs; ++s) {
				const int r0 = s * SG_MR, mr = (i1 - i0) - r0 < SG_MR ? (i1 - i0) - r0 : SG_MR;
				sg_tile(kc, pack + (size_t)s * KC * SG_MR, bb, SG_NR, 1, D + (size_t)k * ldd + r0, ldd, mr, nr, c0 == j0);
			}
		}
	}
	(void)n;
}

/* C(m x kb) = A B^T, A is m x n, B is kb x n.  B is packed once as [chunk of 128 columns][6-row block][column][6]; a task owns 128
 * rows of A, copies 128 x 128 blocks of it into 16-row panels (the only pass over A, in 512-byte runs) and multiplies out of L2.
 * A short, wide A (H H^T) is cut along the reduction range instead, in fixed 512-column pieces summed in piece order. */
static int sgemm_nt_avx2(int m, int n, int kb, const float* A, int lda, const float* B, int ldb, float* C, int ldc) {
	enum { KC = 128, RB = 128, PIECE = 512 };
	if (n <= 0) { for (int k = 0; k < kb; ++k) for (int i = 0; i < m; ++i) C[(size_t)k * ldc + i] = 0; return 1; }
	const int nblk = (kb + SG_NR - 1) / SG_NR, chunks = (n + KC - 1) / KC;
	float* Bp = (float*)malloc(sizeof(float) * (size_t)chunks * nblk * KC * SG_NR);
	if (!Bp) return 0;
#pragma omp parallel for schedule(static)
	for (int c = 0; c < chunks; ++c) {
		const int c0 = c * KC, kc = c0 + KC < n ? KC : n - c0;
		for (int b = 0; b < nblk; ++b) {
			float* dst = Bp + ((size_t)c * nblk + b) * KC * SG_NR;
			for (int j = 0; j < kc; ++j)
				for (int q = 0; q < SG_NR; ++q) dst[(size_t)j * SG_NR + q] = b * SG_NR + q < kb ? B[(size_t)(c0 + j) * ldb + b * SG_NR + q] : 0.0f;
		}
	}
	int ok = 1;
	if (m > 2 * RB || n <= PIECE) {
		const int tasks = (m + RB - 1) / RB;
#pragma omp parallel
		{
			float* pack = (float*)malloc(sizeof(float) * (size_t)(RB / SG_MR) * KC * SG_MR);
			if (!pack) {
#pragma omp atomic write
				ok = 0;
			}
#pragma omp for schedule(dynamic, 1)
			for (int t = 0; t < tasks; ++t) {
				const int i0 = t * RB, i1 = i0 + RB < m ? i0 + RB : m;
				if (pack) sg_nt_block(i0, i1, 0, n, kb, A, lda, Bp, n, C + i0, ldc, pack);
			}
			free(pack);
		}
	} else {
		const int pieces = (n + PIECE - 1) / PIECE, rblocks = (m + RB - 1) / RB;
		float* part = (float*)malloc(sizeof(float) * (size_t)pieces * m * kb);
		if (!part) { free(Bp); return 0; }
#pragma omp parallel
		{
			float* pack = (float*)malloc(sizeof(float) * (size_t)(RB / SG_MR) * KC * SG_MR);
			if (!pack) {
#pragma omp atomic write
				ok = 0;
			}
#pragma omp for schedule(dynamic, 1) collapse(2)
			for (int p = 0; p < pieces; ++p)
				for (int t = 0; t < rblocks; ++t) {
					const int i0 = t * RB, i1 = i0 + RB < m ? i0 + RB : m;
					const int j0 = p * PIECE, j1 = j0 + PIECE < n ? j0 + PIECE : n;
					if (pack) sg_nt_block(i0, i1, j0, j1, kb, A, lda, Bp, n, part + (size_t)p * m * kb + i0, m, pack);
				}
			free(pack);
		}
		if (ok)
			for (int k = 0; k < kb; ++k)
				for (int i = 0; i < m; ++i) {
					float s = part[(size_t)k * m + i];
					for (int p = 1; p < pieces; ++p) s += part[((size_t)p * kb + k) * m + i];
					C[(size_t)k * ldc + i] = s;
				}
		free(part);
	}
	free(Bp);
	return ok;
}

#endif
