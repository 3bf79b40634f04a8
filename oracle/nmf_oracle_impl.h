/*
 * nmf_oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE).
 *
 * Included twice by nmf_oracle.c, once with T=float / SFX=f32 and once with
 * T=double / SFX=f64.  See nmf_oracle.c for the status header ("parity
 * unpinned", who may call this, what it restates).
 *
 * All matrices are column-major with an explicit leading dimension, exactly
 * like the reference's DeviceMatrix (source/common/Matrix.h:446-452).
 * Every product accumulates in T, one rounding per multiply-add at most --
 * the reference's cuBLAS calls accumulate in the matrix type as well.
 */

#define CAT2(a, b) a##_##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, SFX)

/* ---- products -------------------------------------------------------------------------- */

/* C(ka x kb) = A^T B, A is m x ka, B is m x kb.  Restates the gemm-TN calls
 * `W.transposed() * W` and `W.transposed() * V`
 * (AlgorithmMultiplicativeFrobenius.h:168-169,176-178,187-188 via Matrix.h:361-376). */
static void FN(gemm_tn)(int m, int ka, int kb, const T* A, int lda, const T* B, int ldb, T* C, int ldc) {
#ifdef ORACLE_FAST_SGEMM
	if (sgemm_tn_avx2(m, ka, kb, A, lda, B, ldb, C, ldc)) return;      /* float only: packed 16 x 6 micro-kernel (sgemm_avx2.h) */
#endif
	/* Blocked for the cache hierarchy: a thread owns a pair of columns of B; the reduction range is walked in chunks of
	 * IC rows (the chunk of A -- ka columns x IC rows -- stays in L2 while all column blocks of A pass over it), and a
	 * 4 x 2 block of C is accumulated per pass with simd reductions (vector-wide partial sums; without the pragma a float
	 * reduction stays scalar).  Before this the product streamed all of A once per column of B: 26 GFLOP/s on 8 threads. */
	enum { IC = 2048, AB = 4 };
#pragma omp parallel for schedule(static)
	for (int j0 = 0; j0 < kb; j0 += 2) {
		const int jl = j0 + 2 <= kb ? 2 : 1;
		const T* b0 = B + (size_t)j0 * ldb;
		const T* b1 = B + (size_t)(j0 + jl - 1) * ldb;      /* = b0 when the last column stands alone */
		for (int a = 0; a < ka; ++a) { C[(size_t)j0 * ldc + a] = 0; if (jl == 2) C[(size_t)(j0 + 1) * ldc + a] = 0; }
		for (int i0 = 0; i0 < m; i0 += IC) {
			const int i1 = i0 + IC < m ? i0 + IC : m;
			int a = 0;
			for (; a + AB <= ka; a += AB) {
				const T* a0 = A + (size_t)(a + 0) * lda;
				const T* a1 = A + (size_t)(a + 1) * lda;
				const T* a2 = A + (size_t)(a + 2) * lda;
				const T* a3 = A + (size_t)(a + 3) * lda;
				T s00 = 0, s10 = 0, s20 = 0, s30 = 0, s01 = 0, s11 = 0, s21 = 0, s31 = 0;
#pragma omp simd reduction(+ : s00, s10, s20, s30, s01, s11, s21, s31)
				for (int i = i0; i < i1; ++i) {
					const T v0 = b0[i], v1 = b1[i];
					s00 += a0[i] * v0; s10 += a1[i] * v0; s20 += a2[i] * v0; s30 += a3[i] * v0;
					s01 += a0[i] * v1; s11 += a1[i] * v1; s21 += a2[i] * v1; s31 += a3[i] * v1;
				}
				T* c0 = C + (size_t)j0 * ldc + a;
				c0[0] += s00; c0[1] += s10; c0[2] += s20; c0[3] += s30;
				if (jl == 2) { T* c1 = C + (size_t)(j0 + 1) * ldc + a; c1[0] += s01; c1[1] += s11; c1[2] += s21; c1[3] += s31; }
			}
			for (; a < ka; ++a) {
				const T* a0 = A + (size_t)a * lda;
				T s0 = 0, s1 = 0;
#pragma omp simd reduction(+ : s0, s1)
				for (int i = i0; i < i1; ++i) { s0 += a0[i] * b0[i]; s1 += a0[i] * b1[i]; }
				C[(size_t)j0 * ldc + a] += s0;
				if (jl == 2) C[(size_t)(j0 + 1) * ldc + a] += s1;
			}
		}
	}
}

/* C(m x kb) = A B^T, A is m x n, B is kb x n.  Restates gemm-NT `V * H.transposed()`
 * and `H * H.transposed()` (AlgorithmMultiplicativeFrobenius.h:208-209,231-232,240-241). */
static void FN(gemm_nt)(int m, int n, int kb, const T* A, int lda, const T* B, int ldb, T* C, int ldc) {
#ifdef ORACLE_FAST_SGEMM
	if (sgemm_nt_avx2(m, n, kb, A, lda, B, ldb, C, ldc)) return;
#endif
	const int RB = 256; /* row block owned by one thread: C block stays in L2 */
#pragma omp parallel for schedule(static)
	for (int i0 = 0; i0 < m; i0 += RB) {
		int i1 = i0 + RB < m ? i0 + RB : m;
		for (int k = 0; k < kb; ++k)
			for (int i = i0; i < i1; ++i) C[(size_t)k * ldc + i] = 0;
		for (int j = 0; j < n; ++j) {
			const T* a = A + (size_t)j * lda;
			for (int k = 0; k < kb; ++k) {
				T h = B[(size_t)j * ldb + k];
				T* c = C + (size_t)k * ldc;
				for (int i = i0; i < i1; ++i) c[i] += a[i] * h;
			}
		}
	}
}

/* C(m x n) = A B, A is m x k, B is k x n.  Restates gemm-NN `RR * H`, `W * RR`, `W * S`, `S * H`. */
static void FN(gemm_nn)(int m, int n, int k, const T* A, int lda, const T* B, int ldb, T* C, int ldc) {
#pragma omp parallel for schedule(static)
	for (int j = 0; j < n; ++j) {
		T* c = C + (size_t)j * ldc;
		for (int i = 0; i < m; ++i) c[i] = 0;
		for (int p = 0; p < k; ++p) {
			T b = B[(size_t)j * ldb + p];
			const T* a = A + (size_t)p * lda;
			for (int i = 0; i < m; ++i) c[i] += a[i] * b;
		}
	}
}

/* ---- the reference's custom kernels ----------------------------------------------------- */

/* x = x * num / (den + eps); restates KernelMultiplyDivide.cu:39-42 (operation order kept:
 * the product first, then the division). */
static void FN(multiply_divide)(int rows, int cols, T* X, int ldx, const T* Num, int ldn, const T* Den, int ldd, T eps) {
#pragma omp parallel for schedule(static)
	for (int j = 0; j < cols; ++j)
		for (int i = 0; i < rows; ++i) {
			T value = X[(size_t)j * ldx + i];
			T upper = Num[(size_t)j * ldn + i];
			T lower = Den[(size_t)j * ldd + i];
			X[(size_t)j * ldx + i] = value * upper / (lower + eps);
		}
}

/* Column L2 normalisation with the `sum > 0` guard; restates KernelNormalizeColumns.cu:37-58. */
static void FN(normalize_columns)(int rows, int cols, T* A, int lda) {
#pragma omp parallel for schedule(static)
	for (int j = 0; j < cols; ++j) {
		T* a = A + (size_t)j * lda;
		T sum = 0;
		for (int i = 0; i < rows; ++i) sum += a[i] * a[i];
		if (sum > 0) {
			sum = (T)sqrt((double)sum);
			for (int i = 0; i < rows; ++i) a[i] = a[i] / sum;
		}
	}
}

/* ps[d] = sum_i A(i,d) B(i,d)       (transposeA: diagonal of A^T B)
 * ps[d] = sum_i A(d,i) B(i,d)       (otherwise:  diagonal of A B)
 * nDiag = columns of B, inner length = rows of B; restates KernelTraceMultiplication.cu:43-80
 * (the vector of per-diagonal sums is the output; it is NOT summed here). */
static void FN(trace_multiplication)(int transposeA, int nDiag, int inner, const T* A, int lda, const T* B, int ldb, T* ps) {
	for (int d = 0; d < nDiag; ++d) {
		T sum = 0;
		for (int i = 0; i < inner; ++i) {
			T a = transposeA ? A[(size_t)d * lda + i] : A[(size_t)i * lda + d];
			sum += a * B[(size_t)d * ldb + i];
		}
		ps[d] = sum;
	}
}

/* A(i,j) = [reuse] A(i,j) + (i == j ? diag : offdiag); restates KernelFillMatrix.cu:29-45. */
static void FN(fill_matrix)(int reuse, int rows, int cols, T* A, int lda, T offdiag, T diag) {
	for (int j = 0; j < cols; ++j)
		for (int i = 0; i < rows; ++i) {
			T old = reuse ? A[(size_t)j * lda + i] : (T)0;
			A[(size_t)j * lda + i] = old + (i == j ? diag : offdiag);
		}
}

/* A = max(A, 0); restates KernelMakeNonNegative.cu:30-46. */
static void FN(make_non_negative)(int rows, int cols, T* A, int lda) {
	for (int j = 0; j < cols; ++j)
		for (int i = 0; i < rows; ++i)
			if (A[(size_t)j * lda + i] < 0) A[(size_t)j * lda + i] = 0;
}

static int FN(cmp_asc)(const void* a, const void* b) {
	T x = *(const T*)a, y = *(const T*)b;
	return (x > y) - (x < y);
}

/* Restates resolveFrobenius, FrobeniusResolver.cpp:29-51: the two iteration-dependent vectors
 * are sorted in place, then the three vectors are accumulated interleaved in double; the factor
 * 2.f is applied in T.  No clamp: a negative radicand gives NaN, like the reference. */
double FN(oracle_resolve_frobenius)(const T* vtv_sorted, int n_vtv, T* htwtv, int n_htwtv, T* hhtwtw, int n_hhtwtw) {
	qsort(htwtv, n_htwtv, sizeof(T), FN(cmp_asc));
	qsort(hhtwtw, n_hhtwtw, sizeof(T), FN(cmp_asc));
	double acc = 0.0;
	int mx = n_vtv > n_htwtv ? n_vtv : n_htwtv;
	if (n_hhtwtw > mx) mx = n_hhtwtw;
	for (int j = 0; j < mx; ++j) {
		if (j < n_vtv) acc += vtv_sorted[j];
		if (j < n_htwtv) acc -= (T)(2.f * htwtv[j]);
		if (j < n_hhtwtw) acc += hhtwtw[j];
	}
	return sqrt(acc);
}

/* ps[j] = sum_i V(i,j)^2, sorted ascending: the tr(V^T V) vector every algorithm builds once in
 * allocateMemory (AlgorithmMultiplicativeFrobenius.h:119-126). */
void FN(oracle_vtv_sorted)(int m, int n, const T* V, int ldv, T* ps) {
	FN(trace_multiplication)(1, n, m, V, ldv, V, ldv, ps);
	qsort(ps, n, sizeof(T), FN(cmp_asc));
}

/* ---- sparse -> dense, honouring the index base (Matrix.h:145-232) ----------------------- */

void FN(oracle_densify_csr)(int rows, int cols, const T* values, const int* rowPtr, const int* colIdx, int base, T* out, int ld) {
	for (int j = 0; j < cols; ++j) for (int i = 0; i < rows; ++i) out[(size_t)j * ld + i] = 0;
	for (int i = 0; i < rows; ++i)
		for (int p = rowPtr[i] - base; p < rowPtr[i + 1] - base; ++p)
			out[(size_t)(colIdx[p] - base) * ld + i] = values[p];
}

void FN(oracle_densify_csc)(int rows, int cols, const T* values, const int* colPtr, const int* rowIdx, int base, T* out, int ld) {
	for (int j = 0; j < cols; ++j) for (int i = 0; i < rows; ++i) out[(size_t)j * ld + i] = 0;
	for (int j = 0; j < cols; ++j)
		for (int p = colPtr[j] - base; p < colPtr[j + 1] - base; ++p)
			out[(size_t)j * ld + (rowIdx[p] - base)] = values[p];
}

/* COO: the reference compresses the row indices with a hard-wired base of zero and then
 * densifies with the matrix's own base (Matrix.h:209,215-217), which is only self-consistent
 * for base zero.  The oracle applies `base` to both index arrays ("COO base"). */
void FN(oracle_densify_coo)(int rows, int cols, const T* values, const int* rowIdx, const int* colIdx, int nnz, int base, T* out, int ld) {
	for (int j = 0; j < cols; ++j) for (int i = 0; i < rows; ++i) out[(size_t)j * ld + i] = 0;
	for (int p = 0; p < nnz; ++p)
		out[(size_t)(colIdx[p] - base) * ld + (rowIdx[p] - base)] = values[p];
}

/* ---- Householder QR of the r x r normal matrix and the two solves ----------------------- */
/* cuSOLVER geqrf / ormqr and cuBLAS trsm are closed third-party code (CUDA >= 7.0, no version
 * pin beyond that: CMakeLists.txt:57-59); what is restated here is the published LAPACK
 * algorithm they implement (xGEQR2 / xORM2R / back substitution), applied at the reference's
 * call sites Matrix.h:565-618. */

/* In place: R in the upper triangle, Householder vectors below the diagonal, tau[r]. */
static void FN(qr_factor)(int r, T* A, int lda, T* tau) {
	for (int k = 0; k < r; ++k) {
		T* col = A + (size_t)k * lda;
		T alpha = col[k];
		T xnorm2 = 0;
		for (int i = k + 1; i < r; ++i) xnorm2 += col[i] * col[i];
		if (xnorm2 == 0) { tau[k] = 0; continue; }
		T beta = (T)sqrt((double)(alpha * alpha + xnorm2));
		if (alpha >= 0) beta = -beta;
		tau[k] = (beta - alpha) / beta;
		T scale = (T)1 / (alpha - beta);
		for (int i = k + 1; i < r; ++i) col[i] *= scale;
		col[k] = beta;
		/* apply H_k = I - tau v v^T to the trailing columns */
		for (int j = k + 1; j < r; ++j) {
			T* cj = A + (size_t)j * lda;
			T w = cj[k];
			for (int i = k + 1; i < r; ++i) w += col[i] * cj[i];
			w *= tau[k];
			cj[k] -= w;
			for (int i = k + 1; i < r; ++i) cj[i] -= w * col[i];
		}
	}
}

/* X(r x n) <- R^-1 Q^T X : ormqr(Left, transposed) then trsm(Left, Upper, NoTrans)
 * (AlgorithmAlternatingLeastSquares.h:163-169). */
static void FN(qr_solve_left)(int r, const T* QR, int ldq, const T* tau, int n, T* X, int ldx) {
#pragma omp parallel for schedule(static)
	for (int j = 0; j < n; ++j) {
		T* x = X + (size_t)j * ldx;
		for (int k = 0; k < r; ++k) {
			if (tau[k] == 0) continue;
			const T* v = QR + (size_t)k * ldq;
			T w = x[k];
			for (int i = k + 1; i < r; ++i) w += v[i] * x[i];
			w *= tau[k];
			x[k] -= w;
			for (int i = k + 1; i < r; ++i) x[i] -= w * v[i];
		}
		for (int k = r - 1; k >= 0; --k) {
			T s = x[k];
			for (int p = k + 1; p < r; ++p) s -= QR[(size_t)p * ldq + k] * x[p];
			x[k] = s / QR[(size_t)k * ldq + k];
		}
	}
}

/* X(m x r) <- X Q R^-T : ormqr(Right, not transposed) then trsm(Right, Upper, Trans)
 * (AlgorithmAlternatingLeastSquares.h:212-216). */
static void FN(qr_solve_right)(int r, const T* QR, int ldq, const T* tau, int m, T* X, int ldx) {
	T* row = (T*)malloc(sizeof(T) * (size_t)r);
	for (int i = 0; i < m; ++i) {
		for (int c = 0; c < r; ++c) row[c] = X[(size_t)c * ldx + i];
		/* x^T Q = x^T H_0 H_1 ... H_{r-1}: apply the reflectors in ascending order */
		for (int k = 0; k < r; ++k) {
			if (tau[k] == 0) continue;
			const T* v = QR + (size_t)k * ldq;
			T w = row[k];
			for (int c = k + 1; c < r; ++c) w += v[c] * row[c];
			w *= tau[k];
			row[k] -= w;
			for (int c = k + 1; c < r; ++c) row[c] -= w * v[c];
		}
		/* y R^T = x  <=>  R y^T = x^T : back substitution */
		for (int k = r - 1; k >= 0; --k) {
			T s = row[k];
			for (int p = k + 1; p < r; ++p) s -= QR[(size_t)p * ldq + k] * row[p];
			row[k] = s / QR[(size_t)k * ldq + k];
		}
		for (int c = 0; c < r; ++c) X[(size_t)c * ldx + i] = row[c];
	}
	free(row);
}

/* ---- workspace ------------------------------------------------------------------------- */

typedef struct {
	int m, n, r;
	T *RR, *RR2, *S, *RN, *RN2, *MR, *MR2, *tau;
	T *psN, *psR, *vtv; /* vtv: sorted tr(V^T V) vector */
} FN(ws_t);

static FN(ws_t)* FN(ws_new)(int m, int n, int r, const T* V, int ldv) {
	FN(ws_t)* w = (FN(ws_t)*)calloc(1, sizeof(FN(ws_t)));
	w->m = m; w->n = n; w->r = r;
	size_t big = (size_t)r * n > (size_t)m * r ? (size_t)r * n : (size_t)m * r;
	w->RR = (T*)calloc((size_t)r * r, sizeof(T));
	w->RR2 = (T*)calloc((size_t)r * r, sizeof(T));
	w->S = (T*)calloc((size_t)r * r, sizeof(T));
	w->RN = (T*)calloc(big, sizeof(T));
	w->RN2 = (T*)calloc(big, sizeof(T));
	w->MR = (T*)calloc((size_t)m * r, sizeof(T));
	w->MR2 = (T*)calloc((size_t)m * r, sizeof(T));
	w->tau = (T*)calloc((size_t)r, sizeof(T));
	w->psN = (T*)calloc((size_t)(n > r ? n : r), sizeof(T));
	w->psR = (T*)calloc((size_t)r, sizeof(T));
	w->vtv = (T*)calloc((size_t)n, sizeof(T));
	FN(oracle_vtv_sorted)(m, n, V, ldv, w->vtv);
	return w;
}

static void FN(ws_free)(FN(ws_t)* w) {
	free(w->RR); free(w->RR2); free(w->S); free(w->RN); free(w->RN2); free(w->MR); free(w->MR2);
	free(w->tau); free(w->psN); free(w->psR); free(w->vtv); free(w);
}

/* ---- one iteration of each algorithm ---------------------------------------------------- */

/* Lee-Seung multiplicative update, Frobenius objective.
 * Restates AlgorithmMultiplicativeFrobenius<T>::computeIteration, :150-248:
 *   H step :165-198  RR = W^T W; RN2 = RR H; RN = W^T V; H .*= RN ./ (RN2 + eps);
 *                    [error] psN[j] = sum_k H(k,j) RN(k,j)
 *   W step :201-248  [error] psR[d] = sum_i (H H^T)(d,i) (W^T W)(i,d)  -- W^T W from BEFORE this W update
 *                    RR = H H^T; MR2 = W RR; MR = V H^T; W .*= MR ./ (MR2 + eps); normalise columns
 *   error  :155-161  frob = resolve(vtv, psN, psR); rmsd = frob / sqrt(m n)
 * eps = machine epsilon of T (:191,244). */
static void FN(iter_mu)(FN(ws_t)* w, const T* V, int ldv, T* W, int ldw, T* H, int ldh,
                        int computeError, int constW, double* frob, double* rmsd) {
	int m = w->m, n = w->n, r = w->r;
	const T eps = EPS_T;
	FN(gemm_tn)(m, r, r, W, ldw, W, ldw, w->RR, r);
	FN(gemm_nn)(r, n, r, w->RR, r, H, ldh, w->RN2, r);
	FN(gemm_tn)(m, r, n, W, ldw, V, ldv, w->RN, r);
	FN(multiply_divide)(r, n, H, ldh, w->RN, r, w->RN2, r, eps);
	if (computeError) FN(trace_multiplication)(1, n, r, H, ldh, w->RN, r, w->psN);

	if (computeError) {
		memcpy(w->RR2, w->RR, sizeof(T) * (size_t)r * r);            /* tmpRR <- W^T W   :204-205 */
		FN(gemm_nt)(r, n, r, H, ldh, H, ldh, w->RR, r);               /* RR = H H^T        :208-209 */
		FN(trace_multiplication)(0, r, r, w->RR, r, w->RR2, r, w->psR); /*                  :212 */
		if (!constW) FN(gemm_nn)(m, r, r, W, ldw, w->RR, r, w->MR2, m);
	} else if (!constW) {
		FN(gemm_nt)(r, n, r, H, ldh, H, ldh, w->RR, r);
		FN(gemm_nn)(m, r, r, W, ldw, w->RR, r, w->MR2, m);
	}
	if (!constW) {
		FN(gemm_nt)(m, n, r, V, ldv, H, ldh, w->MR, m);
		FN(multiply_divide)(m, r, W, ldw, w->MR, m, w->MR2, m, eps);
		FN(normalize_columns)(m, r, W, ldw);
	}
	if (computeError) {
		*frob = FN(oracle_resolve_frobenius)(w->vtv, n, w->psN, n, w->psR, r);
		*rmsd = *frob / sqrt((double)((unsigned)m * (unsigned)n));
	}
}

/* non-smooth NMF.  Restates AlgorithmNonSmoothNMF<T>, :131-134 (S), :174-187 (H step),
 * :190-218 (W step), :160-171 (error).  S = (1-theta) I + (theta/r) 1 1^T. */
static void FN(iter_nsnmf)(FN(ws_t)* w, const T* V, int ldv, T* W, int ldw, T* H, int ldh,
                           int computeError, int constW, double* frob, double* rmsd) {
	int m = w->m, n = w->n, r = w->r;
	const T eps = EPS_T;
	/* H step: MR2 = W S; RR = MR2^T MR2; RN = MR2^T V; RN2 = RR H */
	FN(gemm_nn)(m, r, r, W, ldw, w->S, r, w->MR2, m);
	FN(gemm_tn)(m, r, r, w->MR2, m, w->MR2, m, w->RR, r);
	FN(gemm_tn)(m, r, n, w->MR2, m, V, ldv, w->RN, r);
	FN(gemm_nn)(r, n, r, w->RR, r, H, ldh, w->RN2, r);
	FN(multiply_divide)(r, n, H, ldh, w->RN, r, w->RN2, r, eps);
	if (computeError) FN(trace_multiplication)(1, n, r, H, ldh, w->RN, r, w->psN);

	if (!computeError && constW) return;
	/* W step: RN2 = S H; RR = RN2 RN2^T */
	FN(gemm_nn)(r, n, r, w->S, r, H, ldh, w->RN2, r);
	FN(gemm_nt)(r, n, r, w->RN2, r, w->RN2, r, w->RR, r);
	if (computeError) {
		FN(gemm_tn)(m, r, r, W, ldw, W, ldw, w->RR2, r);              /* viewRR2 = W^T W   :201-202 */
		FN(trace_multiplication)(0, r, r, w->RR, r, w->RR2, r, w->psR);
	}
	if (!constW) {
		FN(gemm_nt)(m, n, r, V, ldv, w->RN2, r, w->MR, m);
		FN(gemm_nn)(m, r, r, W, ldw, w->RR, r, w->MR2, m);
		FN(multiply_divide)(m, r, W, ldw, w->MR, m, w->MR2, m, eps);
		FN(normalize_columns)(m, r, W, ldw);
	}
	if (computeError) {
		*frob = FN(oracle_resolve_frobenius)(w->vtv, n, w->psN, n, w->psR, r);
		*rmsd = *frob / sqrt((double)((unsigned)m * (unsigned)n));
	}
}

/* Least-squares H shared by GDCLS / ALS / ACLS / AHCLS:
 *   RR = W^T W (+ saved copy RR2 on error iterations) + (offdiag, diag) regulariser; qr(RR);
 *   H = W^T V; H = R^-1 Q^T H; H = max(H, 0).
 * Restates AlgorithmGradientDescentConstrainedLeastSquares.h:175-210 and
 * AlgorithmAlternatingHoyerConstrainedLeastSquares.h:183-218. */
static void FN(ls_solve_h)(FN(ws_t)* w, const T* V, int ldv, const T* W, int ldw, T* H, int ldh,
                           int computeError, T offdiag, T diag) {
	int m = w->m, n = w->n, r = w->r;
	FN(gemm_tn)(m, r, r, W, ldw, W, ldw, w->RR, r);
	if (computeError) memcpy(w->RR2, w->RR, sizeof(T) * (size_t)r * r);
	FN(fill_matrix)(1, r, r, w->RR, r, offdiag, diag);
	FN(qr_factor)(r, w->RR, r, w->tau);
	FN(gemm_tn)(m, r, n, W, ldw, V, ldv, H, ldh);
	FN(qr_solve_left)(r, w->RR, r, w->tau, n, H, ldh);
	FN(make_non_negative)(r, n, H, ldh);
}

/* GDCLS.  Restates AlgorithmGradientDescentConstrainedLeastSquares<T>::computeIteration,
 * :159-171 with computeMatrixH :175-210 and computeMatrixW :213-265.  Note the quirk kept
 * from the reference: tr(H^T W^T V) is evaluated as diag(MR^T W) with the UPDATED W (:259-264)
 * while tr(H H^T W^T W) uses the W^T W saved before the update (:216-217). */
static void FN(iter_gdcls)(FN(ws_t)* w, const T* V, int ldv, T* W, int ldw, T* H, int ldh,
                           int computeError, int constW, T lambda, double* frob, double* rmsd) {
	int m = w->m, n = w->n, r = w->r;
	const T eps = EPS_T;
	FN(ls_solve_h)(w, V, ldv, W, ldw, H, ldh, computeError, (T)0, lambda);
	FN(gemm_nt)(r, n, r, H, ldh, H, ldh, w->RR, r);
	if (computeError) FN(trace_multiplication)(0, r, r, w->RR, r, w->RR2, r, w->psR);
	if (!constW) {
		FN(gemm_nn)(m, r, r, W, ldw, w->RR, r, w->MR2, m);
		FN(gemm_nt)(m, n, r, V, ldv, H, ldh, w->MR, m);
		FN(multiply_divide)(m, r, W, ldw, w->MR, m, w->MR2, m, eps);
		FN(normalize_columns)(m, r, W, ldw);
	}
	if (computeError) {
		/* with constant basis vectors the reference reads a stale MR here (:259-264); the oracle
		 * computes the product the formula names instead. */
		if (constW) FN(gemm_nt)(m, n, r, V, ldv, H, ldh, w->MR, m);
		FN(trace_multiplication)(1, r, m, w->MR, m, W, ldw, w->psN);
		*frob = FN(oracle_resolve_frobenius)(w->vtv, n, w->psN, r, w->psR, r);
		*rmsd = *frob / sqrt((double)((unsigned)m * (unsigned)n));
	}
}

/* ALS / ACLS / AHCLS.  Restates AlgorithmHoyerConstrainedAlternatingLeastSquares<T>::computeIteration,
 * :171-296 (ACLS: offdiag 0, diag lambda; AHCLS: offdiag -lambda, diag lambda*beta - lambda with
 * beta = ((1-alpha) sqrt(r) + alpha)^2, :81-84) and AlgorithmAlternatingLeastSquares.h:146-224
 * (lambda = 0).  Error terms: tr(W_old^T (V H^T)) over r diagonals (:252-270). */
static void FN(iter_als_family)(FN(ws_t)* w, const T* V, int ldv, T* W, int ldw, T* H, int ldh,
                                int computeError, int constW,
                                T offH, T diagH, T offW, T diagW, double* frob, double* rmsd) {
	int m = w->m, n = w->n, r = w->r;
	FN(ls_solve_h)(w, V, ldv, W, ldw, H, ldh, computeError, offH, diagH);
	FN(gemm_nt)(r, n, r, H, ldh, H, ldh, w->RR, r);
	if (computeError) FN(trace_multiplication)(0, r, r, w->RR, r, w->RR2, r, w->psR);
	if (!constW) {
		FN(fill_matrix)(1, r, r, w->RR, r, offW, diagW);
		FN(qr_factor)(r, w->RR, r, w->tau);
	}
	if (computeError) /* save W_old :247-249 */
		for (int c = 0; c < r; ++c) memcpy(w->MR + (size_t)c * m, W + (size_t)c * ldw, sizeof(T) * (size_t)m);
	if (!constW) FN(gemm_nt)(m, n, r, V, ldv, H, ldh, W, ldw);
	if (computeError) FN(trace_multiplication)(1, r, m, w->MR, m, W, ldw, w->psN);
	if (!constW) {
		FN(qr_solve_right)(r, w->RR, r, w->tau, m, W, ldw);
		FN(make_non_negative)(m, r, W, ldw);
		FN(normalize_columns)(m, r, W, ldw);
	}
	if (computeError) {
		*frob = FN(oracle_resolve_frobenius)(w->vtv, n, w->psN, r, w->psR, r);
		*rmsd = *frob / sqrt((double)((unsigned)m * (unsigned)n));
	}
}

/* ---- the run loop ----------------------------------------------------------------------- */

/* One run of SingleGpuDispatcher::dispatch's inner loop (SingleGpuDispatcher.cpp:165-205) from
 * the given W, H (i.e. initMethod CopyExisting): error evaluated when iteration % 10 == 0 or on
 * the last iteration; stop when lastError != 0 and |error - lastError| < threshold on the
 * selected measure; the interrupt hook is not modelled.
 *
 * algorithm: 0 MU, 1 GDCLS, 2 ALS, 3 ACLS, 4 AHCLS, 5 nsNMF (NmfAlgorithm order, nmfgpu.h:107-114)
 * params: [lambda, lambdaW, lambdaH, alphaW, alphaH, theta]
 * history (optional, 2 doubles per error evaluation: frobenius, rmsd), history_cap entries.
 * Returns the number of iterations executed (after the min() of :205). */
int FN(oracle_run)(int algorithm, int m, int n, int r, const T* V, int ldv, T* W, int ldw, T* H, int ldh,
                   int numIterations, int thresholdType, double thresholdValue, int constW,
                   const double* params, double* out_frob, double* out_rmsd,
                   double* history, int history_cap, int* history_len) {
	FN(ws_t)* w = FN(ws_new)(m, n, r, V, ldv);
	double frob = 0.0, rmsd = 0.0, lastError = 0.0;
	int hl = 0;
	T lambda = (T)params[0], lambdaW = (T)params[1], lambdaH = (T)params[2];
	T alphaW = (T)params[3], alphaH = (T)params[4];
	T offH = 0, diagH = 0, offW = 0, diagW = 0;
	if (algorithm == 3) { diagH = lambdaH; diagW = lambdaW; }
	if (algorithm == 4) {
		/* :81-84 -- sqrt of the unsigned feature count is evaluated in double, the product in T */
		T betaW = (T)((1 - alphaW) * sqrt((double)(unsigned)r) + alphaW); betaW *= betaW;
		T betaH = (T)((1 - alphaH) * sqrt((double)(unsigned)r) + alphaH); betaH *= betaH;
		offH = -lambdaH; diagH = lambdaH * betaH - lambdaH;
		offW = -lambdaW; diagW = lambdaW * betaW - lambdaW;
	}
	if (algorithm == 5) {
		/* :131-133 */
		T off = (T)params[5] / (T)(unsigned)r;
		T diag = (T)((1.0 - (T)params[5]) + off);
		FN(fill_matrix)(0, r, r, w->S, r, off, diag);
	}
	int iteration = 1;
	for (; iteration <= numIterations; ++iteration) {
		int computeError = (iteration % 10 == 0) || iteration == numIterations;
		switch (algorithm) {
		case 0: FN(iter_mu)(w, V, ldv, W, ldw, H, ldh, computeError, constW, &frob, &rmsd); break;
		case 1: FN(iter_gdcls)(w, V, ldv, W, ldw, H, ldh, computeError, constW, lambda, &frob, &rmsd); break;
		case 2: case 3: case 4:
			FN(iter_als_family)(w, V, ldv, W, ldw, H, ldh, computeError, constW, offH, diagH, offW, diagW, &frob, &rmsd); break;
		case 5: FN(iter_nsnmf)(w, V, ldv, W, ldw, H, ldh, computeError, constW, &frob, &rmsd); break;
		default: FN(ws_free)(w); return -1;
		}
		if (computeError) {
			if (history && hl < history_cap) { history[2 * hl] = frob; history[2 * hl + 1] = rmsd; }
			++hl;
			double cur = thresholdType == 0 ? frob : rmsd;
			double delta = cur - lastError;
			if (lastError != 0.0 && fabs(delta) < thresholdValue) break;
			lastError = cur;
		}
	}
	if (iteration > numIterations) iteration = numIterations;
	if (algorithm == 5) {
		/* storeFactorization returns W S, not W (AlgorithmNonSmoothNMF.h:221-225) */
		FN(gemm_nn)(m, r, r, W, ldw, w->S, r, w->MR, m);
		for (int c = 0; c < r; ++c) memcpy(W + (size_t)c * ldw, w->MR + (size_t)c * m, sizeof(T) * (size_t)m);
	}
	*out_frob = frob; *out_rmsd = rmsd;
	if (history_len) *history_len = hl;
	FN(ws_free)(w);
	return iteration;
}

/* ---- KL-divergence multiplicative update (EXTENSION: no reference counterpart) ----------- */
/* The reference implements only the Frobenius form (NmfAlgorithm has no KL member, nmfgpu.h:107-114).
 * This restates the update of Lee & Seung, "Algorithms for Non-negative Matrix Factorization" (NIPS 2001,
 * cited by the reference's README), in the skeleton of the reference's MU iteration
 * (AlgorithmMultiplicativeFrobenius.h:150-248): H step, W step with the new H, column normalisation of W,
 * eps = machine epsilon added to every denominator, error terms referring to (W_{k-1}, H_k):
 *     Q = V ./ (W H + eps);   H .*= (W^T Q) ./ (colsum(W) + eps)
 *     Q = V ./ (W H + eps);   W .*= (Q H^T) ./ (rowsum(H) + eps);   normalise columns of W
 * Zero entries of V contribute nothing (Q = 0), so a sparse evaluation over the stored entries is the
 * same arithmetic.  Reported: the Frobenius error by the reference's trace formula (per-ROW terms of
 * tr(H^T W^T V) here) and the generalised KL divergence
 *     D = sum_{v>0} v log(v / (wh + eps)) - sum v + sum_c colsum(W)_c rowsum(H)_c.
 * PARITY UNPINNED by the reference. */
int FN(oracle_kl_run)(int m, int n, int r, const T* V, int ldv, T* W, int ldw, T* H, int ldh, int numIterations,
                      double* out_frob, double* out_rmsd, double* out_kl) {
	const T eps = EPS_T;
	T* WH = (T*)malloc(sizeof(T) * (size_t)m * n);
	T* Q = (T*)malloc(sizeof(T) * (size_t)m * n);
	T* NH = (T*)malloc(sizeof(T) * (size_t)r * n);
	T* NW = (T*)malloc(sizeof(T) * (size_t)m * r);
	T* sW = (T*)malloc(sizeof(T) * (size_t)r);
	T* sH = (T*)malloc(sizeof(T) * (size_t)r);
	T* G = (T*)malloc(sizeof(T) * (size_t)r * r);
	T* HHt = (T*)malloc(sizeof(T) * (size_t)r * r);
	T* psRow = (T*)malloc(sizeof(T) * (size_t)m);
	T* psR = (T*)malloc(sizeof(T) * (size_t)r);
	T* vtv = (T*)malloc(sizeof(T) * (size_t)n);
	FN(oracle_vtv_sorted)(m, n, V, ldv, vtv);
	double frob = 0, rmsd = 0, kl = 0;
	for (int it = 1; it <= numIterations; ++it) {
		const int computeError = (it % 10 == 0) || it == numIterations;
		/* H step */
		FN(gemm_nn)(m, n, r, W, ldw, H, ldh, WH, m);
		for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) Q[(size_t)j * m + i] = V[(size_t)j * ldv + i] / (WH[(size_t)j * m + i] + eps);
		FN(gemm_tn)(m, r, n, W, ldw, Q, m, NH, r);
		for (int c = 0; c < r; ++c) { T s = 0; for (int i = 0; i < m; ++i) s += W[(size_t)c * ldw + i]; sW[c] = s; }
		for (int j = 0; j < n; ++j) for (int c = 0; c < r; ++c) H[(size_t)j * ldh + c] = H[(size_t)j * ldh + c] * NH[(size_t)j * r + c] / (sW[c] + eps);
		/* W step */
		FN(gemm_nn)(m, n, r, W, ldw, H, ldh, WH, m);
		for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) Q[(size_t)j * m + i] = V[(size_t)j * ldv + i] / (WH[(size_t)j * m + i] + eps);
		for (int c = 0; c < r; ++c) { T s = 0; for (int j = 0; j < n; ++j) s += H[(size_t)j * ldh + c]; sH[c] = s; }
		if (computeError) {
			double sumv = 0, d = 0;
			for (int i = 0; i < m; ++i) {
				T t = 0;
				for (int j = 0; j < n; ++j) {
					const T v = V[(size_t)j * ldv + i], wh = WH[(size_t)j * m + i];
					t += v * wh;
					sumv += (double)v;
					if (v > 0) d += (double)v * log((double)v / (double)(wh + eps));
				}
				psRow[i] = t;
			}
			FN(gemm_tn)(m, r, r, W, ldw, W, ldw, G, r);
			FN(gemm_nt)(r, n, r, H, ldh, H, ldh, HHt, r);
			FN(trace_multiplication)(0, r, r, HHt, r, G, r, psR);
			frob = FN(oracle_resolve_frobenius)(vtv, n, psRow, m, psR, r);
			rmsd = frob / sqrt((double)((unsigned)m * (unsigned)n));
			for (int c = 0; c < r; ++c) d += (double)sW[c] * (double)sH[c];
			kl = d - sumv;
		}
		FN(gemm_nt)(m, n, r, Q, m, H, ldh, NW, m);
		for (int c = 0; c < r; ++c) for (int i = 0; i < m; ++i) W[(size_t)c * ldw + i] = W[(size_t)c * ldw + i] * NW[(size_t)c * m + i] / (sH[c] + eps);
		FN(normalize_columns)(m, r, W, ldw);
	}
	*out_frob = frob; *out_rmsd = rmsd; *out_kl = kl;
	free(WH); free(Q); free(NH); free(NW); free(sW); free(sH); free(G); free(HHt); free(psRow); free(psR); free(vtv);
	return numIterations;
}

/* The same KL iteration evaluated over the STORED entries of a CSR matrix (0-based): the form BASELINE config 3
 * (100 000 x 20 000 at 1 %) needs -- the dense form above would hold 2 x 10^9 elements.  Zero entries contribute
 * nothing to any sum (Q = 0), so this is the arithmetic of oracle_kl_run with the products written per entry:
 *   WH(i, j) = sum_c W(i, c) H(c, j) in ascending c;  (W^T Q)(:, j) over the rows of column j in ascending i;
 *   (Q H^T)(i, :) over the columns of row i in ascending j.  W is m x r, H r x n, both column-major.
 * No reference counterpart (the reference densifies sparse input, Matrix.h:145-232): PARITY UNPINNED. */
int FN(oracle_kl_run_csr)(int m, int n, int r, const int* ptr, const int* idx, const T* val, T* W, int ldw, T* H, int ldh,
                          int numIterations, double* out_frob, double* out_rmsd, double* out_kl) {
	const T eps = EPS_T;
	if (r > 1024) return -1;   /* per-thread accumulators live on the stack */
	const long nnz = ptr[m];
	int* cptr = (int*)calloc((size_t)n + 1, sizeof(int));
	int* crow = (int*)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
	long* cmap = (long*)malloc(sizeof(long) * (size_t)(nnz > 0 ? nnz : 1));
	for (long p = 0; p < nnz; ++p) ++cptr[idx[p] + 1];
	for (int j = 0; j < n; ++j) cptr[j + 1] += cptr[j];
	{
		int* fill = (int*)malloc(sizeof(int) * (size_t)n);
		memcpy(fill, cptr, sizeof(int) * (size_t)n);
		for (int i = 0; i < m; ++i) for (int p = ptr[i]; p < ptr[i + 1]; ++p) { const int d = fill[idx[p]]++; crow[d] = i; cmap[d] = p; }
		free(fill);
	}
	T* Wr = (T*)malloc(sizeof(T) * (size_t)m * r);      /* row-major copy of W: W(i, :) contiguous */
	T* Q = (T*)malloc(sizeof(T) * (size_t)(nnz > 0 ? nnz : 1));
	T* sW = (T*)malloc(sizeof(T) * (size_t)r);
	T* sH = (T*)malloc(sizeof(T) * (size_t)r);
	T* G = (T*)malloc(sizeof(T) * (size_t)r * r);
	T* HHt = (T*)malloc(sizeof(T) * (size_t)r * r);
	T* psRow = (T*)malloc(sizeof(T) * (size_t)m);
	double* dRow = (double*)malloc(sizeof(double) * (size_t)m);
	T* psR = (T*)malloc(sizeof(T) * (size_t)r);
	T* vtv = (T*)malloc(sizeof(T) * (size_t)n);
	double sumv = 0;
	for (int j = 0; j < n; ++j) { T s = 0; for (int p = cptr[j]; p < cptr[j + 1]; ++p) { const T v = val[cmap[p]]; s += v * v; sumv += (double)v; } vtv[j] = s; }
	qsort(vtv, n, sizeof(T), FN(cmp_asc));
	double frob = 0, rmsd = 0, kl = 0;
	for (int it = 1; it <= numIterations; ++it) {
		const int computeError = (it % 10 == 0) || it == numIterations;
		for (int half = 0; half < 2; ++half) {
			if (half == 0) {
#pragma omp parallel for schedule(static)
				for (int i = 0; i < m; ++i) for (int c = 0; c < r; ++c) Wr[(size_t)i * r + c] = W[(size_t)c * ldw + i];
			}
			/* quotients on the stored entries (and, second evaluation of an error iteration, the per-row error terms) */
#pragma omp parallel for schedule(dynamic, 64)
			for (int i = 0; i < m; ++i) {
				const T* wi = Wr + (size_t)i * r;
				T t = 0; double d = 0;
				for (int p = ptr[i]; p < ptr[i + 1]; ++p) {
					const T* hj = H + (size_t)idx[p] * ldh;
					T wh = 0;
					for (int c = 0; c < r; ++c) wh += wi[c] * hj[c];
					const T v = val[p];
					Q[p] = v / (wh + eps);
					if (half == 1 && computeError) { t += v * wh; if (v > 0) d += (double)v * log((double)v / (double)(wh + eps)); }
				}
				if (half == 1 && computeError) { psRow[i] = t; dRow[i] = d; }
			}
			if (half == 0) {
				for (int c = 0; c < r; ++c) { T s = 0; for (int i = 0; i < m; ++i) s += W[(size_t)c * ldw + i]; sW[c] = s; }
#pragma omp parallel for schedule(dynamic, 64)
				for (int j = 0; j < n; ++j) {
					T* hj = H + (size_t)j * ldh;
					T acc[1024];
					for (int c = 0; c < r; ++c) acc[c] = 0;
					for (int p = cptr[j]; p < cptr[j + 1]; ++p) {
						const T q = Q[cmap[p]];
						const T* wi = Wr + (size_t)crow[p] * r;
						for (int c = 0; c < r; ++c) acc[c] += q * wi[c];
					}
					for (int c = 0; c < r; ++c) hj[c] = hj[c] * acc[c] / (sW[c] + eps);
				}
			}
		}
		for (int c = 0; c < r; ++c) { T s = 0; for (int j = 0; j < n; ++j) s += H[(size_t)j * ldh + c]; sH[c] = s; }
		if (computeError) {
			double d = 0;
			for (int i = 0; i < m; ++i) d += dRow[i];
			FN(gemm_tn)(m, r, r, W, ldw, W, ldw, G, r);
			FN(gemm_nt)(r, n, r, H, ldh, H, ldh, HHt, r);
			FN(trace_multiplication)(0, r, r, HHt, r, G, r, psR);
			frob = FN(oracle_resolve_frobenius)(vtv, n, psRow, m, psR, r);
			rmsd = frob / sqrt((double)((unsigned)m * (unsigned)n));
			for (int c = 0; c < r; ++c) d += (double)sW[c] * (double)sH[c];
			kl = d - sumv;
		}
#pragma omp parallel for schedule(dynamic, 64)
		for (int i = 0; i < m; ++i) {
			T acc[1024];
			for (int c = 0; c < r; ++c) acc[c] = 0;
			for (int p = ptr[i]; p < ptr[i + 1]; ++p) {
				const T q = Q[p];
				const T* hj = H + (size_t)idx[p] * ldh;
				for (int c = 0; c < r; ++c) acc[c] += q * hj[c];
			}
			for (int c = 0; c < r; ++c) W[(size_t)c * ldw + i] = W[(size_t)c * ldw + i] * acc[c] / (sH[c] + eps);
		}
		FN(normalize_columns)(m, r, W, ldw);
	}
	*out_frob = frob; *out_rmsd = rmsd; *out_kl = kl;
	free(cptr); free(crow); free(cmap); free(Wr); free(Q); free(sW); free(sH); free(G); free(HHt); free(psRow); free(dRow); free(psR); free(vtv);
	return numIterations;
}

/* Exposed single products / kernels so the parity tests can check each HIP kernel on its own. */
void FN(oracle_gemm_tn)(int m, int ka, int kb, const T* A, int lda, const T* B, int ldb, T* C, int ldc) { FN(gemm_tn)(m, ka, kb, A, lda, B, ldb, C, ldc); }
void FN(oracle_gemm_nt)(int m, int n, int kb, const T* A, int lda, const T* B, int ldb, T* C, int ldc) { FN(gemm_nt)(m, n, kb, A, lda, B, ldb, C, ldc); }
void FN(oracle_gemm_nn)(int m, int n, int k, const T* A, int lda, const T* B, int ldb, T* C, int ldc) { FN(gemm_nn)(m, n, k, A, lda, B, ldb, C, ldc); }
void FN(oracle_multiply_divide)(int rows, int cols, T* X, int ldx, const T* Num, int ldn, const T* Den, int ldd) { FN(multiply_divide)(rows, cols, X, ldx, Num, ldn, Den, ldd, EPS_T); }
void FN(oracle_normalize_columns)(int rows, int cols, T* A, int lda) { FN(normalize_columns)(rows, cols, A, lda); }
void FN(oracle_trace_multiplication)(int transposeA, int nDiag, int inner, const T* A, int lda, const T* B, int ldb, T* ps) { FN(trace_multiplication)(transposeA, nDiag, inner, A, lda, B, ldb, ps); }
/* X <- (A + regulariser)^-1 X via Householder QR, as the LS algorithms do for H (left) */
void FN(oracle_qr_solve_left)(int r, T* A, int lda, int n, T* X, int ldx) {
	T* tau = (T*)calloc((size_t)r, sizeof(T));
	FN(qr_factor)(r, A, lda, tau);
	FN(qr_solve_left)(r, A, lda, tau, n, X, ldx);
	free(tau);
}
/* X <- X Q R^-T, as the LS algorithms do for W (right) */
void FN(oracle_qr_solve_right)(int r, T* A, int lda, int m, T* X, int ldx) {
	T* tau = (T*)calloc((size_t)r, sizeof(T));
	FN(qr_factor)(r, A, lda, tau);
	FN(qr_solve_right)(r, A, lda, tau, m, X, ldx);
	free(tau);
}

#undef CAT2
#undef CAT
#undef FN
