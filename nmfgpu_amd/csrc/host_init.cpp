// host_init.cpp -- the initialisation strategies that run once per run, outside the iteration
// loop, kept on the host (BASELINE.json north star: "the k-means/NNDSVD init stays host-side C++").
//
// Restates on the CPU, over a small std::thread pool (no OpenMP runtime is pulled into the host
// process), in the reference's own summation orders:
//   k-means (Lloyd, Forgy start)        source/kmeans/kMeans.cu:126-278
//   KMeans* strategies                  source/init/KMeansStrategy.cpp:31-65
//   EIn-NMF membership transform        source/init/EInNMF.cu:44-91
//   MeanColumns                         source/init/MeanColumnStrategy.cpp:43-56, KernelMeanColumn.cu:30-50
// and one strategy the reference does not have (BASELINE north star names it; include/nmfgpu.h:80-100 has no such member, so it is selected by a Parameter):
//   NNDSVD / NNDSVDa / NNDSVDar         Boutsidis & Gallopoulos, "SVD based initialization: a head start for NMF", Pattern Recognition 41 (2008);
//                                       Parameter{"nndsvd", 0 | 1 | 2} overrides initMethod; truncated SVD by block subspace iteration, in double
// Summation orders kept from the reference so that memberships and centres agree bit for bit:
//   * squared distances: 32 strided lane partials, fused multiply-add, then the xor-butterfly of
//     sumWarpReduction (kMeans.cu:41-50, KernelHelper.cuh:31-43);
//   * centre = (sum of members in ascending column order) / count (kMeans.cu:100-117, members
//     sorted by (cluster, column) at :185-189);
//   * EIn-NMF: Hillis-Steele scan over 32 lanes in double, carry rounded to T (EInNMF.cu:59-90).
// Deliberate differences from the reference (each a defect there, see DESIGN.md "quirks"):
//   * memberships start at "none" instead of uninitialised device memory (kMeans.cu:53-78);
//   * every row of a centroid is updated (the reference halves its grid and skips the last
//     32-row block when ceil(rows/32) is odd, kMeans.cu:222-225);
//   * MeanColumns draws all 5*r indices (the reference leaves the last one uninitialised,
//     MeanColumnStrategy.cpp:48);
//   * KMeansAndAbsoluteWTV computes |W^T V| as nmfgpu.h documents (unhandled upstream,
//     KMeansStrategy.cpp:32-39).
#include "host_init.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <numeric>
#include <random>
#include <thread>
#include <vector>

#include <sched.h>

namespace nmfgpu {
namespace hostinit {

double uniform01(unsigned long long seed, unsigned long long index, bool single_precision) {
	uint64_t z = seed * 0x9E3779B97F4A7C15ull + index + 0x632BE59BD9B4E019ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	z = z ^ (z >> 31);
	if (single_precision) return (double)(float)(((z >> 40) + 1) * (1.0f / 16777216.0f));
	return ((z >> 11) + 1) * (1.0 / 9007199254740992.0);
}

namespace {

// Dense host copy of a MatrixDescription (column-major, ld = rows), honouring the index base.
template <typename T>
std::vector<T> to_dense(const MatrixDescription<T>& M) {
	const size_t rows = M.rows, cols = M.columns;
	std::vector<T> out(rows * cols, T(0));
	switch (M.format) {
	case StorageFormat::Dense:
		for (size_t j = 0; j < cols; ++j)
			std::memcpy(&out[j * rows], M.dense.values + j * (size_t)M.dense.leadingDimension, sizeof(T) * rows);
		break;
	case StorageFormat::CSR: {
		const int base = M.csr.base == IndexBase::One ? 1 : 0;
		for (size_t i = 0; i < rows; ++i)
			for (int p = M.csr.rowPtr[i] - base; p < M.csr.rowPtr[i + 1] - base; ++p)
				out[(size_t)(M.csr.columnIndices[p] - base) * rows + i] = M.csr.values[p];
		break;
	}
	case StorageFormat::CSC: {
		const int base = M.csc.base == IndexBase::One ? 1 : 0;
		for (size_t j = 0; j < cols; ++j)
			for (int p = M.csc.columnPtr[j] - base; p < M.csc.columnPtr[j + 1] - base; ++p)
				out[j * rows + (size_t)(M.csc.rowIndices[p] - base)] = M.csc.values[p];
		break;
	}
	case StorageFormat::COO: {
		const int base = M.coo.base == IndexBase::One ? 1 : 0;
		for (unsigned p = 0; p < M.coo.nnz; ++p)
			out[(size_t)(M.coo.columnIndices[p] - base) * rows + (size_t)(M.coo.rowIndices[p] - base)] = M.coo.values[p];
		break;
	}
	}
	return out;
}

// ---- threads -------------------------------------------------------------------------------
unsigned host_threads() {
	static const unsigned n = [] {
		if (const char* e = std::getenv("NMFAMD_HOST_THREADS")) { int v = std::atoi(e); if (v > 0) return (unsigned)std::min(v, 256); }
		cpu_set_t set;
		unsigned avail = sched_getaffinity(0, sizeof(set), &set) == 0 ? (unsigned)CPU_COUNT(&set) : std::thread::hardware_concurrency();
		return std::max(1u, std::min(avail, 16u));   // 16 = a GPU box's CPU share per GPU
	}();
	return n;
}

// f(begin, end, slot) over [0, n) cut into contiguous chunks that are multiples of `align`.
template <typename F>
void parallel_chunks(size_t n, size_t align, size_t min_chunk, F&& f) {
	size_t units = (n + align - 1) / align;
	size_t want = std::max<size_t>(1, std::min<size_t>(host_threads(), n / std::max<size_t>(min_chunk, 1)));
	size_t per = (units + want - 1) / want;
	size_t chunks = per ? (units + per - 1) / per : 0;
	if (chunks <= 1) { if (n) f(size_t(0), n, size_t(0)); return; }
	std::vector<std::thread> pool;
	pool.reserve(chunks - 1);
	for (size_t c = 1; c < chunks; ++c)
		pool.emplace_back([&, c] { f(c * per * align, std::min(n, (c + 1) * per * align), c); });
	f(size_t(0), std::min(n, per * align), size_t(0));
	for (auto& t : pool) t.join();
}

// ---- the reference's warp-shaped reductions ---------------------------------------------------
// x86-64-v3 clones give the compiler vfmadd; the baseline clone calls libm's fma -- same bits.
// The *_impl templates are force-inlined so that each clone compiles them for its own target.
#ifndef NMFAMD_CLONES      // (a sanitizer build defines it empty: gcc cannot combine function multiversioning with -fsanitize)
#define NMFAMD_CLONES __attribute__((target_clones("default", "arch=x86-64-v3")))
#endif
#define NMFAMD_INLINE inline __attribute__((always_inline))

template <typename T>
NMFAMD_INLINE T butterfly32(T* p) {
	for (int s = 16; s >= 1; s >>= 1)
		for (int t = 0; t < s; ++t) p[t] += p[t + s];
	return p[0];
}

template <typename T>
NMFAMD_INLINE T dist_sq_impl(const T* a, const T* b, size_t m) {
	T p[32];
	for (int t = 0; t < 32; ++t) p[t] = T(0);
	size_t i = 0;
	for (; i + 32 <= m; i += 32)
		for (int t = 0; t < 32; ++t) { T d = a[i + t] - b[i + t]; p[t] = std::fma(d, d, p[t]); }
	for (int t = 0; i + t < m; ++t) { T d = a[i + t] - b[i + t]; p[t] = std::fma(d, d, p[t]); }
	return butterfly32(p);
}

template <typename T>
NMFAMD_INLINE T dot_impl(const T* a, const T* b, size_t m) {
	T p[32];
	for (int t = 0; t < 32; ++t) p[t] = T(0);
	size_t i = 0;
	for (; i + 32 <= m; i += 32)
		for (int t = 0; t < 32; ++t) p[t] = std::fma(a[i + t], b[i + t], p[t]);
	for (int t = 0; i + t < m; ++t) p[t] = std::fma(a[i + t], b[i + t], p[t]);
	return butterfly32(p);
}

// nearest centre of each column in [q0, q1); returns how many memberships changed
template <typename T>
NMFAMD_INLINE unsigned assign_impl(const T* data, size_t m, size_t q0, size_t q1, const T* clusters, size_t ldc, size_t k, unsigned* membership) {
	unsigned changed = 0;
	for (size_t q = q0; q < q1; ++q) {
		const T* x = data + q * m;
		unsigned best = 0;
		T bestd = dist_sq_impl(x, clusters, m);
		for (size_t c = 1; c < k; ++c) {
			T s = dist_sq_impl(x, clusters + c * ldc, m);
			if (s < bestd) { bestd = s; best = (unsigned)c; }   // strict '<': first minimum wins (kMeans.cu:66-71)
		}
		if (membership[q] != best) { membership[q] = best; ++changed; }
	}
	return changed;
}

// rows [i0, i1) of every non-empty centre = sum of its members (ascending column) / count
template <typename T>
NMFAMD_INLINE void centres_impl(const T* data, size_t m, size_t n, size_t i0, size_t i1, T* clusters, size_t ldc, size_t k,
                         const unsigned* membership, const unsigned* count) {
	for (size_t c = 0; c < k; ++c)
		if (count[c]) std::fill(clusters + c * ldc + i0, clusters + c * ldc + i1, T(0));
	for (size_t q = 0; q < n; ++q) {
		T* ctr = clusters + (size_t)membership[q] * ldc;
		const T* x = data + q * m;
		for (size_t i = i0; i < i1; ++i) ctr[i] += x[i];
	}
	for (size_t c = 0; c < k; ++c)
		if (count[c]) { T* ctr = clusters + c * ldc; const T cnt = T(count[c]); for (size_t i = i0; i < i1; ++i) ctr[i] /= cnt; }
}

// EIn-NMF memberships (MODE 0), max(W^T v, 0) (MODE 1) or |W^T v| (MODE 2) for columns [q0, q1)
template <typename T, int MODE>
NMFAMD_INLINE void h_from_centres_impl(const T* V, const T* W, size_t m, size_t r, size_t q0, size_t q1, T* H) {
	for (size_t q = q0; q < q1; ++q) {
		const T* x = V + q * m;
		T* h = H + q * r;
		if (MODE == 0) {
			// h_c = 1 / (d_c * sum_{c' <= c} 1 / (d_c' + 1e-9) + 1e-9): 32 centres at a time,
			// Hillis-Steele in double, the running carry rounded to T (EInNMF.cu:59-90)
			T carry = T(0);
			for (size_t base = 0; base < r; base += 32) {
				double v[32], nv[32];
				T dist[32];
				const size_t lanes = std::min<size_t>(32, r - base);
				for (size_t t = 0; t < lanes; ++t) dist[t] = dist_sq_impl(W + (base + t) * m, x, m);
				for (size_t t = 0; t < 32; ++t) v[t] = t < lanes ? 1.f / (dist[t] + 1.e-9) : 0.0;
				for (size_t s = 1; s < 32; s *= 2) {
					for (size_t t = 0; t < 32; ++t) nv[t] = t >= s ? v[t] + v[t - s] : v[t];
					std::memcpy(v, nv, sizeof(v));
				}
				for (size_t t = 0; t < lanes; ++t) {
					v[t] += carry;
					h[base + t] = (T)(1.f / std::fma((double)dist[t], v[t], 1.e-9));   // d * value + 1e-9 is one DFMA under nvcc
				}
				carry = (T)v[31];
			}
		} else {
			for (size_t c = 0; c < r; ++c) {
				T s = dot_impl(W + c * m, x, m);
				h[c] = MODE == 2 ? (T)std::fabs(s) : std::max(s, T(0));
			}
		}
	}
}

NMFAMD_CLONES unsigned assign_cols(const float* d, size_t m, size_t q0, size_t q1, const float* c, size_t ldc, size_t k, unsigned* mb) { return assign_impl(d, m, q0, q1, c, ldc, k, mb); }
NMFAMD_CLONES unsigned assign_cols(const double* d, size_t m, size_t q0, size_t q1, const double* c, size_t ldc, size_t k, unsigned* mb) { return assign_impl(d, m, q0, q1, c, ldc, k, mb); }
NMFAMD_CLONES void centre_rows(const float* d, size_t m, size_t n, size_t i0, size_t i1, float* c, size_t ldc, size_t k, const unsigned* mb, const unsigned* cnt) { centres_impl(d, m, n, i0, i1, c, ldc, k, mb, cnt); }
NMFAMD_CLONES void centre_rows(const double* d, size_t m, size_t n, size_t i0, size_t i1, double* c, size_t ldc, size_t k, const unsigned* mb, const unsigned* cnt) { centres_impl(d, m, n, i0, i1, c, ldc, k, mb, cnt); }
NMFAMD_CLONES void h_einnmf(const float* V, const float* W, size_t m, size_t r, size_t q0, size_t q1, float* H) { h_from_centres_impl<float, 0>(V, W, m, r, q0, q1, H); }
NMFAMD_CLONES void h_einnmf(const double* V, const double* W, size_t m, size_t r, size_t q0, size_t q1, double* H) { h_from_centres_impl<double, 0>(V, W, m, r, q0, q1, H); }
NMFAMD_CLONES void h_wtv_nonneg(const float* V, const float* W, size_t m, size_t r, size_t q0, size_t q1, float* H) { h_from_centres_impl<float, 1>(V, W, m, r, q0, q1, H); }
NMFAMD_CLONES void h_wtv_nonneg(const double* V, const double* W, size_t m, size_t r, size_t q0, size_t q1, double* H) { h_from_centres_impl<double, 1>(V, W, m, r, q0, q1, H); }
NMFAMD_CLONES void h_wtv_abs(const float* V, const float* W, size_t m, size_t r, size_t q0, size_t q1, float* H) { h_from_centres_impl<float, 2>(V, W, m, r, q0, q1, H); }
NMFAMD_CLONES void h_wtv_abs(const double* V, const double* W, size_t m, size_t r, size_t q0, size_t q1, double* H) { h_from_centres_impl<double, 2>(V, W, m, r, q0, q1, H); }

// Lloyd iterations; data m x n (ld m), clusters m x k (ld ldc).  Returns iterations done.
template <typename T>
unsigned lloyd(const T* data, size_t m, size_t n, T* clusters, size_t ldc, size_t k, unsigned* membership,
               unsigned seed, unsigned maxiter, double threshold) {
	// Forgy start: the first k entries of a shuffled index list (kMeans.cu:135-146)
	std::vector<unsigned> idx(n);
	std::mt19937 generator(seed);
	std::iota(idx.begin(), idx.end(), 0u);
	std::shuffle(idx.begin(), idx.end(), generator);
	for (size_t c = 0; c < k; ++c) std::memcpy(clusters + c * ldc, data + (size_t)idx[c] * m, sizeof(T) * m);

	std::fill(membership, membership + n, std::numeric_limits<unsigned>::max());
	// ~2 flops per element per centre: below ~1 Mflop per thread the fork/join costs more than it saves
	const size_t cols_per_thread = std::max<size_t>(1, (size_t)(5e5 / double(std::max<size_t>(m * k, 1))));
	auto assign = [&]() -> unsigned {
		std::vector<unsigned> part(host_threads() + 1, 0u);
		parallel_chunks(n, 1, cols_per_thread, [&](size_t q0, size_t q1, size_t slot) {
			part[slot] = assign_cols(data, m, q0, q1, clusters, ldc, k, membership);
		});
		return std::accumulate(part.begin(), part.end(), 0u);
	};

	unsigned iteration = 0;
	double change = 0.0;
	std::vector<unsigned> count(k);
	do {
		unsigned changed = assign();
		change = changed / double(n);
		if (changed > 0) {
			std::fill(count.begin(), count.end(), 0u);
			for (size_t q = 0; q < n; ++q) ++count[membership[q]];
			// an empty cluster keeps its centre (kMeans.cu:89-93); threads own disjoint row ranges
			const size_t rows_per_thread = std::max<size_t>(64, (size_t)(5e5 / double(std::max<size_t>(n, 1))));
			parallel_chunks(m, 16, rows_per_thread, [&](size_t i0, size_t i1, size_t) {
				centre_rows(data, m, n, i0, i1, clusters, ldc, k, membership, count.data());
			});
		}
	} while (++iteration < maxiter && change > threshold);
	if (change > 0.0) assign();   // final memberships against the last centres (kMeans.cu:262-270)
	return iteration;
}

template <typename T>
void fill_uniform(T* P, size_t r, size_t len, size_t ld, unsigned seed) {
	for (size_t y = 0; y < len; ++y)
		for (size_t c = 0; c < r; ++c) P[y * ld + c] = (T)uniform01(seed, y * r + c, sizeof(T) == 4);
}

} // namespace


// ---- NNDSVD ------------------------------------------------------------------------------------------------------------------------------------------
// Truncated SVD of V (m x n, column-major T) by block subspace iteration in double: Q (m x b, orthonormal columns) <- orth(V orth(V^T Q)) until the Ritz values
// stand still; then the SVD of the small B = Q^T V by one-sided Jacobi.  Blocks are ROW-major (a row = b contiguous doubles): both big products are
// "row of the result += scalar x row of the block", vectorised over b, and every output element is summed in one fixed order whatever the thread count.
namespace {

// The two big products of the subspace iteration, register-blocked (round 5): a micro-kernel keeps an RB x CB block of the result in registers (4 x 12 doubles =
// twelve AVX2 registers) over a chunk of the reduction index, the streamed operand goes through it once per chunk, and the chunks follow each other in ascending
// order -- every output element is still ONE chain of fused multiply-adds in ascending reduction index, whatever the thread count or the blocking.  The first
// version ("row of the result += scalar x row of the block" over all b columns, compiled for baseline x86-64) ran at 2.7 GFLOP/s per thread: 4.5 s of NNDSVD in
// front of 0.2 s of factorisation at config 2's size.
constexpr size_t NND_RB = 4, NND_CB = 12, NND_CHUNK = 256;

// acc(rb, cb) <- acc + sum_{t in [t0, t1)} a(rb, t) * B(t, cb); a(rb, t) = A[rb * a_rs + t * a_ts] (element type T), B(t, :) = Bm + t * b (row-major doubles)
template <typename T, size_t RB, size_t CB>
NMFAMD_INLINE void nnd_block(const T* A, size_t a_rs, size_t a_ts, const double* Bm, size_t b, size_t t0, size_t t1, double* C, size_t c_rs, bool first) {
	double acc[RB][CB];
	for (size_t r = 0; r < RB; ++r)
		for (size_t c = 0; c < CB; ++c) acc[r][c] = first ? 0.0 : C[r * c_rs + c];
	for (size_t t = t0; t < t1; ++t) {
		const double* z = Bm + t * b;
		for (size_t r = 0; r < RB; ++r) {
			const double v = (double)A[r * a_rs + t * a_ts];
			for (size_t c = 0; c < CB; ++c) acc[r][c] = std::fma(v, z[c], acc[r][c]);
		}
	}
	for (size_t r = 0; r < RB; ++r)
		for (size_t c = 0; c < CB; ++c) C[r * c_rs + c] = acc[r][c];
}

// the same for ragged edges (rb <= RB rows, cb <= CB columns): plain loops, same order of operations per element
template <typename T>
NMFAMD_INLINE void nnd_block_edge(const T* A, size_t a_rs, size_t a_ts, const double* Bm, size_t b, size_t t0, size_t t1, double* C, size_t c_rs, bool first, size_t rb, size_t cb) {
	for (size_t r = 0; r < rb; ++r)
		for (size_t c = 0; c < cb; ++c) {
			double acc = first ? 0.0 : C[r * c_rs + c];
			for (size_t t = t0; t < t1; ++t) acc = std::fma((double)A[r * a_rs + t * a_ts], Bm[t * b + c], acc);
			C[r * c_rs + c] = acc;
		}
}

// rows [i0, i1) of C (row-major, b columns) = sum over t in [0, T) of a(i, t) B(t, :): the reduction index in chunks, the result block by block inside a chunk
template <typename T>
NMFAMD_INLINE void nnd_panel_impl(const T* A, size_t a_rs, size_t a_ts, const double* Bm, size_t b, size_t Tn, double* C, size_t i0, size_t i1) {
	for (size_t t0 = 0; t0 < Tn; t0 += NND_CHUNK) {
		const size_t t1 = std::min(Tn, t0 + NND_CHUNK);
		const bool first = t0 == 0;
		for (size_t i = i0; i < i1; i += NND_RB) {
			const size_t rb = std::min(NND_RB, i1 - i);
			for (size_t j = 0; j < b; j += NND_CB) {
				const size_t cb = std::min(NND_CB, b - j);
				if (rb == NND_RB && cb == NND_CB) nnd_block<T, NND_RB, NND_CB>(A + i * a_rs, a_rs, a_ts, Bm + j, b, t0, t1, C + i * b + j, b, first);
				else nnd_block_edge<T>(A + i * a_rs, a_rs, a_ts, Bm + j, b, t0, t1, C + i * b + j, b, first, rb, cb);
			}
		}
	}
}

// (one clone per instruction set: the x86-64-v3 one runs the micro-kernel on vfmadd231pd, the baseline one calls fma() -- same bits)
NMFAMD_CLONES void nnd_panel(const float* A, size_t a_rs, size_t a_ts, const double* Bm, size_t b, size_t Tn, double* C, size_t i0, size_t i1) {
	nnd_panel_impl<float>(A, a_rs, a_ts, Bm, b, Tn, C, i0, i1);
}
NMFAMD_CLONES void nnd_panel(const double* A, size_t a_rs, size_t a_ts, const double* Bm, size_t b, size_t Tn, double* C, size_t i0, size_t i1) {
	nnd_panel_impl<double>(A, a_rs, a_ts, Bm, b, Tn, C, i0, i1);
}

// Y (m x b) = V Z, Z (n x b): a(i, k) = V[i + k m]
template <typename T>
void nnd_mul_v(const T* V, size_t m, size_t n, const double* Z, size_t b, double* Y) {
	parallel_chunks(m, 16, 64, [&](size_t i0, size_t i1, size_t) { nnd_panel(V, 1, m, Z, b, n, Y, i0, i1); });
}

// Z (n x b) = V^T Q, Q (m x b): a(k, i) = V[k m + i]
template <typename T>
void nnd_mul_vt(const T* V, size_t m, size_t n, const double* Q, size_t b, double* Z) {
	parallel_chunks(n, 8, 32, [&](size_t k0, size_t k1, size_t) { nnd_panel(V, m, 1, Q, b, m, Z, k0, k1); });
}

// Orthonormalises the columns of Y (rows x b, row-major) in place by Cholesky QR, twice; G = Y^T Y from fixed 512-row chunks added in chunk order.  A column that
// depends on its predecessors (rank-deficient V) becomes zero.
void nnd_orthonormalize(double* Y, size_t rows, size_t b) {
	for (int pass = 0; pass < 2; ++pass) {
		const size_t chunks = (rows + 511) / 512;
		std::vector<double> part(chunks * b * b, 0.0), G(b * b, 0.0), R(b * b, 0.0);
		parallel_chunks(chunks, 1, 1, [&](size_t c0, size_t c1, size_t) {
			for (size_t c = c0; c < c1; ++c) {
				double* g = part.data() + c * b * b;
				for (size_t i = c * 512; i < std::min(rows, (c + 1) * 512); ++i) {
					const double* y = Y + i * b;
					for (size_t p = 0; p < b; ++p) { const double yp = y[p]; if (yp == 0.0) continue; double* gr = g + p * b; for (size_t q = p; q < b; ++q) gr[q] += yp * y[q]; }
				}
			}
		});
		for (size_t c = 0; c < chunks; ++c) for (size_t e = 0; e < b * b; ++e) G[e] += part[c * b * b + e];
		// upper Cholesky factor R (G = R^T R), row by row
		double gmax = 0.0;
		for (size_t p = 0; p < b; ++p) gmax = std::max(gmax, G[p * b + p]);
		std::vector<char> dead(b, 0);
		for (size_t p = 0; p < b; ++p) {
			double d = G[p * b + p];
			for (size_t k = 0; k < p; ++k) d -= R[k * b + p] * R[k * b + p];
			if (!(d > 1e-26 * gmax) || gmax == 0.0) { dead[p] = 1; R[p * b + p] = 1.0; continue; }
			const double rpp = std::sqrt(d);
			R[p * b + p] = rpp;
			for (size_t q = p + 1; q < b; ++q) {
				double v = G[p * b + q];
				for (size_t k = 0; k < p; ++k) v -= R[k * b + p] * R[k * b + q];
				R[p * b + q] = v / rpp;
			}
		}
		// Y <- Y R^{-1}: row by row, forward over the columns
		parallel_chunks(rows, 1, 256, [&](size_t i0, size_t i1, size_t) {
			for (size_t i = i0; i < i1; ++i) {
				double* y = Y + i * b;
				for (size_t q = 0; q < b; ++q) {
					if (dead[q]) { y[q] = 0.0; continue; }
					double v = y[q];
					for (size_t k = 0; k < q; ++k) v -= y[k] * R[k * b + q];
					y[q] = v / R[q * b + q];
				}
			}
		});
	}
}

// G (b x b, both triangles) = Y^T Y of a row-major block: fixed 512-row chunks added in chunk order (the sums of nnd_orthonormalize)
void nnd_gram(const double* Y, size_t rows, size_t b, double* G) {
	const size_t chunks = (rows + 511) / 512;
	std::vector<double> part(chunks * b * b, 0.0);
	parallel_chunks(chunks, 1, 1, [&](size_t c0, size_t c1, size_t) {
		for (size_t c = c0; c < c1; ++c) {
			double* g = part.data() + c * b * b;
			for (size_t i = c * 512; i < std::min(rows, (c + 1) * 512); ++i) {
				const double* y = Y + i * b;
				for (size_t p = 0; p < b; ++p) { const double yp = y[p]; if (yp == 0.0) continue; double* gr = g + p * b; for (size_t q = p; q < b; ++q) gr[q] += yp * y[q]; }
			}
		}
	});
	std::fill(G, G + b * b, 0.0);
	for (size_t c = 0; c < chunks; ++c) for (size_t e = 0; e < b * b; ++e) G[e] += part[c * b * b + e];
	for (size_t p = 0; p < b; ++p) for (size_t q = p + 1; q < b; ++q) G[q * b + p] = G[p * b + q];
}

// Eigen-decomposition of a symmetric b x b matrix by cyclic Jacobi rotations: on return the diagonal of A holds the eigenvalues and, when Rv is given, its columns
// the eigenvectors (A_in = Rv diag Rv^T).  O(b^3) per sweep: what the subspace iteration can afford at every step (the one-sided Jacobi SVD of the n x b block,
// round 5's first version of the convergence test, took 0.33 s per step at n = 2 000 -- 21 of NNDSVD's 24 seconds).
void nnd_sym_jacobi(double* A, size_t b, double* Rv) {
	if (Rv) { std::fill(Rv, Rv + b * b, 0.0); for (size_t j = 0; j < b; ++j) Rv[j * b + j] = 1.0; }
	for (int sweep = 0; sweep < 60; ++sweep) {
		double off = 0.0, diag = 0.0;
		for (size_t p = 0; p < b; ++p) { diag = std::max(diag, std::fabs(A[p * b + p])); for (size_t q = p + 1; q < b; ++q) off = std::max(off, std::fabs(A[p * b + q])); }
		if (off <= 1e-17 * diag || off == 0.0) break;
		for (size_t p = 0; p + 1 < b; ++p)
			for (size_t q = p + 1; q < b; ++q) {
				const double apq = A[p * b + q];
				if (apq == 0.0) continue;
				const double zeta = (A[q * b + q] - A[p * b + p]) / (2.0 * apq);
				const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
				const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
				for (size_t k = 0; k < b; ++k) { const double x = A[k * b + p], y = A[k * b + q]; A[k * b + p] = c * x - sn * y; A[k * b + q] = sn * x + c * y; }
				for (size_t k = 0; k < b; ++k) { const double x = A[p * b + k], y = A[q * b + k]; A[p * b + k] = c * x - sn * y; A[q * b + k] = sn * x + c * y; }
				if (Rv) for (size_t k = 0; k < b; ++k) { const double x = Rv[k * b + p], y = Rv[k * b + q]; Rv[k * b + p] = c * x - sn * y; Rv[k * b + q] = sn * x + c * y; }
			}
	}
}

// One-sided Jacobi SVD of A (rows x b, row-major): on return the columns of A are mutually orthogonal (A <- A Rot), rot (b x b, row-major) = Rot, sigma[j] = the
// norm of column j.  A_in = (A_out columns / sigma) diag(sigma) Rot^T.  rot_given: rot already holds a rotation that has been applied to A (a preconditioner:
// the eigenvectors of A^T A leave the columns orthogonal up to rounding, and the sweeps below only polish) -- it is carried on instead of the identity.
void nnd_jacobi(double* A, size_t rows, size_t b, double* rot, double* sigma, bool rot_given = false) {
	if (!rot_given) {
		std::fill(rot, rot + b * b, 0.0);
		for (size_t j = 0; j < b; ++j) rot[j * b + j] = 1.0;
	}
	for (int sweep = 0; sweep < 60; ++sweep) {
		double off = 0.0;
		for (size_t p = 0; p + 1 < b; ++p)
			for (size_t q = p + 1; q < b; ++q) {
				double app = 0, aqq = 0, apq = 0;
				for (size_t i = 0; i < rows; ++i) { const double x = A[i * b + p], y = A[i * b + q]; app += x * x; aqq += y * y; apq += x * y; }
				if (std::fabs(apq) <= 1e-15 * std::sqrt(app * aqq) || apq == 0.0) continue;
				off = std::max(off, std::fabs(apq) / std::sqrt(app * aqq));
				const double zeta = (aqq - app) / (2.0 * apq);
				const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
				const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
				for (size_t i = 0; i < rows; ++i) { const double x = A[i * b + p], y = A[i * b + q]; A[i * b + p] = c * x - sn * y; A[i * b + q] = sn * x + c * y; }
				for (size_t i = 0; i < b; ++i) { const double x = rot[i * b + p], y = rot[i * b + q]; rot[i * b + p] = c * x - sn * y; rot[i * b + q] = sn * x + c * y; }
			}
		if (off <= 1e-15) break;
	}
	for (size_t j = 0; j < b; ++j) { double s = 0; for (size_t i = 0; i < rows; ++i) s += A[i * b + j] * A[i * b + j]; sigma[j] = std::sqrt(s); }
}

// U (m x r, column-major), S (r), Vr (n x r, column-major): the r largest singular triplets of V
template <typename T>
void truncated_svd(const T* V, size_t m, size_t n, size_t r, std::vector<double>& U, std::vector<double>& S, std::vector<double>& Vr) {
	const size_t full = std::min(m, n);
	const size_t b = std::min(full, r + std::max<size_t>(8, r / 2));
	std::vector<double> Q(m * b), Z(n * b), rot(b * b), sig(b), prev(b, 0.0);
	// start: Z = a fixed pseudo-random block (the counter generator of the device fill: reproducible), Q = orth(V Z)
	for (size_t e = 0; e < n * b; ++e) Z[e] = uniform01(0x4e4e4453ull, e, false) - 0.5;
	nnd_orthonormalize(Z.data(), n, b);
	nnd_mul_v(V, m, n, Z.data(), b, Q.data());
	nnd_orthonormalize(Q.data(), m, b);
	// (a work budget: ~2e11 multiply-adds for the iteration -- six seconds of the host's threads at config 2's size, where a random matrix's flat spectrum would
	//  otherwise keep the block turning for minutes; a start value needs the leading directions, not their last digits)
	const double per_it = 4.0 * (double)m * (double)n * (double)b;
	const int max_it = b == full ? 2 : (int)std::max(8.0, std::min(300.0, 2e11 / per_it));
	// Per step the Ritz values come from the b x b matrix Z^T Z (nnd_sym_jacobi); the SVD of the block itself -- one-sided Jacobi, which does not square the
	// condition number -- runs ONCE, on the last Z, behind a rotation by the eigenvectors of Z^T Z (its sweeps then only polish).
	std::vector<double> Zs, G(b * b), Rv(b * b);
	for (int it = 0; it < max_it; ++it) {
		nnd_mul_vt(V, m, n, Q.data(), b, Z.data());                 // Z = V^T Q = B^T
		nnd_gram(Z.data(), n, b, G.data());
		nnd_sym_jacobi(G.data(), b, nullptr);
		std::vector<double> sorted(b);
		for (size_t j = 0; j < b; ++j) sorted[j] = std::sqrt(std::max(G[j * b + j], 0.0));
		std::sort(sorted.begin(), sorted.end(), std::greater<double>());
		double change = 0.0;
		for (size_t j = 0; j < std::min(r, b); ++j) change = std::max(change, std::fabs(sorted[j] - prev[j]));
		prev = sorted;
		if (it + 1 >= max_it || (it >= 2 && change <= 1e-14 * sorted[0])) break;
		nnd_orthonormalize(Z.data(), n, b);
		nnd_mul_v(V, m, n, Z.data(), b, Q.data());
		nnd_orthonormalize(Q.data(), m, b);
	}
	{
		nnd_gram(Z.data(), n, b, G.data());
		nnd_sym_jacobi(G.data(), b, Rv.data());
		Zs.assign(n * b, 0.0);
		parallel_chunks(n, 1, 256, [&](size_t i0, size_t i1, size_t) {
			for (size_t i = i0; i < i1; ++i) {
				const double* z = Z.data() + i * b;
				double* o = Zs.data() + i * b;
				for (size_t k = 0; k < b; ++k) { const double zk = z[k]; const double* rv = Rv.data() + k * b; for (size_t c = 0; c < b; ++c) o[c] += zk * rv[c]; }
			}
		});
		rot = Rv;
		nnd_jacobi(Zs.data(), n, b, rot.data(), sig.data(), true);
	}
	// B^T = Z = (Zs / sig) diag(sig) rot^T  =>  V ~ Q B = (Q rot) diag(sig) (Zs / sig)^T: left vectors Q rot, right vectors Zs / sig
	std::vector<size_t> order(b);
	std::iota(order.begin(), order.end(), size_t(0));
	std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return sig[x] > sig[y]; });
	U.assign(m * r, 0.0); S.assign(r, 0.0); Vr.assign(n * r, 0.0);
	for (size_t j = 0; j < std::min(r, b); ++j) {
		const size_t c = order[j];
		S[j] = sig[c];
		if (!(sig[c] > 0.0)) continue;
		for (size_t i = 0; i < n; ++i) Vr[j * n + i] = Zs[i * b + c] / sig[c];
		parallel_chunks(m, 1, 2048, [&](size_t i0, size_t i1, size_t) {
			for (size_t i = i0; i < i1; ++i) { double s = 0; const double* q = Q.data() + i * b; for (size_t k = 0; k < b; ++k) s += q[k] * rot[k * b + c]; U[j * m + i] = s; }
		});
	}
}

} // namespace

int nndsvd_variant(const Parameter* parameters, unsigned count) {
	for (unsigned i = 0; i < count; ++i)
		if (parameters != nullptr && parameters[i].name != nullptr && std::strcmp(parameters[i].name, "nndsvd") == 0) {
			const double v = parameters[i].value;
			return v == 1.0 ? 1 : (v == 2.0 ? 2 : 0);      // (nmfgpu::compute rejects anything but 0 / 1 / 2 before it gets here, abi.cpp)
		}
	return -1;
}

// W (m x r, ld m), H (r x n, ld r; may be null).  variant 0: NNDSVD (zeros stay), 1: NNDSVDa (zeros -> mean of V), 2: NNDSVDar (zeros -> mean / 100 x U(0, 1], seeded)
template <typename T>
void nndsvd(const T* V, size_t m, size_t n, size_t r, int variant, unsigned seed, T* W, T* H) {
	std::vector<double> U, S, Vr;
	truncated_svd(V, m, n, r, U, S, Vr);
	std::vector<double> Wd(m * r, 0.0), Hd(r * n, 0.0);
	for (size_t j = 0; j < r; ++j) {
		const double* x = U.data() + j * m;
		const double* y = Vr.data() + j * n;
		if (j == 0) {
			// the leading pair of a non-negative matrix has one sign (Perron-Frobenius): its absolute values
			const double s = std::sqrt(S[0]);
			for (size_t i = 0; i < m; ++i) Wd[i] = s * std::fabs(x[i]);
			for (size_t i = 0; i < n; ++i) Hd[i * r] = s * std::fabs(y[i]);
			continue;
		}
		double xp = 0, xn = 0, yp = 0, yn = 0;
		for (size_t i = 0; i < m; ++i) { if (x[i] > 0) xp += x[i] * x[i]; else xn += x[i] * x[i]; }
		for (size_t i = 0; i < n; ++i) { if (y[i] > 0) yp += y[i] * y[i]; else yn += y[i] * y[i]; }
		xp = std::sqrt(xp); xn = std::sqrt(xn); yp = std::sqrt(yp); yn = std::sqrt(yn);
		const double mp = xp * yp, mn = xn * yn;
		const bool pos = mp > mn;
		const double sigma = pos ? mp : mn, nx = pos ? xp : xn, ny = pos ? yp : yn;
		if (!(sigma > 0.0) || !(S[j] > 0.0)) continue;
		const double l = std::sqrt(S[j] * sigma);
		for (size_t i = 0; i < m; ++i) { const double v = pos ? std::max(x[i], 0.0) : std::max(-x[i], 0.0); Wd[j * m + i] = l * v / nx; }
		for (size_t i = 0; i < n; ++i) { const double v = pos ? std::max(y[i], 0.0) : std::max(-y[i], 0.0); Hd[i * r + j] = l * v / ny; }
	}
	// entries below machine precision count as zeros (as the reference implementations of the method do)
	const double tiny = 1e-6 * (sizeof(T) == 4 ? 1.0 : 1e-4);
	double mean = 0.0;
	if (variant != 0) { for (size_t e = 0; e < m * n; ++e) mean += (double)V[e]; mean /= (double)(m * n); }
	auto finish = [&](std::vector<double>& P, unsigned long long stream) {
		for (size_t e = 0; e < P.size(); ++e) {
			if (P[e] < tiny) P[e] = variant == 0 ? 0.0 : (variant == 1 ? mean : mean * 0.01 * uniform01((unsigned long long)seed + stream, e, false));
		}
	};
	finish(Wd, 0); finish(Hd, 0x9e3779b9ull);
	for (size_t e = 0; e < m * r; ++e) W[e] = (T)Wd[e];
	if (H) for (size_t e = 0; e < r * n; ++e) H[e] = (T)Hd[e];
}

template <typename T>
bool initialize(const NmfDescription<T>& d, T* W, T* H) {
	const size_t m = d.inputMatrix.rows, n = d.inputMatrix.columns, r = d.features;
	const std::vector<T> V = to_dense(d.inputMatrix);
	// Parameter "nndsvd" present: the SVD-based start, whatever initMethod says (the enum of the reference has no member for it)
	if (const int variant = nndsvd_variant(d.parameters, d.numParameters); variant >= 0) {
		nndsvd<T>(V.data(), m, n, r, variant, d.seed, W, H);
		return true;
	}
	switch (d.initMethod) {
	case NmfInitializationMethod::MeanColumns: {
		const unsigned meanCount = 5;
		std::mt19937 gen(d.seed);
		std::uniform_int_distribution<unsigned> pick(0, (unsigned)n - 1);
		for (size_t c = 0; c < r; ++c) {
			unsigned cols[meanCount];
			for (unsigned i = 0; i < meanCount; ++i) cols[i] = pick(gen);
			for (size_t i = 0; i < m; ++i) {
				T s = 0;
				for (unsigned q = 0; q < meanCount; ++q) s += V[(size_t)cols[q] * m + i];
				W[c * m + i] = s / meanCount;
			}
		}
		// the W panel is stored transposed on the device; here W is plain m x r and H is r x n
		if (H) fill_uniform(H, r, n, r, d.seed);
		return true;
	}
	case NmfInitializationMethod::KMeansAndRandomValues:
	case NmfInitializationMethod::KMeansAndAbsoluteWTV:
	case NmfInitializationMethod::KMeansAndNonNegativeWTV:
	case NmfInitializationMethod::EInNMF: {
		std::vector<unsigned> membership(n);
		lloyd<T>(V.data(), m, n, W, m, r, membership.data(), d.seed, 100, 0.005);
		if (!H) return true;
		if (d.initMethod == NmfInitializationMethod::KMeansAndRandomValues) {
			fill_uniform(H, r, n, r, d.seed + 1);
			return true;
		}
		const size_t cols_per_thread = std::max<size_t>(1, (size_t)(5e5 / double(std::max<size_t>(m * r, 1))));
		parallel_chunks(n, 1, cols_per_thread, [&](size_t q0, size_t q1, size_t) {
			if (d.initMethod == NmfInitializationMethod::EInNMF) h_einnmf(V.data(), W, m, r, q0, q1, H);
			else if (d.initMethod == NmfInitializationMethod::KMeansAndAbsoluteWTV) h_wtv_abs(V.data(), W, m, r, q0, q1, H);
			else h_wtv_nonneg(V.data(), W, m, r, q0, q1, H);
		});
		return true;
	}
	default:
		return false;
	}
}

template <typename T>
ResultType compute_kmeans(KMeansDescription<T>& desc, KMeansSummary* summary) {
	// argument checks of Interface.cpp:366-389
	if (desc.numClusters >= desc.inputMatrix.columns) {
		std::cerr << " [ERROR] Number of clusters must be smaller than number of samples in dataset!" << std::endl;
		return ResultType::ErrorInvalidArgument;
	}
	if (desc.outputMatrixClusters.format != StorageFormat::Dense) {
		std::cerr << " [ERROR] Cluster matrix must have a dense storage format!" << std::endl;
		return ResultType::ErrorInvalidArgument;
	}
	if (desc.inputMatrix.rows != desc.outputMatrixClusters.rows) {
		std::cerr << " [ERROR] Input and output matrices must have the same amount of rows!" << std::endl;
		return ResultType::ErrorInvalidArgument;
	}
	if (desc.numClusters == 0) {
		std::cerr << " [ERROR] Number of clusters must be smaller than column count of the input matrix!" << std::endl;
		return ResultType::ErrorInvalidArgument;
	}
	const size_t m = desc.inputMatrix.rows, n = desc.inputMatrix.columns, k = desc.numClusters;
	std::vector<T> V;
	try { V = to_dense(desc.inputMatrix); } catch (const std::bad_alloc&) { return ResultType::ErrorNotEnoughHostMemory; }
	std::vector<unsigned> membership(n);
	unsigned iterations = lloyd<T>(V.data(), m, n, desc.outputMatrixClusters.dense.values, desc.outputMatrixClusters.dense.leadingDimension, k,
	                               membership.data(), desc.seed, desc.numIterations, desc.thresholdValue);
	if (desc.outputMemberships) std::memcpy(desc.outputMemberships, membership.data(), sizeof(unsigned) * n);
	if (summary) summary->iterations = iterations;   // the reference leaves the summary untouched (Interface.cpp:404-406)
	return ResultType::Success;
}

template bool initialize<float>(const NmfDescription<float>&, float*, float*);
template bool initialize<double>(const NmfDescription<double>&, double*, double*);
template ResultType compute_kmeans<float>(KMeansDescription<float>&, KMeansSummary*);
template ResultType compute_kmeans<double>(KMeansDescription<double>&, KMeansSummary*);

} // namespace hostinit
} // namespace nmfgpu
