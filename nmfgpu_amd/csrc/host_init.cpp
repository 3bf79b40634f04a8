// host_init.cpp -- the initialisation strategies that run once per run, outside the iteration
// loop, kept on the host (BASELINE.json north star: "the k-means/NNDSVD init stays host-side C++").
//
// Restates, on the CPU with OpenMP:
//   k-means (Lloyd, Forgy start)        source/kmeans/kMeans.cu:126-278
//   KMeans* strategies                  source/init/KMeansStrategy.cpp:31-65
//   EIn-NMF membership transform        source/init/EInNMF.cu:44-91
//   MeanColumns                         source/init/MeanColumnStrategy.cpp:43-56, KernelMeanColumn.cu:30-50
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
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <numeric>
#include <random>
#include <vector>

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
	auto assign = [&]() -> unsigned {
		unsigned changed = 0;
#pragma omp parallel for schedule(static) reduction(+ : changed)
		for (long q = 0; q < (long)n; ++q) {
			const T* x = data + (size_t)q * m;
			unsigned best = 0; T bestd = std::numeric_limits<T>::max();
			for (size_t c = 0; c < k; ++c) {
				const T* ctr = clusters + c * ldc;
				T s = 0;
				for (size_t i = 0; i < m; ++i) { T diff = x[i] - ctr[i]; s += diff * diff; }
				if (c == 0 || s < bestd) { bestd = s; best = (unsigned)c; }   // strict '<': first minimum wins (kMeans.cu:66-71)
			}
			if (membership[q] != best) { membership[q] = best; ++changed; }
		}
		return changed;
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
			// centroid = mean of its members; an empty cluster keeps its centre (kMeans.cu:89-93)
#pragma omp parallel for schedule(static)
			for (long c = 0; c < (long)k; ++c) {
				if (count[c] == 0) continue;
				T* ctr = clusters + (size_t)c * ldc;
				std::fill(ctr, ctr + m, T(0));
				for (size_t q = 0; q < n; ++q)
					if (membership[q] == (unsigned)c) { const T* x = data + q * m; for (size_t i = 0; i < m; ++i) ctr[i] += x[i]; }
				for (size_t i = 0; i < m; ++i) ctr[i] /= T(count[c]);
			}
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

template <typename T>
bool initialize(const NmfDescription<T>& d, T* W, T* H) {
	const size_t m = d.inputMatrix.rows, n = d.inputMatrix.columns, r = d.features;
	const std::vector<T> V = to_dense(d.inputMatrix);
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
#pragma omp parallel for schedule(static)
		for (long q = 0; q < (long)n; ++q) {
			const T* x = V.data() + (size_t)q * m;
			T* h = H + (size_t)q * r;
			if (d.initMethod == NmfInitializationMethod::EInNMF) {
				// h_k = 1 / (d_k * sum_{k' <= k} 1 / (d_k' + 1e-9) + 1e-9), d = squared distance to centroid k
				T prefix = 0;
				for (size_t c = 0; c < r; ++c) {
					const T* w = W + c * m;
					T dist = 0;
					for (size_t i = 0; i < m; ++i) { T diff = w[i] - x[i]; dist += diff * diff; }
					prefix += (T)(1.f / (dist + 1.e-9));
					h[c] = (T)(1.f / (dist * prefix + 1.e-9));
				}
			} else {
				for (size_t c = 0; c < r; ++c) {
					const T* w = W + c * m;
					T s = 0;
					for (size_t i = 0; i < m; ++i) s += w[i] * x[i];
					h[c] = d.initMethod == NmfInitializationMethod::KMeansAndAbsoluteWTV ? (T)std::fabs(s) : std::max(s, T(0));
				}
			}
		}
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
