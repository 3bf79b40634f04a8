// kmeans_oracle.cpp -- CPU restatement of the reference's k-means and EIn-NMF initialisers.
//
// TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library; the shipped engine never does.
//
// PARITY UNPINNED: the reference's k-means is CUDA (source/kmeans/kMeans.cu, source/init/EInNMF.cu),
// cannot be compiled here, and the reference holds no golden vectors for it.  What anchors this file
// is the reference's text alone: each function below walks the kernel it cites thread by thread
// (32 lanes, the same shuffles), in scalar C++ with explicit fma() where nvcc contracts by default.
//
// Two defects of the reference are NOT restated (the engine does not reproduce them either):
//   * memberships are compared against uninitialised device memory in the first pass
//     (kMeans.cu:53-78 reads dataClusterMembership before anything wrote it); here they start
//     at UINT_MAX, so every column counts as changed in pass 0;
//   * the centre kernel's grid is halved (kMeans.cu:222-225), which skips the last 32-row block
//     whenever ceil(rows/32) is odd and > 1; here every row is updated.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <random>
#include <utility>
#include <vector>

namespace {

// sumWarpReduction, source/nmf/KernelHelper.cuh:31-43: var += shfl_xor(var, 16 / 8 / 4 / 2 / 1).
// Every lane ends with the same bits (IEEE addition commutes), so lane 0 is "the" result.
template <typename T>
T warp_sum(const T lane[32]) {
	T v[32], w[32];
	std::memcpy(v, lane, sizeof(v));
	for (int mask = 16; mask >= 1; mask >>= 1) {
		for (int t = 0; t < 32; ++t) w[t] = v[t] + v[t ^ mask];
		std::memcpy(v, w, sizeof(v));
	}
	return v[0];
}

// distanceSq, source/kmeans/kMeans.cu:41-50 (and source/init/EInNMF.cu:30-39, same body):
// lane t accumulates rows t, t+32, ...; "sum += diff * diff" is an FFMA/DFMA under nvcc's default -fmad=true.
template <typename T>
T distance_sq(const T* a, const T* b, unsigned rows) {
	T lane[32];
	for (unsigned t = 0; t < 32; ++t) {
		T sum = T(0);
		for (unsigned i = t; i < rows; i += 32) {
			T diff = a[i] - b[i];
			sum = std::fma(diff, diff, sum);
		}
		lane[t] = sum;
	}
	return warp_sum(lane);
}

// kernelUpdateMembership, kMeans.cu:53-78
template <typename T>
unsigned update_membership(const T* data, long ld, unsigned rows, unsigned n, const T* clusters, long ldc, unsigned k, unsigned* membership) {
	unsigned changed = 0;
	for (unsigned id = 0; id < n; ++id) {
		unsigned best = 0;
		T best_distance = distance_sq(data + (size_t)id * ld, clusters, rows);
		for (unsigned c = 1; c < k; ++c) {
			T distance = distance_sq(data + (size_t)id * ld, clusters + (size_t)c * ldc, rows);
			if (distance < best_distance) { best_distance = distance; best = c; }
		}
		if (membership[id] != best) { membership[id] = best; ++changed; }
	}
	return changed;
}

// computeKMeans, kMeans.cu:126-278
template <typename T>
unsigned kmeans(const T* data, long ld, unsigned rows, unsigned n, T* clusters, long ldc, unsigned k, unsigned* membership,
                unsigned seed, unsigned maxiter, double threshold) {
	// Forgy start, kMeans.cu:135-146
	std::vector<unsigned> random_indices(n);
	std::mt19937 generator(seed);
	std::iota(random_indices.begin(), random_indices.end(), 0u);
	std::shuffle(random_indices.begin(), random_indices.end(), generator);
	for (unsigned c = 0; c < k; ++c)
		std::memcpy(clusters + (size_t)c * ldc, data + (size_t)random_indices[c] * ld, sizeof(T) * rows);

	std::fill(membership, membership + n, UINT32_MAX);
	std::vector<std::pair<unsigned, unsigned>> info;
	unsigned iteration = 0;
	double percentage_change = 0.0;
	do {
		unsigned changed = update_membership(data, ld, rows, n, clusters, ldc, k, membership);   // :165-175
		percentage_change = changed / double(n);
		if (changed > 0) {
			// members sorted by (cluster, column), kMeans.cu:185-189, entry points/counts :198-210
			info.clear();
			for (unsigned i = 0; i < n; ++i) info.emplace_back(membership[i], i);
			std::sort(info.begin(), info.end());
			std::vector<unsigned> entry_point(k, 0u), entry_count(k, 0u);
			unsigned last = UINT32_MAX, index = 0;
			for (const auto& m : info) {
				++entry_count[m.first];
				if (last != m.first) { entry_point[m.first] = index; last = m.first; }
				++index;
			}
			// kernelUpdateClusterCenters, kMeans.cu:81-122: one thread per (row, cluster), members in
			// sorted order, then a division by the count; empty clusters return early (:89-93)
			for (unsigned c = 0; c < k; ++c) {
				if (entry_count[c] == 0) continue;
				for (unsigned row = 0; row < rows; ++row) {
					T sum = T(0);
					for (unsigned i = 0; i < entry_count[c]; ++i) sum += data[(size_t)info[entry_point[c] + i].second * ld + row];
					sum /= T(entry_count[c]);
					clusters[(size_t)c * ldc + row] = sum;
				}
			}
		}
	} while (++iteration < maxiter && percentage_change > threshold);
	if (percentage_change > 0.0) update_membership(data, ld, rows, n, clusters, ldc, k, membership);   // :262-270
	return iteration;
}

// computeDistanceMatrix + kernel_EInNMF_prefix_scan_kepler, source/init/EInNMF.cu:44-119.
// H (r x n) first holds squared distances centre-to-column, then per column, 32 centres at a time:
//   value = 1.f / (d + 1.e-9)            -- double arithmetic whatever NumericType is (:67)
//   inclusive Hillis-Steele scan with shfl_up (:70-76), value += previousSum (:78)
//   H = 1.f / (d * value + 1.e-9) (:81);  previousSum = shfl(value, 31), stored as NumericType (:84)
template <typename T>
void einnmf_h(const T* V, long ldv, const T* W, long ldw, unsigned rows, unsigned r, unsigned n, T* H, long ldh) {
	for (unsigned col = 0; col < n; ++col) {
		T* h = H + (size_t)col * ldh;
		for (unsigned c = 0; c < r; ++c) h[c] = distance_sq(W + (size_t)c * ldw, V + (size_t)col * ldv, rows);
		T previous_sum = T(0);
		for (unsigned base = 0; base < r; base += 32) {
			double value[32], saved[32], shifted[32];
			const unsigned active = std::min(32u, r - base);
			for (unsigned t = 0; t < 32; ++t) {
				saved[t] = t < active ? (double)h[base + t] : 0.0;
				value[t] = t < active ? 1.f / (saved[t] + 1.e-9) : 0.0;   // lanes past m have left the loop; nobody reads them
			}
			for (unsigned i = 1; i < 32; i *= 2) {
				for (unsigned t = 0; t < 32; ++t) shifted[t] = t >= i ? value[t - i] : 0.0;
				for (unsigned t = 0; t < 32; ++t) if (t >= i) value[t] += shifted[t];
			}
			for (unsigned t = 0; t < 32; ++t) value[t] += previous_sum;
			for (unsigned t = 0; t < active; ++t) h[base + t] = (T)(1.f / std::fma(saved[t], value[t], 1.e-9));   // mul+add: one DFMA under -fmad=true
			previous_sum = (T)value[31];
		}
	}
}

} // namespace

extern "C" {

unsigned oracle_kmeans_f32(const float* data, long ld, unsigned rows, unsigned n, float* clusters, long ldc, unsigned k,
                           unsigned* membership, unsigned seed, unsigned maxiter, double threshold) {
	return kmeans<float>(data, ld, rows, n, clusters, ldc, k, membership, seed, maxiter, threshold);
}
unsigned oracle_kmeans_f64(const double* data, long ld, unsigned rows, unsigned n, double* clusters, long ldc, unsigned k,
                           unsigned* membership, unsigned seed, unsigned maxiter, double threshold) {
	return kmeans<double>(data, ld, rows, n, clusters, ldc, k, membership, seed, maxiter, threshold);
}
void oracle_einnmf_h_f32(const float* V, long ldv, const float* W, long ldw, unsigned rows, unsigned r, unsigned n, float* H, long ldh) {
	einnmf_h<float>(V, ldv, W, ldw, rows, r, n, H, ldh);
}
void oracle_einnmf_h_f64(const double* V, long ldv, const double* W, long ldw, unsigned rows, unsigned r, unsigned n, double* H, long ldh) {
	einnmf_h<double>(V, ldv, W, ldw, rows, r, n, H, ldh);
}

} // extern "C"
