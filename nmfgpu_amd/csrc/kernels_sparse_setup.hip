// kernels_sparse_setup.hip -- the CSR and CSC images of a sparse V, built ON THE DEVICE from the caller's CSR / CSC / COO arrays (round 6).
//
// The reference converts sparse input on the device (cuSPARSE, source/common/Matrix.h:145-232: csr2dense / csc2dense / xcoo2csr with the caller's index base);
// the sparse-compute extension keeps V sparse and needs both orientations.  Rounds 1 - 5 built them on the host with stable counting sorts: 456 ms for config 3's
// 2 * 10^7 entries, longer than 300 iterations of the factorisation (profiles/r05_compute_paths.txt).  Here: the caller's arrays are uploaded as they are and
//   1. k_sp_expand:        0-based (row, column) of every stored entry (CSR / CSC: the outer index by binary search in the pointer array), range / pointer checks
//   2. k_sp_order_check:   is the sequence already sorted by (row, column)?  (what a CSR input usually is: then the CSR image is the input itself)
//   3. otherwise:          row histogram -> exclusive scan -> the entries' positions p dealt to their rows (atomics: any order) -> every row's list sorted by
//                          (column, p) in LDS (bitonic) = ascending columns, duplicates in input order: exactly the host's stable sorts
//   4. CSC image:          column histogram -> scan -> the CSR positions q dealt to their columns -> every column's list of q sorted ascending in LDS:
//                          CSR order is (row, column) order, so ascending q is ascending row with duplicates in CSR order -- the host's stable counting sort
//   5. k_sp_col_sumsq:     tr(V^T V) terms per column in the image's order (one thread per column, separate multiply and add: the host loop's bits) and the
//                          column sums in double for the KL divergence
//   6. k_sp_boundaries:    the per-(row, block) boundary pointers of the blocked KL gather
// The atomics only decide where inside its segment an entry waits for the sort: the images are the same bits every run and the same bits as the host path's
// (tests/test_gpu_sparse_setup.py).  Inputs the kernels do not cover fall back to the host path (Engine::upload_sparse says which): entries outside the matrix
// or outside every row's pointer range (the host path drops them), pointer arrays that do not ascend, a row or column longer than the LDS sort takes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace nmfamd {

// flags word: bit 0 = an entry outside the matrix / outside every pointer range, bit 1 = pointer array does not ascend, bit 2 = not sorted by (row, column)
__global__ __launch_bounds__(256) void k_sp_expand(int format, const int* __restrict__ a, const int* __restrict__ b, long nnz, int outer, int base, int m, int n,
                                                   int* __restrict__ row, int* __restrict__ col, int* __restrict__ flags) {
	const long p = (long)blockIdx.x * 256 + threadIdx.x;
	if (format != 3 && p < outer) {
		if (a[p] > a[p + 1]) atomicOr(flags, 2);
	}
	if (p >= nnz) return;
	int i, j;
	if (format == 3) { i = a[p] - base; j = b[p] - base; }
	else {
		// the outer index o with a[o] - base <= p < a[o + 1] - base: upper bound of p + base in a[0 .. outer], minus one
		const long key = p + base;
		int lo = 0, hi = outer + 1;
		while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long)a[mid] <= key) lo = mid + 1; else hi = mid; }
		const int o = lo - 1;
		const int inner = b[p] - base;
		if (o < 0 || o >= outer) { atomicOr(flags, 1); i = j = 0; }
		else if (format == 1) { i = o; j = inner; }
		else { i = inner; j = o; }
	}
	if (i < 0 || i >= m || j < 0 || j >= n) { atomicOr(flags, 1); i = j = 0; }
	row[p] = i; col[p] = j;
}

__global__ __launch_bounds__(256) void k_sp_order_check(const int* __restrict__ row, const int* __restrict__ col, long nnz, int* __restrict__ flags) {
	const long p = (long)blockIdx.x * 256 + threadIdx.x + 1;
	if (p >= nnz) return;
	const int r0 = row[p - 1], r1 = row[p];
	if (r0 > r1 || (r0 == r1 && col[p - 1] > col[p])) atomicOr(flags, 4);
}

// counts[key[p]] += 1
__global__ __launch_bounds__(256) void k_sp_histogram(const int* __restrict__ key, long count, int* __restrict__ counts) {
	const long p = (long)blockIdx.x * 256 + threadIdx.x;
	if (p < count) atomicAdd(counts + key[p], 1);
}

// ptr[0 .. count] = exclusive scan of counts[0 .. count - 1]; maxlen[0] = the largest count.  ONE workgroup of 1024 threads (count is a row / column count).
__global__ __launch_bounds__(1024) void k_sp_scan(const int* __restrict__ counts, int count, int* __restrict__ ptr, int* __restrict__ maxlen) {
	__shared__ int s_sum[1024];
	__shared__ int s_max[1024];
	const int tid = threadIdx.x;
	const long per = ((long)count + 1023) / 1024;
	const long lo = (long)tid * per < count ? (long)tid * per : count, hi = lo + per < count ? lo + per : count;
	int s = 0, mx = 0;
	for (long i = lo; i < hi; ++i) { const int c = counts[i]; s += c; mx = c > mx ? c : mx; }
	s_sum[tid] = s; s_max[tid] = mx;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) {
		const int v = tid >= off ? s_sum[tid - off] : 0;
		const int w = tid >= off ? s_max[tid - off] : 0;
		__syncthreads();
		s_sum[tid] += v; s_max[tid] = w > s_max[tid] ? w : s_max[tid];
		__syncthreads();
	}
	int run = s_sum[tid] - s;      // exclusive prefix of this thread's chunk
	for (long i = lo; i < hi; ++i) { ptr[i] = run; run += counts[i]; }
	if (tid == 1023) { ptr[count] = s_sum[1023]; maxlen[0] = s_max[1023]; }
}

// item p goes to some free place of segment key[p]: out[ptr[key[p]] + (a ticket)] = p.  fill: zeroed counters, one per segment
__global__ __launch_bounds__(256) void k_sp_scatter(const int* __restrict__ key, long count, const int* __restrict__ ptr, int* __restrict__ fill, int* __restrict__ out) {
	const long p = (long)blockIdx.x * 256 + threadIdx.x;
	if (p >= count) return;
	const int s = key[p];
	out[ptr[s] + atomicAdd(fill + s, 1)] = (int)p;
}

// Every segment [ptr[s], ptr[s + 1]) of `items` sorted ascending -- by the item itself (minor == nullptr) or by (minor[item], item).  One workgroup per segment,
// bitonic network in LDS over the next power of two (padding keys compare greatest); dynamic LDS: 8 bytes * pow2(longest segment).
__global__ __launch_bounds__(256) void k_sp_segment_sort(const int* __restrict__ ptr, int* __restrict__ items, const int* __restrict__ minor) {
	extern __shared__ __attribute__((aligned(16))) unsigned long long s_key[];
	const int s = blockIdx.x, tid = threadIdx.x;
	const int lo = ptr[s], len = ptr[s + 1] - lo;
	if (len <= 1) return;
	int P = 2;
	while (P < len) P <<= 1;
	for (int t = tid; t < P; t += 256) {
		unsigned long long k = ~0ull;
		if (t < len) { const unsigned it = (unsigned)items[lo + t]; k = ((unsigned long long)(minor != nullptr ? (unsigned)minor[it] : 0u) << 32) | it; }
		s_key[t] = k;
	}
	__syncthreads();
	for (int k = 2; k <= P; k <<= 1)
		for (int j = k >> 1; j > 0; j >>= 1) {
			for (int t = tid; t < (P >> 1); t += 256) {
				const int i = ((t / j) * 2 * j) + (t % j), l = i + j;
				const bool asc = (i & k) == 0;
				const unsigned long long x = s_key[i], y = s_key[l];
				if ((x > y) == asc) { s_key[i] = y; s_key[l] = x; }
			}
			__syncthreads();
		}
	for (int t = tid; t < len; t += 256) items[lo + t] = (int)(unsigned)(s_key[t] & 0xffffffffull);
}

// the CSR image from the order of the entries: position q holds entry p = order[q] (order == nullptr: the input itself is in CSR order)
template <typename T>
__global__ __launch_bounds__(256) void k_sp_gather_csr(const int* __restrict__ order, const int* __restrict__ row, const int* __restrict__ col, const T* __restrict__ val, long nnz,
                                                       int* __restrict__ csr_idx, T* __restrict__ csr_val, int* __restrict__ rowq) {
	const long q = (long)blockIdx.x * 256 + threadIdx.x;
	if (q >= nnz) return;
	const long p = order != nullptr ? order[q] : q;
	csr_idx[q] = col[p]; csr_val[q] = val[p]; rowq[q] = row[p];
}

// the CSC image: position k holds CSR entry q = cq[k]
template <typename T>
__global__ __launch_bounds__(256) void k_sp_gather_csc(const int* __restrict__ cq, const int* __restrict__ rowq, const T* __restrict__ csr_val, long nnz,
                                                       int* __restrict__ csc_idx, T* __restrict__ csc_val) {
	const long k = (long)blockIdx.x * 256 + threadIdx.x;
	if (k >= nnz) return;
	const int q = cq[k];
	csc_idx[k] = rowq[q]; csc_val[k] = csr_val[q];
}


// vtv[j] = sum of squares of column j in the image's order, accumulated in T with a separate multiply and add (the host loop, Engine::upload_triplets; and
// kernel::traceMultiplication's accumulation type); colsum[j] = the column's sum in double
template <typename T>
__global__ __launch_bounds__(256) void k_sp_col_sumsq(const int* __restrict__ csc_ptr, const T* __restrict__ csc_val, int n, T* __restrict__ vtv, double* __restrict__ colsum) {
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	T s = 0;
	double d = 0;
	{
		// (no contraction into a fused multiply-add: HIP's __fmul_rn / __fadd_rn are plain operators that the compiler fuses again -- the first version of this
		//  kernel differed from the host loop in the last bit of some terms)
#pragma clang fp contract(off)
		for (int p = csc_ptr[j]; p < csc_ptr[j + 1]; ++p) { const T v = csc_val[p]; const T sq = v * v; s = s + sq; d += (double)v; }
	}
	vtv[j] = s; colsum[j] = d;
}

// bp[i][b] = first position of row i whose index is >= b * per (b < blocks), bp[i][blocks] = ptr[i + 1]: a row's entries of block b of the gathered index
// are one range (its indices ascend)
__global__ __launch_bounds__(256) void k_sp_boundaries(const int* __restrict__ ptr, const int* __restrict__ idx, int rows, long per, int blocks, int* __restrict__ bp) {
	const long t = (long)blockIdx.x * 256 + threadIdx.x;
	if (t >= (long)rows * (blocks + 1)) return;
	const int i = (int)(t / (blocks + 1)), b = (int)(t % (blocks + 1));
	int lo = ptr[i], hi = ptr[i + 1];
	if (b == blocks) { bp[t] = hi; return; }
	const long lim = (long)b * per;
	while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long)idx[mid] < lim) lo = mid + 1; else hi = mid; }
	bp[t] = lo;
}

static unsigned blocks_for(long count) { return (unsigned)std::max<long>(1, (count + 255) / 256); }

hipError_t launch_sp_expand(int format, const int* a, const int* b, long nnz, int outer, int base, int m, int n, int* row, int* col, int* flags, hipStream_t stream) {
	hipLaunchKernelGGL(k_sp_expand, dim3(blocks_for(std::max<long>(nnz, (long)outer + 1))), dim3(256), 0, stream, format, a, b, nnz, outer, base, m, n, row, col, flags);
	if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
	if (nnz > 1) hipLaunchKernelGGL(k_sp_order_check, dim3(blocks_for(nnz - 1)), dim3(256), 0, stream, row, col, nnz, flags);
	return hipGetLastError();
}

hipError_t launch_sp_histogram_scan(const int* key, long count, int segments, int* counts, int* ptr, int* maxlen, hipStream_t stream) {
	if (hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)segments, stream); e != hipSuccess) return e;
	if (count > 0) hipLaunchKernelGGL(k_sp_histogram, dim3(blocks_for(count)), dim3(256), 0, stream, key, count, counts);
	hipLaunchKernelGGL(k_sp_scan, dim3(1), dim3(1024), 0, stream, counts, segments, ptr, maxlen);
	return hipGetLastError();
}

long sp_segment_sort_capacity() { return 8192; }      // keys of 8 bytes in 64 KiB of LDS

hipError_t launch_sp_scatter_sort(const int* key, long count, int segments, const int* ptr, int* fill, int* items, const int* minor, int maxlen, hipStream_t stream) {
	if (maxlen > sp_segment_sort_capacity()) return hipErrorInvalidValue;
	if (hipError_t e = hipMemsetAsync(fill, 0, sizeof(int) * (size_t)segments, stream); e != hipSuccess) return e;
	if (count > 0) hipLaunchKernelGGL(k_sp_scatter, dim3(blocks_for(count)), dim3(256), 0, stream, key, count, ptr, fill, items);
	if (maxlen > 1) {
		int P = 2;
		while (P < maxlen) P <<= 1;
		const size_t lds = sizeof(unsigned long long) * (size_t)P;
		static std::atomic<unsigned long long> lds_done{0ull};
		if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_sp_segment_sort), 65536, lds_done); e != hipSuccess) return e;
		hipLaunchKernelGGL(k_sp_segment_sort, dim3((unsigned)segments), dim3(256), lds, stream, ptr, items, minor);
	}
	return hipGetLastError();
}

template <typename T>
hipError_t launch_sp_gather_csr(const int* order, const int* row, const int* col, const T* val, long nnz, int* csr_idx, T* csr_val, int* rowq, hipStream_t stream) {
	if (nnz > 0) hipLaunchKernelGGL((k_sp_gather_csr<T>), dim3(blocks_for(nnz)), dim3(256), 0, stream, order, row, col, val, nnz, csr_idx, csr_val, rowq);
	return hipGetLastError();
}
template hipError_t launch_sp_gather_csr<float>(const int*, const int*, const int*, const float*, long, int*, float*, int*, hipStream_t);
template hipError_t launch_sp_gather_csr<double>(const int*, const int*, const int*, const double*, long, int*, double*, int*, hipStream_t);

template <typename T>
hipError_t launch_sp_gather_csc(const int* cq, const int* rowq, const T* csr_val, long nnz, int* csc_idx, T* csc_val, hipStream_t stream) {
	if (nnz > 0) hipLaunchKernelGGL((k_sp_gather_csc<T>), dim3(blocks_for(nnz)), dim3(256), 0, stream, cq, rowq, csr_val, nnz, csc_idx, csc_val);
	return hipGetLastError();
}
template hipError_t launch_sp_gather_csc<float>(const int*, const int*, const float*, long, int*, float*, hipStream_t);
template hipError_t launch_sp_gather_csc<double>(const int*, const int*, const double*, long, int*, double*, hipStream_t);

template <typename T>
hipError_t launch_sp_col_sumsq(const int* csc_ptr, const T* csc_val, int n, T* vtv, double* colsum, hipStream_t stream) {
	hipLaunchKernelGGL((k_sp_col_sumsq<T>), dim3(blocks_for(n)), dim3(256), 0, stream, csc_ptr, csc_val, n, vtv, colsum);
	return hipGetLastError();
}
template hipError_t launch_sp_col_sumsq<float>(const int*, const float*, int, float*, double*, hipStream_t);
template hipError_t launch_sp_col_sumsq<double>(const int*, const double*, int, double*, double*, hipStream_t);

hipError_t launch_sp_boundaries(const int* ptr, const int* idx, int rows, long range, int blocks, int* bp, hipStream_t stream) {
	const long per = (range + blocks - 1) / blocks;
	hipLaunchKernelGGL(k_sp_boundaries, dim3(blocks_for((long)rows * (blocks + 1))), dim3(256), 0, stream, ptr, idx, rows, per, blocks, bp);
	return hipGetLastError();
}

} // namespace nmfamd
