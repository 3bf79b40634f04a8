// kernels_fast.hip -- fp32, padded rank 64 specialisations of the small per-iteration kernels.
//
// At BASELINE config 2 (r = 64) everything that is not one of the two products against V is
// r x r x (m + n) sized work (syrk/symm calls, multiplyDivide,
// normalizeColumns).  The generic kernels in kernels.hip are correct for every rank and type
// but latency-bound; these forms put the r x r products on the fp32 MFMA pipe and cut the
// serial partial-sum chains.  Same results up to fp32 summation order (the parity tests compare
// both against the fp64 oracle).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace nmfamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// Gram matrix G = P P^T of a 64-row panel on the MFMA pipe
// (reference: syrk / gemm for W^T W and H H^T, AlgorithmMultiplicativeFrobenius.h:168-178,208-209,231-232)
// ------------------------------------------------------------------------------------------
// Workgroup = 4 waves, each wave takes one contiguous slice of y and accumulates the four 32x32
// blocks of G with v_mfma_f32_32x32x2_f32.  A and B operands of the MFMA have the same lane map
// (lane l holds element (l & 31, k = l >> 5)), so one coalesced load per 32-row block feeds both.
// The four wave tiles are summed through LDS in wave order; one partial per workgroup.
__global__ __launch_bounds__(256) void k_gram64_f32(const float* __restrict__ P, int len, int parts, float* __restrict__ partial) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const int steps_total = (len + 1) / 2;
	const int nw = parts * 4, widx = blockIdx.x * 4 + wave;
	const int s0 = (int)(((long)steps_total * widx) / nw);
	const int s1 = (int)(((long)steps_total * (widx + 1)) / nw);

	f32x16 acc[2][2];
#pragma unroll
	for (int a = 0; a < 2; ++a)
#pragma unroll
		for (int b = 0; b < 2; ++b)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

	const float* p = P + (long)(2 * s0 + half) * 64 + l31;
	int st = s0;
	for (; st + 4 <= s1; st += 4) {
		float v0[4], v1[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { v0[u] = p[(long)u * 128]; v1[u] = p[(long)u * 128 + 32]; }
		p += 4 * 128;
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[u], v0[u], acc[0][0], 0, 0, 0);
			acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[u], v1[u], acc[0][1], 0, 0, 0);
			acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[u], v0[u], acc[1][0], 0, 0, 0);
			acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[u], v1[u], acc[1][1], 0, 0, 0);
		}
	}
	for (; st < s1; ++st) {
		float v0 = p[0], v1 = p[32];
		p += 128;
		acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, v0, acc[0][0], 0, 0, 0);
		acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, v1, acc[0][1], 0, 0, 0);
		acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, v0, acc[1][0], 0, 0, 0);
		acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, v1, acc[1][1], 0, 0, 0);
	}

	// LDS image: [src wave 4][tile 4][q 4][lane 64] float4 = 64 KiB; wave w then owns tile w
	f32x4* l4 = reinterpret_cast<f32x4*>(lds);
#pragma unroll
	for (int tl = 0; tl < 4; ++tl)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			f32x4 v;
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) v[gi] = acc[tl >> 1][tl & 1][4 * q + gi];
			l4[((wave * 4 + tl) * 4 + q) * 64 + lane] = v;
		}
	__syncthreads();
	const int ab = wave >> 1, bb = wave & 1;
	float* out = partial + (long)blockIdx.x * 4096;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		f32x4 s = l4[((0 * 4 + wave) * 4 + q) * 64 + lane];
#pragma unroll
		for (int src = 1; src < 4; ++src) s += l4[((src * 4 + wave) * 4 + q) * 64 + lane];
		// D(a, b): a = gi + 8 q + 4 half (row of block ab), b = l31 (column of block bb).  G is bitwise
		// symmetric (same products, same order), so D(a, b) is stored as G(row = bb*32 + b, col = ab*32 + a):
		// consecutive lanes -> consecutive addresses.
#pragma unroll
		for (int gi = 0; gi < 4; ++gi) out[(long)(ab * 32 + gi + 8 * q + 4 * half) * 64 + bb * 32 + l31] = s[gi];
	}
}

// out[e] = sum_p partial[p][e] in a fixed order: four groups of consecutive partials are summed
// by four threads with independent loads in flight, then added group 0..3.
template <typename T>
__global__ __launch_bounds__(256) void k_reduce_partials(const T* __restrict__ partial, int parts, long stride, T* __restrict__ out, long count) {
	__shared__ T red[4][64];
	const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
	const long e = (long)blockIdx.x * 64 + tx;
	const int p0 = (parts * g) / 4, p1 = (parts * (g + 1)) / 4;
	T s = 0;
	if (e < count) {
		// batches of eight requested together, the last one clamped (the duplicates are this thread's own first line again): round 3's tail loop took the last
		// (p1 - p0) % 8 partials one dependent load at a time -- seven of the forty per group at config 5's W side
		for (int p = p0; p < p1; p += 8) {
			T v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = partial[(long)(p + u < p1 ? p + u : p0) * stride + e];
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (p + u < p1) s += v[u];
		}
	}
	red[g][tx] = s;
	__syncthreads();
	if (g == 0 && e < count) out[e] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
}

template <typename T>
hipError_t launch_reduce_partials(const T* partial, int parts, long stride, T* out, long count, hipStream_t stream) {
	hipLaunchKernelGGL((k_reduce_partials<T>), dim3((unsigned)((count + 63) / 64)), dim3(256), 0, stream, partial, parts, stride, out, count);
	return hipGetLastError();
}
template hipError_t launch_reduce_partials<float>(const float*, int, long, float*, long, hipStream_t);
template hipError_t launch_reduce_partials<double>(const double*, int, long, double*, long, hipStream_t);

hipError_t launch_gram64_f32(const float* P, int len, int parts, float* partial, float* G, hipStream_t stream) {
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_gram64_f32), 65536, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL(k_gram64_f32, dim3(parts), dim3(256), 65536, stream, P, len, parts, partial);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	return launch_reduce_partials<float>(partial, parts, 4096, G, 4096, stream);
}

// ------------------------------------------------------------------------------------------
// panel update, 64-row panels: slab reduction + r x r product on the MFMA pipe + element-wise
// update + error / norm partial sums in one kernel (what k_panel_update does for every rank).
// ------------------------------------------------------------------------------------------
// One wave = 32 panel columns (y).  D(c, y) = sum_c' Q(c, c') vec(c', y) as 2 x 32 MFMAs:
//   A operand: lane (i = l & 31, k = l >> 5) holds Q(mb*32 + i, c'_k(t))
//   B operand: lane (y = l & 31, k = l >> 5) holds vec(c'_k(t), y)
// with the K order chosen as c'_k(t) = 32*cb + 8*q + 4*k + gi for t = (cb, q, gi): exactly the set
// of rows the MFMA C/D map gives lane (y, k) -- so the registers that hold the old panel values
// (or the reduced numerator) for the element-wise step ARE the B operands.
template <int MODE>
__global__ __launch_bounds__(256) void k_panel_update64_f32(
	float* __restrict__ P, const float* __restrict__ slabs, int S, long slab_stride,
	const float* __restrict__ Q, float eps, float* __restrict__ ps, int len_valid,
	float* __restrict__ sumsq_part, float* __restrict__ num_out) {
	__shared__ __attribute__((aligned(16))) float s_sq[4][32][68];
	__shared__ float s_red[4][64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const int ycol = blockIdx.x * 128 + wave * 32 + l31;
	const long ybase = (long)ycol * 64;

	// reduced numerator and old values in the C/D register layout
	f32x4 numv[2][4], oldv[2][4];
#pragma unroll
	for (int cb = 0; cb < 2; ++cb)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const long off = ybase + 32 * cb + 8 * q + 4 * half;
			numv[cb][q] = *reinterpret_cast<const f32x4*>(slabs + off);
			if (MODE == PANEL_MU) oldv[cb][q] = *reinterpret_cast<const f32x4*>(P + off);
		}
	for (int k = 1; k < S; ++k) {
		f32x4 t[2][4];
#pragma unroll
		for (int cb = 0; cb < 2; ++cb)
#pragma unroll
			for (int q = 0; q < 4; ++q) t[cb][q] = *reinterpret_cast<const f32x4*>(slabs + (long)k * slab_stride + ybase + 32 * cb + 8 * q + 4 * half);
#pragma unroll
		for (int cb = 0; cb < 2; ++cb)
#pragma unroll
			for (int q = 0; q < 4; ++q) numv[cb][q] += t[cb][q];
	}
	if (num_out) {
#pragma unroll
		for (int cb = 0; cb < 2; ++cb)
#pragma unroll
			for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(num_out + ybase + 32 * cb + 8 * q + 4 * half) = numv[cb][q];
	}

	// all 64 A operands (Q) are requested up front, next to the numerator loads: one memory latency
	// for the whole product instead of one per group of MFMAs
	float qa[2][32];
#pragma unroll
	for (int cb = 0; cb < 2; ++cb)
#pragma unroll
		for (int q = 0; q < 4; ++q)
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) {
				const int cp = 32 * cb + 8 * q + 4 * half + gi;
				qa[0][cb * 16 + q * 4 + gi] = Q[(long)cp * 64 + l31];
				qa[1][cb * 16 + q * 4 + gi] = Q[(long)cp * 64 + 32 + l31];
			}

	f32x16 acc[2];
#pragma unroll
	for (int g = 0; g < 16; ++g) { acc[0][g] = 0.f; acc[1][g] = 0.f; }
#pragma unroll
	for (int cb = 0; cb < 2; ++cb)
#pragma unroll
		for (int q = 0; q < 4; ++q)
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) {
				const float b = (MODE == PANEL_MU) ? oldv[cb][q][gi] : numv[cb][q][gi];
				acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[0][cb * 16 + q * 4 + gi], b, acc[0], 0, 0, 0);
				acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[1][cb * 16 + q * 4 + gi], b, acc[1], 0, 0, 0);
			}

	float psum = 0.f;
#pragma unroll
	for (int mb = 0; mb < 2; ++mb)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			f32x4 o;
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) {
				const float dot = acc[mb][4 * q + gi];
				if (MODE == PANEL_MU) o[gi] = oldv[mb][q][gi] * numv[mb][q][gi] / (dot + eps);
				else o[gi] = dot > 0.f ? dot : 0.f;
				psum += o[gi] * numv[mb][q][gi];
			}
			*reinterpret_cast<f32x4*>(P + ybase + 32 * mb + 8 * q + 4 * half) = o;
			if (sumsq_part) *reinterpret_cast<f32x4*>(&s_sq[wave][l31][32 * mb + 8 * q + 4 * half]) = o * o;
		}
	if (ps) {
		psum += __shfl_xor(psum, 32);
		if (half == 0 && ycol < len_valid) ps[ycol] = psum;
	}
	if (sumsq_part) {
		__syncthreads();
		float s = 0.f;
#pragma unroll 8
		for (int y = 0; y < 32; ++y) s += s_sq[wave][y][lane];
		s_red[wave][lane] = s;
		__syncthreads();
		if (wave == 0) sumsq_part[(long)blockIdx.x * 64 + lane] = ((s_red[0][lane] + s_red[1][lane]) + s_red[2][lane]) + s_red[3][lane];
	}
}

hipError_t launch_panel_update64_f32(int mode, float* P, const float* slabs, int S, long slab_stride, const float* Q, int len_pad,
                                     float eps, float* ps, int len_valid, float* sumsq_part, float* num_out, hipStream_t stream) {
	dim3 grid(len_pad / 128), block(256);
	if (mode == PANEL_MU) hipLaunchKernelGGL((k_panel_update64_f32<PANEL_MU>), grid, block, 0, stream, P, slabs, S, slab_stride, Q, eps, ps, len_valid, sumsq_part, num_out);
	else hipLaunchKernelGGL((k_panel_update64_f32<PANEL_LS>), grid, block, 0, stream, P, slabs, S, slab_stride, Q, eps, ps, len_valid, sumsq_part, num_out);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// column normalisation of W = row scaling of the Wt panel (kernel::normalizeColumns,
// KernelNormalizeColumns.cu:37-58): sum > 0 ? x / sqrt(sum) : x.  One workgroup = 128 panel
// columns; every workgroup re-derives the RP norms from the partial sums (fixed order).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_normalize_panel_v2(T* __restrict__ P, int RP, const T* __restrict__ sumsq_part, int parts, int rows) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	T* s_norm = reinterpret_cast<T*>(smem_raw);        // [RP]
	T* s_grp = s_norm + RP;                            // [4][RP]
	const int groups = 4;
	for (int w = threadIdx.x; w < groups * RP; w += 256) {
		const int c = w % RP, g = w / RP;
		const int p0 = (parts * g) / groups, p1 = (parts * (g + 1)) / groups;
		T s = 0;
		int p = p0;
		for (; p + 8 <= p1; p += 8) {
			T v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = sumsq_part[(long)(p + u) * RP + c];
#pragma unroll
			for (int u = 0; u < 8; ++u) s += v[u];
		}
		for (; p < p1; ++p) s += sumsq_part[(long)p * RP + c];
		s_grp[g * RP + c] = s;
	}
	__syncthreads();
	for (int c = threadIdx.x; c < RP; c += 256) {
		T s = ((s_grp[c] + s_grp[RP + c]) + s_grp[2 * RP + c]) + s_grp[3 * RP + c];
		s_norm[c] = s > T(0) ? (T)sqrt(s) : T(0);
	}
	__syncthreads();
	// `rows` panel rows x RP values, one 16-byte (fp32) / 32-byte (fp64) vector per thread and step, eight steps in flight.  (Round 4: the scalar form of this loop --
	// four element loads and four guarded element stores per step -- ran config 3's 51 MB panel at 3.2 TB/s.)  RP is a multiple of 64 and a thread's column quad
	// (4 e) % RP stays the same for all its steps when RP divides 1 024 (64 ... 512): its four norms then live in registers; other ranks look them up per step.
	typedef T T4 __attribute__((ext_vector_type(4)));
	const long base = (long)blockIdx.x * rows * RP;
	const int quads = rows * RP / 4;
	const int cq = (4 * (int)threadIdx.x) % RP;
	const bool fixed_quad = (1024 % RP) == 0;
	T4 nr;
#pragma unroll
	for (int k = 0; k < 4; ++k) nr[k] = s_norm[cq + k];
	for (int e0 = threadIdx.x; e0 < quads; e0 += 256 * 8) {
		T4 v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int e = e0 + u * 256;
			if (e < quads) v[u] = *reinterpret_cast<const T4*>(P + base + 4 * (long)e);
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int e = e0 + u * 256;
			if (e < quads) {
				if (!fixed_quad) {
#pragma unroll
					for (int k = 0; k < 4; ++k) nr[k] = s_norm[(4 * e) % RP + k];
				}
				T4 o;
#pragma unroll
				for (int k = 0; k < 4; ++k) o[k] = nr[k] > T(0) ? v[u][k] / nr[k] : v[u][k];
				*reinterpret_cast<T4*>(P + base + 4 * (long)e) = o;
			}
		}
	}
}

// Many partials (short workgroup tiles on a long panel): sum them into NORM_GROUPS groups first, so that
// the scaling workgroups re-derive the norms from NORM_GROUPS rows instead of `parts` rows each.
constexpr int NORM_GROUPS = 16;
template <typename T>
__global__ __launch_bounds__(256) void k_compact_partials(const T* __restrict__ partial, int parts, int RP, T* __restrict__ out) {
	__shared__ T red[4][64];
	const int tx = threadIdx.x & 63, sub = threadIdx.x >> 6;
	const int c = blockIdx.x * 64 + tx, g = blockIdx.y;
	const int g0 = (int)(((long)parts * g) / NORM_GROUPS), g1 = (int)(((long)parts * (g + 1)) / NORM_GROUPS);
	const int p0 = g0 + ((g1 - g0) * sub) / 4, p1 = g0 + ((g1 - g0) * (sub + 1)) / 4;
	T s = 0;
	int p = p0;
	for (; p + 8 <= p1; p += 8) {
		T v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) v[u] = partial[(long)(p + u) * RP + c];
#pragma unroll
		for (int u = 0; u < 8; ++u) s += v[u];
	}
	for (; p < p1; ++p) s += partial[(long)p * RP + c];
	red[sub][tx] = s;
	__syncthreads();
	if (sub == 0) out[(long)g * RP + c] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
}

// sumsq_part: parts * RP partial sums followed by NORM_GROUPS * RP elements of scratch.
template <typename T>
hipError_t launch_normalize_panel_v2(T* P, int RP, int len_pad, T* sumsq_part, int parts, hipStream_t stream) {
	const T* src = sumsq_part;
	if (parts > 8 * NORM_GROUPS) {
		T* compact = sumsq_part + (long)parts * RP;
		hipLaunchKernelGGL((k_compact_partials<T>), dim3(RP / 64, NORM_GROUPS), dim3(256), 0, stream, sumsq_part, parts, RP, compact);
		src = compact;
		parts = NORM_GROUPS;
	}
	// short panels: 32 rows per workgroup so that the pass is spread over more than a handful of CUs
	const int rows = len_pad / 128 >= 256 ? 128 : 32;
	hipLaunchKernelGGL((k_normalize_panel_v2<T>), dim3(len_pad / rows), dim3(256), 5 * RP * sizeof(T), stream, P, RP, src, parts, rows);
	return hipGetLastError();
}
template hipError_t launch_normalize_panel_v2<float>(float*, int, int, float*, int, hipStream_t);
template hipError_t launch_normalize_panel_v2<double>(double*, int, int, double*, int, hipStream_t);

} // namespace nmfamd
