// kernels_tri.hip -- the factor-side passes of an iteration at padded rank 256 (BASELINE config 4: nsNMF, r = 256, bf16
// operands for the products against V): what lies between the update of a factor panel and the next product that streams it.
//
// Round 1 ran, per factor: normalise (W only; read + write of the panel), smooth (read + write), pack to bf16 fragments
// (read + half a write), Gram matrix of the smoothed panel (k_gram_wide_x3: every wave split its own operands -- VALU-bound).
// Here:
//   k_finish_panel_bf16   ONE pass: column normalisation (kernel::normalizeColumns, KernelNormalizeColumns.cu:37-58) of the rows
//                         it touches, nsNMF smoothing (AlgorithmNonSmoothNMF.h:131-134,175,194) in registers, bf16 fragments out.
//                         The smoothed fp32 panel is never written.
//   k_gram_tri_x3         G = P^T P of the UNSMOOTHED panel at fp32 accuracy on the bf16 matrix pipe (exact three-way operand split,
//                         six cross terms, as kernels_x3.hip): a workgroup owns all 36 upper-triangle 32 x 32 tiles of the 256 x 256
//                         result over its slice of panel rows, splits each 16-row K-step ONCE into LDS, and its eight waves read
//                         their tiles' fragments from there (reference: syrk, AlgorithmMultiplicativeFrobenius.h:168-178,208-209).
//   k_gram_tri_reduce     partial tiles summed in slice order, mirrored (G exactly symmetric).
//   k_smooth_gram         the Gram matrix of the smoothed panel from the unsmoothed one: with S = a I + b 1 1^T (symmetric),
//                         S G S = a^2 G + a b (g 1^T + 1 g^T) + b^2 t 1 1^T, g = G 1, t = 1^T g -- O(r^2) instead of a second
//                         pass over the panel (AlgorithmNonSmoothNMF.h:176,196 take syrk of the smoothed matrix; same value up
//                         to fp32 rounding).  The unsmoothed W^T W the error term needs (:201-202) is the by-product.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.h"
#include "split3.h"

namespace nmfamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TRI_RP = 256;            // padded rank these kernels are written for
constexpr int TRI_NB = TRI_RP / 32;    // column blocks
constexpr int TRI_TILES = TRI_NB * (TRI_NB + 1) / 2;

bool tri_kernels_available(int RP) { return RP == TRI_RP; }

// tile t of the upper triangle, enumerated row by row: (i, j), i <= j
__device__ inline void tri_tile(int t, int& i, int& j) {
	i = 0;
	int rem = t;
	while (rem >= TRI_NB - i) { rem -= TRI_NB - i; ++i; }
	j = i + rem;
}

// ---- one pass over a freshly updated panel --------------------------------------------------------------------------
// Workgroup = 4 waves = 32 panel rows; wave w owns rows 8 w .. 8 w + 7 (one half of a 16-row K-step), lane l the columns
// 4 l .. 4 l + 3: row-wise 16-byte loads (1 KB per wave and row), and the lane ends up holding, for each of its four
// columns, exactly the eight values of one bf16 fragment -- no transposition through LDS.
// colsq != nullptr: P(y, c) <- sum[c] > 0 ? P / sqrt(sum[c]) : P, written back; sum = the colsq_parts partial vectors of RP sums of squares, added in order.  Then out = offdiag * (rowsum - x) + diag * x
// for c < r (the analytic S of k_smooth_panel), rounded to nearest-even bf16, stored at the fragment slot of (ks0 + row / 16).
// (device body: eight rows starting at row0 of the launch's row range, one wave; Pin may be Pout)
__device__ __forceinline__ void finish_rows_wave(const float* Pin, float* Pout, long row0, int lane, const float* __restrict__ colsq, int colsq_parts, int r, float offdiag,
                                        float diag, bf16x8* __restrict__ dst, long ks0, long KS) {
	const float* p = Pin + row0 * TRI_RP + 4 * lane;
	float* po = Pout + row0 * TRI_RP + 4 * lane;
	float x[8][4];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const f32x4 t = *reinterpret_cast<const f32x4*>(p + (long)k * TRI_RP);
#pragma unroll
		for (int i = 0; i < 4; ++i) x[k][i] = t[i];
	}
	if (colsq != nullptr) {
		f32x4 sq = *reinterpret_cast<const f32x4*>(colsq + 4 * lane);
		for (int k = 1; k < colsq_parts; ++k) sq += *reinterpret_cast<const f32x4*>(colsq + (long)k * TRI_RP + 4 * lane);      // partial sums, in order
		// (branch-free: a column without a norm divides by one)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const float nrm = sq[i] > 0.f ? sqrtf(sq[i]) : 0.f;
			const float den = nrm > 0.f ? nrm : 1.0f;
#pragma unroll
			for (int k = 0; k < 8; ++k) x[k][i] = x[k][i] / den;
		}
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			f32x4 t;
#pragma unroll
			for (int i = 0; i < 4; ++i) t[i] = x[k][i];
			*reinterpret_cast<f32x4*>(po + (long)k * TRI_RP) = t;
		}
	}
	float s[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) s[k] = (x[k][0] + x[k][1]) + (x[k][2] + x[k][3]);      // (columns >= r hold zeros)
#pragma unroll
	for (int w = 32; w > 0; w >>= 1)
#pragma unroll
		for (int k = 0; k < 8; ++k) s[k] += __shfl_xor(s[k], w);
	const long ks = ks0 + (row0 >> 4);
	if (ks >= KS) return;
	const int h = (int)((row0 >> 3) & 1);
	bf16x8* o = dst + (ks * TRI_NB + (lane >> 3)) * 64 + h * 32 + 4 * (lane & 7);
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const float keep = 4 * lane + i < r ? 1.0f : 0.0f;
		bf16x8 f;
#pragma unroll
		for (int k = 0; k < 8; ++k) f[k] = (__bf16)(keep * (offdiag * (s[k] - x[k][i]) + diag * x[k][i]));
		o[i] = f;
	}
}

__global__ __launch_bounds__(256, 4) void k_finish_panel_bf16(float* P, const float* __restrict__ colsq, int colsq_parts, int r, float offdiag, float diag,
                                                              bf16x8* __restrict__ dst, long ks0, long KS) {
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	finish_rows_wave(P, P, (long)blockIdx.x * 32 + 8 * wave, lane, colsq, colsq_parts, r, offdiag, diag, dst, ks0, KS);
}

hipError_t launch_finish_panel_bf16(float* P, int RP, int r, long row0, long rows, const float* colsq, int colsq_parts, float offdiag, float diag, void* dst, long KS, hipStream_t stream) {
	if (RP != TRI_RP || rows <= 0 || rows % 32 != 0 || row0 % 32 != 0) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_finish_panel_bf16, dim3((unsigned)(rows / 32)), dim3(256), 0, stream, P + row0 * TRI_RP, colsq, colsq_parts, r, offdiag, diag,
	                   reinterpret_cast<bf16x8*>(dst), row0 / 16, KS);
	return hipGetLastError();
}

// The update kernel leaves one vector of RP partial sums of squares per 32 panel rows (1563 of them at config 4).  Stage 1 of their sum:
// TRI_SQ_STAGE workgroups, each adds a contiguous range of the vectors (wave q a quarter of it, 16-byte loads, eight in flight; quarters
// added in order) -> out[TRI_SQ_STAGE][RP].  The consumer (k_finish_panel_bf16) adds the staged vectors itself, in order.
constexpr int TRI_SQ_STAGE = 16;
__global__ __launch_bounds__(256) void k_colsq_stage(const float* __restrict__ part, int parts, float* __restrict__ out) {
	__shared__ f32x4 s_q[4][64];
	const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
	const int b0 = (int)(((long)parts * blockIdx.x) / TRI_SQ_STAGE), b1 = (int)(((long)parts * (blockIdx.x + 1)) / TRI_SQ_STAGE);
	const int k0 = b0 + ((b1 - b0) * q) / 4, k1 = b0 + ((b1 - b0) * (q + 1)) / 4;
	const float* p = part + 4 * l;
	f32x4 sum = {0.f, 0.f, 0.f, 0.f};
	int k = k0;
	for (; k + 8 <= k1; k += 8) {
		f32x4 v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (long)(k + u) * TRI_RP);
#pragma unroll
		for (int u = 0; u < 8; ++u) sum += v[u];
	}
	for (; k < k1; ++k) sum += *reinterpret_cast<const f32x4*>(p + (long)k * TRI_RP);
	s_q[q][l] = sum;
	__syncthreads();
	if (q == 0) *reinterpret_cast<f32x4*>(out + (long)blockIdx.x * TRI_RP + 4 * l) = ((s_q[0][l] + s_q[1][l]) + s_q[2][l]) + s_q[3][l];
}

// one vector from the staged ones (callers that need the finished sums: the row-block W step all-reduces them)
__global__ __launch_bounds__(256) void k_colsq_final(const float* __restrict__ staged, float* __restrict__ out) {
	float s = staged[threadIdx.x];
	for (int k = 1; k < TRI_SQ_STAGE; ++k) s += staged[k * TRI_RP + threadIdx.x];
	out[threadIdx.x] = s;
}

int colsq_stage_parts() { return TRI_SQ_STAGE; }

// part: `parts` vectors of RP sums -> staged[TRI_SQ_STAGE][RP]; final != nullptr: also their sum
hipError_t launch_colsq_stage(const float* part, int RP, int parts, float* staged, float* final_sum, hipStream_t stream) {
	if (RP != TRI_RP || parts <= 0) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_colsq_stage, dim3(TRI_SQ_STAGE), dim3(256), 0, stream, part, parts, staged);
	if (final_sum != nullptr) hipLaunchKernelGGL(k_colsq_final, dim3(1), dim3(TRI_RP), 0, stream, staged, final_sum);
	return hipGetLastError();
}

// ---- Gram matrix ----------------------------------------------------------------------------------------------------
// Workgroup = 8 waves (two per SIMD), slice = K-steps [s0, s1) of 16 panel rows.  Per K-step: thread (nb = wave, h, c) gathers
// P(16 s + 8 h + j, 32 nb + c), j = 0..7 (eight coalesced 4-byte loads, issued two K-steps ahead), splits them into the three bf16
// planes and stores the fragments to the LDS buffer of that K-step (2 x 24 KB); after the barrier wave w reads the fragments of its
// tiles t = w, w + 8, ... (operand layout of the 32x32x16 MFMA: both operands are "lane (c, h) holds rows 8 h .. 8 h + 7 of column c",
// so one fragment serves as A of tile (i, .) and as B of tile (., i)).
// partial[(part * 36 + t) * 1024 + g * 64 + lane] = accumulator register g of tile t.
__device__ __forceinline__ void gram_tri_block(const float* __restrict__ P, int len, int parts, int part, float* __restrict__ partial, bf16x8* __restrict__ buf) {
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int steps_total = (len + 15) / 16;
	const int s0 = (int)(((long)steps_total * part) / parts);
	const int s1 = (int)(((long)steps_total * (part + 1)) / parts);
	const int steps = s1 - s0;
	constexpr int TPW = (TRI_TILES + 7) / 8;       // tiles per wave (the last one only for waves < TRI_TILES % 8)
	int ti[TPW], tj[TPW];
#pragma unroll
	for (int q = 0; q < TPW; ++q) {
		const int t = wave + 8 * q;
		tri_tile(t < TRI_TILES ? t : 0, ti[q], tj[q]);
	}
	f32x16 acc[TPW];
#pragma unroll
	for (int q = 0; q < TPW; ++q)
#pragma unroll
		for (int g = 0; g < 16; ++g) acc[q][g] = 0.f;

	if (steps > 0) {
		const float* src = P + ((long)16 * s0 + 8 * (lane >> 5)) * TRI_RP + 32 * wave + (lane & 31);
		// register ring of GD K-steps: a gather is issued GD - 1 K-steps before its values are split (one K-step of MFMAs is
		// ~0.8 us, a dependent global round trip under load more)
#ifndef TRI_GD
#define TRI_GD 4
#endif
		constexpr int GD = TRI_GD;
		float v[GD][8];
		auto gather = [&](int s, float (&dst)[8]) {
			s = s < steps ? s : steps - 1;          // past the slice: harmless re-load of its last K-step
#pragma unroll
			for (int j = 0; j < 8; ++j) dst[j] = src[((long)16 * s + j) * TRI_RP];
		};
		auto publish = [&](int b, const float (&val)[8]) {
			bf16x8 hi, mid, lo;
			split3(val, hi, mid, lo);
			bf16x8* o = buf + b * (TRI_NB * 192) + wave * 192 + lane;
			o[0] = hi; o[64] = mid; o[128] = lo;
		};
		auto tiles = [&](int b) {
			const bf16x8* f = buf + b * (TRI_NB * 192);
#pragma unroll
			for (int q = 0; q < TPW; ++q) {
				if (q == TPW - 1 && wave + 8 * q >= TRI_TILES) break;
				const bf16x8* fa = f + ti[q] * 192 + lane;
				const bf16x8* fb = f + tj[q] * 192 + lane;
				const bf16x8 a0 = fa[0], a1 = fa[64], a2 = fa[128];
				const bf16x8 b0 = fb[0], b1 = fb[64], b2 = fb[128];
				// smallest terms first, as in k_factor_product_x3
				acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[q], 0, 0, 0);
				acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[q], 0, 0, 0);
				acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[q], 0, 0, 0);
				acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[q], 0, 0, 0);
				acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[q], 0, 0, 0);
				acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[q], 0, 0, 0);
			}
		};
#pragma unroll
		for (int d = 0; d < GD; ++d) gather(d, v[d]);
		publish(0, v[0]);
		gather(GD, v[0]);
		// K-step s: buffer s & 1; the values of K-step s + 1 sit in v[(s + 1) % GD]
		for (int s = 0; s < steps; s += GD) {
#pragma unroll
			for (int d = 0; d < GD; ++d) {
				if (s + d < steps) {
					__syncthreads();
					if (s + d + 1 < steps) publish((d + 1) & 1, v[(d + 1) % GD]);
					gather(s + d + 1 + GD, v[(d + 1) % GD]);
					tiles(d & 1);
				}
			}
		}
	}
	float* out = partial + (long)part * TRI_TILES * 1024;
#pragma unroll
	for (int q = 0; q < TPW; ++q) {
		const int t = wave + 8 * q;
		if (t < TRI_TILES) {
#pragma unroll
			for (int g = 0; g < 16; ++g) out[(long)t * 1024 + g * 64 + lane] = acc[q][g];
		}
	}
}

__global__ __launch_bounds__(512, 1) void k_gram_tri_x3(const float* __restrict__ P, int len, int parts, float* __restrict__ partial) {
	__shared__ bf16x8 buf[2][TRI_NB * 3 * 64];
	gram_tri_block(P, len, parts, blockIdx.x, partial, &buf[0][0]);
}

// One launch for the two consumers of a freshly updated panel: workgroups [0, parts) take the Gram slices of Pin (the long pole: first
// in the grid), the others 64 rows each of the finishing pass Pin -> Pout (+ bf16 fragments).  Pin != Pout when the pass normalises
// (the Gram workgroups must see ONE version of the panel: the unnormalised one; k_gram_tri_reduce applies the column scales).
__global__ __launch_bounds__(512, 2) void k_finish_and_gram(const float* Pin, float* Pout, int len, int parts, float* __restrict__ partial,
                                                            const float* __restrict__ colsq, int colsq_parts, int r, float offdiag, float diag,
                                                            bf16x8* __restrict__ dst, long KS) {
	__shared__ bf16x8 buf[2][TRI_NB * 3 * 64];
	if ((int)blockIdx.x < parts) { gram_tri_block(Pin, len, parts, blockIdx.x, partial, &buf[0][0]); return; }
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	finish_rows_wave(Pin, Pout, (long)(blockIdx.x - parts) * 64 + 8 * wave, lane, colsq, colsq_parts, r, offdiag, diag, dst, 0, KS);
}

// One thread per accumulator element: partials added in slice order.  C/D map of the 32 x 32 MFMA: register g of lane l is row
// (g & 3) + 8 (g >> 2) + 4 (l >> 5) (index within block i), column l & 31 (within block j).  Diagonal tiles: (r, c) and (c, r) add
// the six terms in different orders, so the upper triangle is kept and mirrored.
// colsq != nullptr: the slices were taken from the panel BEFORE its column normalisation x / sqrt(sum) (sum = the staged vectors, added in order;
// columns without a norm keep their values): G(r, c) is divided by the two norms.
__global__ __launch_bounds__(256) void k_gram_tri_reduce(const float* __restrict__ partial, int parts, float* __restrict__ G, const float* __restrict__ colsq, int colsq_parts,
                                                         bool as_factor) {
	// 64 accumulator elements per workgroup; wave q adds the q-th quarter of the slices, the quarters are added in order
	__shared__ float s_q[4][64];
	const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
	const int e = blockIdx.x * 64 + l;                 // < 36 * 1024
	const float* p = partial + e;
	const long stride = (long)TRI_TILES * 1024;
	const int k0 = (parts * q) / 4, k1 = (parts * (q + 1)) / 4;
	float sum = 0.f;
	int k = k0;
	for (; k + 16 <= k1; k += 16) {
		float v[16];
#pragma unroll
		for (int u = 0; u < 16; ++u) v[u] = p[(long)(k + u) * stride];
#pragma unroll
		for (int u = 0; u < 16; ++u) sum += v[u];
	}
	for (; k < k1; ++k) sum += p[(long)k * stride];
	s_q[q][l] = sum;
	__syncthreads();
	if (q != 0) return;
	sum = ((s_q[0][l] + s_q[1][l]) + s_q[2][l]) + s_q[3][l];
	const int t = e >> 10, g = (e >> 6) & 15;
	int i, j;
	tri_tile(t, i, j);
	const int r = 32 * i + (g & 3) + 8 * (g >> 2) + 4 * (l >> 5), c = 32 * j + (l & 31);
	if (colsq != nullptr && as_factor) {
		// the panel's pending column scale (PanelTriExtras), the same factors every other consumer forms (r <= c on the kept triangle: one order for both mirrors)
		sum = (sum * tri_pending_scale(colsq, colsq_parts, TRI_RP, r)) * tri_pending_scale(colsq, colsq_parts, TRI_RP, c);
	} else if (colsq != nullptr) {
		float sr = colsq[r], sc = colsq[c];
		for (int k = 1; k < colsq_parts; ++k) { sr += colsq[(long)k * TRI_RP + r]; sc += colsq[(long)k * TRI_RP + c]; }
		const float nr = sr > 0.f ? sqrtf(sr) : 1.0f, nc = sc > 0.f ? sqrtf(sc) : 1.0f;
		sum = (sum / nr) / nc;
	}
	if (i != j || r <= c) {
		G[(long)r * TRI_RP + c] = sum;
		if (r != c) G[(long)c * TRI_RP + r] = sum;
	}
}

hipError_t launch_gram_tri(const float* P, int RP, int len, int max_parts, float* partial, float* G, int num_cus, hipStream_t stream) {
	if (RP != TRI_RP || len <= 0) return hipErrorInvalidValue;
	const int steps_total = (len + 15) / 16;
	// one workgroup per CU; at least four K-steps per slice
	const int parts = std::max(1, std::min(std::min(num_cus, max_parts), steps_total / 4));
	hipLaunchKernelGGL(k_gram_tri_x3, dim3(parts), dim3(512), 0, stream, P, len, parts, partial);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k_gram_tri_reduce, dim3(TRI_TILES * 16), dim3(256), 0, stream, partial, parts, G, (const float*)nullptr, 0, false);
	return hipGetLastError();
}

// ---- Gram matrix from the bf16 fragments ---------------------------------------------------------------------------------------------
// The fragments are already MFMA operands ("lane (c, h) holds rows 8 h .. 8 h + 7 of column c" of a 16-row K-step, block nb at
// [(ks * 8 + nb) * 64 + lane]): a workgroup of eight waves streams its slice of K-steps, thread t loads fragment t of the K-step (8 KB per K-step,
// one 16-byte load per thread, GB K-steps in flight), parks it in a two-slot LDS ring, and wave w multiplies the blocks of its tiles w, w + 8, ...
// -- ONE MFMA per tile and K-step (the x3 form above: six, plus the operand split).  Same partial layout as k_gram_tri_x3.
__global__ __launch_bounds__(512, 1) void k_gram_tri_bf16(const bf16x8* __restrict__ frags, int steps_total, int parts, float* __restrict__ partial) {
	__shared__ bf16x8 buf[2][TRI_NB * 64];
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int part = blockIdx.x;
	const int s0 = (int)(((long)steps_total * part) / parts);
	const int s1 = (int)(((long)steps_total * (part + 1)) / parts);
	const int steps = s1 - s0;
	constexpr int TPW = (TRI_TILES + 7) / 8;
	int ti[TPW], tj[TPW];
#pragma unroll
	for (int q = 0; q < TPW; ++q) {
		const int t = wave + 8 * q;
		tri_tile(t < TRI_TILES ? t : 0, ti[q], tj[q]);
	}
	f32x16 acc[TPW];
#pragma unroll
	for (int q = 0; q < TPW; ++q)
#pragma unroll
		for (int g = 0; g < 16; ++g) acc[q][g] = 0.f;
	if (steps > 0) {
		constexpr int GB = 8;
		const bf16x8* src = frags + (long)s0 * (TRI_NB * 64) + tid;
		bf16x8 v[GB];
		auto fetch = [&](int s) { s = s < steps ? s : steps - 1; return src[(long)s * (TRI_NB * 64)]; };      // past the slice: harmless re-load
#pragma unroll
		for (int d = 0; d < GB; ++d) v[d] = fetch(d);
		buf[0][tid] = v[0];
		v[0] = fetch(GB);
		for (int s = 0; s < steps; s += GB) {
#pragma unroll
			for (int d = 0; d < GB; ++d) {
				if (s + d < steps) {
					__syncthreads();                       // K-step s + d is in slot d & 1; the other slot has been read by everybody
					if (s + d + 1 < steps) buf[(d + 1) & 1][tid] = v[(d + 1) % GB];
					v[(d + 1) % GB] = fetch(s + d + 1 + GB);
					const bf16x8* f = buf[d & 1];
#pragma unroll
					for (int q = 0; q < TPW; ++q) {
						if (q == TPW - 1 && wave + 8 * q >= TRI_TILES) break;
						acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ti[q] * 64 + lane], f[tj[q] * 64 + lane], acc[q], 0, 0, 0);
					}
				}
			}
		}
	}
	float* out = partial + (long)part * TRI_TILES * 1024;
#pragma unroll
	for (int q = 0; q < TPW; ++q) {
		const int t = wave + 8 * q;
		if (t < TRI_TILES) {
#pragma unroll
			for (int g = 0; g < 16; ++g) out[(long)t * 1024 + g * 64 + lane] = acc[q][g];
		}
	}
}

hipError_t launch_gram_tri_bf16(const void* frags, int RP, long KS, int max_parts, float* partial, float* G, const float* colsq, int colsq_parts, int num_cus, hipStream_t stream) {
	if (RP != TRI_RP || KS <= 0) return hipErrorInvalidValue;
	const int parts = (int)std::max<long>(1, std::min<long>(std::min(num_cus, max_parts), KS / 4));
	hipLaunchKernelGGL(k_gram_tri_bf16, dim3(parts), dim3(512), 0, stream, reinterpret_cast<const bf16x8*>(frags), (int)KS, parts, partial);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k_gram_tri_reduce, dim3(TRI_TILES * 16), dim3(256), 0, stream, partial, parts, G, colsq, colsq_parts, true);
	return hipGetLastError();
}

// Reduction of the partial tiles that also writes the split image (three bf16 planes in fragment order, store_split3) and the diagonal.  Workgroup = 1 024
// threads = 16 waves: wave (pq, gq) adds quarter pq of the slices for accumulator register g = 4 q + gq of tile t (blockIdx = 4 t + q); the four registers of
// a q are rows 8 q .. 8 q + 7 of the tile's row block (gq + 4 (lane >> 5)), exactly the eight k of one image fragment.  Direct fragments A(c, k) = G(k, c) for
// (k in row block i, c in column block j); for i != j also the mirrored ones A(r, k') = G(r, k') for k' in block j.  Diagonal tiles write the MFMA's own
// values on both sides of the diagonal (the fp32 matrix keeps the upper triangle and mirrors it).
__global__ __launch_bounds__(1024) void k_gram_tri_reduce_image(const float* __restrict__ partial, int parts, float* __restrict__ G, bf16x8* __restrict__ x3, float* __restrict__ diag) {
	__shared__ float s_p[4][4][64];
	__shared__ float s_v[4][64];
	const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, pq = w >> 2, gq = w & 3;
	const int t = blockIdx.x >> 2, q = blockIdx.x & 3, g = 4 * q + gq;
	const float* p = partial + (long)t * 1024 + g * 64 + l;
	const long stride = (long)TRI_TILES * 1024;
	const int k0 = (parts * pq) / 4, k1 = (parts * (pq + 1)) / 4;
	float sum = 0.f;
	int k = k0;
	for (; k + 16 <= k1; k += 16) {
		float v[16];
#pragma unroll
		for (int u = 0; u < 16; ++u) v[u] = p[(long)(k + u) * stride];
#pragma unroll
		for (int u = 0; u < 16; ++u) sum += v[u];
	}
	for (; k < k1; ++k) sum += p[(long)k * stride];
	s_p[pq][gq][l] = sum;
	__syncthreads();
	int i, j;
	tri_tile(t, i, j);
	if (pq == 0) {
		sum = ((s_p[0][gq][l] + s_p[1][gq][l]) + s_p[2][gq][l]) + s_p[3][gq][l];
		const int r = 32 * i + gq + 8 * q + 4 * (l >> 5), c = 32 * j + (l & 31);
		if (i != j || r <= c) {
			G[(long)r * TRI_RP + c] = sum;
			if (r != c) G[(long)c * TRI_RP + r] = sum;
			else if (diag != nullptr) diag[r] = sum;
		}
		s_v[gq][l] = sum;
	}
	__syncthreads();
	if (x3 == nullptr) return;
	if (tid < 32) {
		float v8[8];
#pragma unroll
		for (int kk = 0; kk < 8; ++kk) v8[kk] = s_v[kk & 3][(kk >> 2) * 32 + tid];
		store_split3(x3, (32 * i + 8 * q) >> 4, TRI_NB, j, q & 1, tid, v8);
	} else if (tid < 64 && i != j) {
		const int rr = (tid - 32) >> 2, mm = (tid - 32) & 3;
		float v8[8];
#pragma unroll
		for (int kk = 0; kk < 8; ++kk) v8[kk] = s_v[rr & 3][(rr >> 2) * 32 + 8 * mm + kk];
		store_split3(x3, (32 * j + 8 * mm) >> 4, TRI_NB, i, mm & 1, 8 * q + rr, v8);
	}
}

hipError_t launch_gram_tri_bf16_image(const void* frags, int RP, long KS, int max_parts, float* partial, float* G, void* x3_out, float* diag_out, int num_cus, hipStream_t stream) {
	if (RP != TRI_RP || KS <= 0) return hipErrorInvalidValue;
	// (slices: one per CU.  128 / 64 slices measured at config 4: Gram 17.6 / 23.2 us + reduction 6.4 / 5.1 against 14.8 + 9.6 -- no gain)
	const int parts = (int)std::max<long>(1, std::min<long>(std::min(num_cus, max_parts), KS / 4));
	hipLaunchKernelGGL(k_gram_tri_bf16, dim3(parts), dim3(512), 0, stream, reinterpret_cast<const bf16x8*>(frags), (int)KS, parts, partial);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k_gram_tri_reduce_image, dim3(TRI_TILES * 4), dim3(1024), 0, stream, partial, parts, G, reinterpret_cast<bf16x8*>(x3_out), diag_out);
	return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_scale_panel_tri(float* __restrict__ P, long rows, const float* __restrict__ colsq, int colsq_parts, float* __restrict__ scale_out) {
	const int lane = threadIdx.x & 63;
	const long y = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (y >= rows) return;
	f32x4 d;
#pragma unroll
	for (int j = 0; j < 4; ++j) d[j] = tri_pending_scale(colsq, colsq_parts, TRI_RP, 4 * lane + j);
	f32x4* p = reinterpret_cast<f32x4*>(P + y * TRI_RP) + lane;
	*p = *p * d;
	if (scale_out != nullptr && y == 0) *reinterpret_cast<f32x4*>(scale_out + 4 * lane) = d;
}

hipError_t launch_scale_panel_tri(float* P, int RP, long rows, const float* colsq, int colsq_parts, float* scale_out, hipStream_t stream) {
	if (RP != TRI_RP || rows <= 0 || colsq == nullptr) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_scale_panel_tri, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, P, rows, colsq, colsq_parts, scale_out);
	return hipGetLastError();
}

// launch_finish_panel_bf16 over ALL rows of the panel (rows: a multiple of 64) and launch_gram_tri in one launch + the reduction.
// colsq != nullptr: P_out <- P_in normalised (P_out != P_in), G = the Gram matrix of the NORMALISED panel; else P_out is not written (pass P_in).
hipError_t launch_finish_and_gram(const float* P_in, float* P_out, int RP, int r, long rows, int len, const float* colsq, int colsq_parts, float offdiag, float diag,
                                  void* dst, long KS, int max_parts, float* partial, float* G, int num_cus, hipStream_t stream) {
	if (RP != TRI_RP || rows <= 0 || rows % 64 != 0 || len <= 0 || (colsq != nullptr && P_in == P_out)) return hipErrorInvalidValue;
	const int steps_total = (len + 15) / 16;
	const int parts = std::max(1, std::min(std::min(num_cus, max_parts), steps_total / 4));
	hipLaunchKernelGGL(k_finish_and_gram, dim3((unsigned)(parts + rows / 64)), dim3(512), 0, stream, P_in, P_out, len, parts, partial, colsq, colsq_parts, r, offdiag, diag,
	                   reinterpret_cast<bf16x8*>(dst), KS);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k_gram_tri_reduce, dim3(TRI_TILES * 16), dim3(256), 0, stream, partial, parts, G, colsq, colsq_parts, false);
	return hipGetLastError();
}

long gram_tri_partial_elems(int max_parts) { return (long)max_parts * TRI_TILES * 1024; }

// Gs = S G S for S = a I + b 1 1^T restricted to the first r rows / columns (a = diag - offdiag, b = offdiag of k_smooth_panel).
// Every workgroup forms all row sums (G is symmetric: thread i adds column i, coalesced) and writes four rows.
__global__ __launch_bounds__(1024) void k_smooth_gram(const float* __restrict__ G, float* __restrict__ Gs, int r, float a, float b, bf16x8* __restrict__ x3) {
	__shared__ float s_p[4][TRI_RP];
	__shared__ float s_g[TRI_RP];
	__shared__ float s_w[4];
	const int tid = threadIdx.x, c = tid & 255, q = tid >> 8;
	// thread (q, c): rows 64 q .. 64 q + 63 of column c (G is symmetric: a column sum is a row sum), all loads of a batch in flight together
	float g = 0.f;
	if (c < r) {
		const int j1 = 64 * q + 64 < r ? 64 * q + 64 : r;
		int j = 64 * q;
		for (; j + 16 <= j1; j += 16) {
			float v[16];
#pragma unroll
			for (int u = 0; u < 16; ++u) v[u] = G[(long)(j + u) * TRI_RP + c];
#pragma unroll
			for (int u = 0; u < 16; ++u) g += v[u];
		}
		for (; j < j1; ++j) g += G[(long)j * TRI_RP + c];
	}
	s_p[q][c] = g;
	__syncthreads();
	if (tid < 256) {
		g = ((s_p[0][tid] + s_p[1][tid]) + s_p[2][tid]) + s_p[3][tid];
		s_g[tid] = g;
		float t = g;
		for (int w = 32; w > 0; w >>= 1) t += __shfl_xor(t, w);
		if ((tid & 63) == 0) s_w[tid >> 6] = t;
	}
	__syncthreads();
	if (tid >= 256) return;
	const float t = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
	const float a2 = a * a, ab = a * b, b2t = (b * b) * t;
	// rows 8 blockIdx .. + 7 of column c: exactly one fragment (K-step blockIdx / 2, half blockIdx & 1) of the split image the update
	// kernels take as their r x r operand (k_pack_panel_x3 of Gs: A(c, k) = Gs(k, c))
	float v[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) v[k] = G[(long)(blockIdx.x * 8 + k) * TRI_RP + c];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const int i = blockIdx.x * 8 + k;
		v[k] = (i < r && c < r) ? (a2 * v[k] + ab * (s_g[i] + s_g[c])) + b2t : 0.f;
		Gs[(long)i * TRI_RP + c] = v[k];
	}
	if (x3 != nullptr) store_split3(x3, blockIdx.x >> 1, TRI_NB, c >> 5, blockIdx.x & 1, c & 31, v);
}

hipError_t launch_smooth_gram(const float* G, float* Gs, int RP, int r, float offdiag, float diag, void* x3_out, hipStream_t stream) {
	if (RP != TRI_RP) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_smooth_gram, dim3(TRI_RP / 8), dim3(1024), 0, stream, G, Gs, r, diag - offdiag, offdiag, reinterpret_cast<bf16x8*>(x3_out));
	return hipGetLastError();
}

} // namespace nmfamd
