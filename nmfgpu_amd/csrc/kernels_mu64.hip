// kernels_mu64.hip -- the multiplicative-update iteration at padded rank 64, fp32, as FOUR launches.
//
// Reference sequence per iteration (source/nmf/AlgorithmMultiplicativeFrobenius.h:165-248):
//   syrk W^T W, symm (W^T W) H, gemm W^T V, multiplyDivide(H), [trace kernels],
//   syrk H H^T, symm W (H H^T), gemm V H^T, multiplyDivide(W), normalizeColumns(W)
// Here:
//   K_H  factor product  slabs <- Wu^T V (split-K)          + passenger: G, scale <- partial Grams of Wu
//   U_H  k_mu64_update<false>   H <- H .* (scale .* sum slabs) ./ (G H + eps); tr terms; partial Grams of H
//   K_W  factor product  slabs <- (V H^T)^T                 + passenger: H H^T <- partial Grams of H
//   U_W  k_mu64_update<true>    Wu <- (Wu scale) .* (sum slabs) ./ ((Wu scale) H H^T + eps); partial Grams of Wu
//
// Column normalisation of W (kernel::normalizeColumns, KernelNormalizeColumns.cu:37-58) is carried
// as a 64-entry scale vector instead of a pass over W: W = Wu diag(scale), scale(c) = 1/||Wu(:,c)||
// (1 when the norm is 0, the reference's `sum > 0` guard).  Everything that consumes W applies it:
//   W^T W = diag(scale) (Wu^T Wu) diag(scale)      (Gram reduction passenger)
//   W^T V = diag(scale) (Wu^T V)                   (numerator scaling in U_H)
//   W itself                                       (old-value scaling in U_W, Engine::materialize_w)
// Up to fp32 rounding (a multiplication by 1/norm instead of a division by norm, applied after
// instead of before the products) this is the reference's arithmetic; tolerances in tests/.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "split3.h"
#include "gram_image.h"

namespace nmfamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup = 4 waves = 64 panel columns = one contiguous 16 KiB tile of the panel.
// Global traffic is fully coalesced: every array tile (slabs, old panel values, result) moves as
// 16 B per lane over consecutive lanes, up to eight slabs in flight at once and summed in slab order,
// and the row-per-lane order the MFMA wants is produced by a pass through LDS (a row-per-lane
// pattern straight from global memory is texture-addresser bound: 64 cache lines per instruction).
// MFMA pass: wave w = (ct = w & 1: which 32 columns, mb = w >> 1: which 32 rows of the result);
//   A operand: lane (i = l & 31, k = l >> 5) holds Q(mb*32 + i, c'_k(t))
//   B operand: lane (y = l & 31, k = l >> 5) holds old(c'_k(t), y),  c'_k(t) = 32*cb + 8*q + 4*k + gi
// i.e. the K order is chosen so that the B operands are the float4 groups the C/D register map
// of the MFMA gives lane (y, k): one LDS image serves the product and the element-wise step.
template <bool IS_W>
__global__ __launch_bounds__(256) void k_mu64_update(
	float* __restrict__ P, const float* __restrict__ slabs, int S, long slab_stride,
	const float* __restrict__ Q, const float* __restrict__ scale, float eps,
	float* __restrict__ ps, int len_valid, float* __restrict__ gram_partial,
	const float* __restrict__ Gprev, int compute_error, bf16x8* __restrict__ x3_out, int x3_ks) {
	__shared__ __attribute__((aligned(16))) float s_num[64][68];   // reduced numerator, later the new values
	__shared__ __attribute__((aligned(16))) float s_old[64][68];   // old values (scaled for the W update)
	__shared__ float s_ps[2][64];
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const int ct = wave & 1, mb = wave >> 1;
	const long tile = (long)blockIdx.x * 64 * 64;
	const int c4 = (4 * tid) & 63;          // this thread's four panel rows in the linear pass
	const int yl0 = tid >> 4;               // its panel column in step j is yl0 + 16 j

	// ---- linear pass: slab sum (slab order), pending scale, into LDS ---------------------------
	f32x4 nl[4], ol[4];
	{
		f32x4 t[7][4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const long e = tile + 4 * (tid + 256 * j);
			nl[j] = *reinterpret_cast<const f32x4*>(slabs + e);
			ol[j] = *reinterpret_cast<const f32x4*>(P + e);
#pragma unroll
			for (int u = 0; u < 7; ++u) {
				const int k = 1 + u < S ? 1 + u : 0;   // clamped duplicate, discarded below
				t[u][j] = *reinterpret_cast<const f32x4*>(slabs + (long)k * slab_stride + e);
			}
		}
#pragma unroll
		for (int u = 0; u < 7; ++u)
			if (1 + u < S) {
#pragma unroll
				for (int j = 0; j < 4; ++j) nl[j] += t[u][j];
			}
		for (int k0 = 8; k0 < S; k0 += 7) {      // more than eight slabs: further batches of seven
#pragma unroll
			for (int j = 0; j < 4; ++j)
#pragma unroll
				for (int u = 0; u < 7; ++u) {
					const int k = k0 + u < S ? k0 + u : 0;
					t[u][j] = *reinterpret_cast<const f32x4*>(slabs + (long)k * slab_stride + tile + 4 * (tid + 256 * j));
				}
#pragma unroll
			for (int u = 0; u < 7; ++u)
				if (k0 + u < S) {
#pragma unroll
					for (int j = 0; j < 4; ++j) nl[j] += t[u][j];
				}
		}
	}
	const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4);
	// A operands of the r x r product (Q rows of this wave's M-block), in flight during the LDS pass
	float qa[32];
#pragma unroll
	for (int cb = 0; cb < 2; ++cb)
#pragma unroll
		for (int q = 0; q < 4; ++q)
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) qa[cb * 16 + q * 4 + gi] = Q[(long)(32 * cb + 8 * q + 4 * half + gi) * 64 + 32 * mb + l31];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		// the pending column scale of W goes on the numerator W^T V (H update) or on W itself (W update)
		if (IS_W) ol[j] *= sc; else nl[j] *= sc;
		*reinterpret_cast<f32x4*>(&s_num[yl0 + 16 * j][c4]) = nl[j];
		*reinterpret_cast<f32x4*>(&s_old[yl0 + 16 * j][c4]) = ol[j];
	}
	__syncthreads();

	// ---- MFMA pass: den = Q * old, element-wise update of this wave's 32 x 32 block --------------
	const int yrow = ct * 32 + l31;
	f32x4 oldv[2][4], numv[4];
#pragma unroll
	for (int cb = 0; cb < 2; ++cb)
#pragma unroll
		for (int q = 0; q < 4; ++q) oldv[cb][q] = *reinterpret_cast<const f32x4*>(&s_old[yrow][32 * cb + 8 * q + 4 * half]);
#pragma unroll
	for (int q = 0; q < 4; ++q) numv[q] = *reinterpret_cast<const f32x4*>(&s_num[yrow][32 * mb + 8 * q + 4 * half]);
	f32x16 acc;
#pragma unroll
	for (int g = 0; g < 16; ++g) acc[g] = 0.f;
#pragma unroll
	for (int cb = 0; cb < 2; ++cb)
#pragma unroll
		for (int q = 0; q < 4; ++q)
#pragma unroll
			for (int gi = 0; gi < 4; ++gi)
				acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[cb * 16 + q * 4 + gi], oldv[cb][q][gi], acc, 0, 0, 0);

	float psum = 0.f;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		f32x4 o;
#pragma unroll
		for (int gi = 0; gi < 4; ++gi) {
			const float old = mb == 0 ? oldv[0][q][gi] : oldv[1][q][gi];
			o[gi] = old * numv[q][gi] / (acc[4 * q + gi] + eps);
			psum += o[gi] * numv[q][gi];
		}
		// each (row, 32-row block) of s_num is read and then overwritten by exactly one wave
		*reinterpret_cast<f32x4*>(&s_num[yrow][32 * mb + 8 * q + 4 * half]) = o;
	}
	if (!IS_W && compute_error) {
		psum += __shfl_xor(psum, 32);
		if (half == 0) s_ps[mb][yrow] = psum;
	}
	if (IS_W && compute_error && blockIdx.x == 0) {
		// r terms of tr(H H^T W^T W): ps(d) = sum_i (H H^T)(d, i) (W^T W)(i, d)   (AlgorithmMultiplicativeFrobenius.h:212)
		for (int d = wave * 16; d < wave * 16 + 16; ++d) {
			float v = Q[(long)lane * 64 + d] * Gprev[(long)d * 64 + lane];
			for (int w = 32; w > 0; w >>= 1) v += __shfl_xor(v, w);
			if (lane == 0) ps[d] = v;
		}
	}
	__syncthreads();
	if (!IS_W && compute_error && tid < 64) {
		// per-column terms of tr(H^T W^T V) (kernel::traceMultiplication, AlgorithmMultiplicativeFrobenius.h:194-197)
		const int ycol = blockIdx.x * 64 + tid;
		if (ycol < len_valid) ps[ycol] = s_ps[0][tid] + s_ps[1][tid];
	}

	// ---- result out (coalesced) and partial Gram of the 64 new columns --------------------------
#pragma unroll
	for (int j = 0; j < 4; ++j)
		*reinterpret_cast<f32x4*>(P + tile + 4 * (tid + 256 * j)) = *reinterpret_cast<const f32x4*>(&s_num[yl0 + 16 * j][c4]);
	// the split (3 x bf16) image of the new panel rows for the next factor product (kernels_x3.hip): this tile is
	// four K-steps of 16 rows; a thread emits two (K-step, column block, half, lane) slots of three fragments
	if (x3_out != nullptr) {
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const int slot = tid + 256 * i;
			const int r = slot & 31, h = (slot >> 5) & 1, nb = (slot >> 6) & 1, kk = slot >> 7;
			const long ks = 4l * blockIdx.x + kk;
			if (ks < x3_ks) {
				float v[8];
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const int yl = 16 * kk + 8 * h + j;
					v[j] = blockIdx.x * 64 + yl < len_valid ? s_num[yl][32 * nb + r] : 0.f;
				}
				store_split3(x3_out, ks, 2, nb, h, r, v);
			}
		}
	}
	const int ab = wave >> 1, bb = wave & 1;   // wave (ab, bb) computes one 32 x 32 block over the 64 columns
	f32x16 g;
#pragma unroll
	for (int i = 0; i < 16; ++i) g[i] = 0.f;
#pragma unroll 8
	for (int x = 0; x < 64; x += 2) {
		const float a = s_num[x + half][ab * 32 + l31];
		const float b = s_num[x + half][bb * 32 + l31];
		g = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, g, 0, 0, 0);
	}
	float* out = gram_partial + (long)blockIdx.x * 4096;
#pragma unroll
	for (int q = 0; q < 4; ++q)
#pragma unroll
		for (int gi = 0; gi < 4; ++gi) out[(long)(ab * 32 + gi + 8 * q + 4 * half) * 64 + bb * 32 + l31] = g[4 * q + gi];
}

hipError_t launch_mu64_update(int is_w, float* P, const float* slabs, int S, long slab_stride, const float* Q, const float* scale,
                              float eps, float* ps, int len_valid, int len_pad, float* gram_partial, const float* Gprev,
                              int compute_error, hipStream_t stream, void* x3_out, int x3_ks) {
	dim3 grid(len_pad / 64), block(256);
	bf16x8* xo = reinterpret_cast<bf16x8*>(x3_out);
	if (is_w) hipLaunchKernelGGL((k_mu64_update<true>), grid, block, 0, stream, P, slabs, S, slab_stride, Q, scale, eps, ps, len_valid, gram_partial, Gprev, compute_error, xo, x3_ks);
	else hipLaunchKernelGGL((k_mu64_update<false>), grid, block, 0, stream, P, slabs, S, slab_stride, Q, scale, eps, ps, len_valid, gram_partial, Gprev, compute_error, xo, x3_ks);
	return hipGetLastError();
}

// ---- the same update on 32-column tiles, Gram matrices taken elsewhere (gram_image.h) --------------------------------
// Measured on k_mu64_update at config 2 (profiles/r02_update_kernel_parts.md): of 9.4 / 10.8 us per launch the partial
// Gram costs 2.0 - 2.4, the r x r product 2.2 - 2.4, the split image 1.3 - 1.7, and 5 - 6 us are the floor of one round trip
// (launch, slabs in, LDS, panel out) -- the kernel is ONE wave of workgroups on 79 / 157 of 256 CUs, so its time is the
// serial chain of one workgroup.  Here: half the tile (twice the workgroups, half the MFMA chain per workgroup), the r x r
// product on v_mfma_f32_16x16x4_f32 with the K order chosen so that both operands are 16-byte reads (lane group q of the
// MFMA holds c' = 16 q + t in step t), no Gram.
// Workgroup = 4 waves = 32 panel columns.  Wave w owns result rows c = 16 w .. 16 w + 15 for both 16-column halves.
// U: further slabs requested together with the first one (and per later batch): 7 for the few slabs of a whole problem, 13 for the many a short
// column shard's W^T V is cut into (26 slabs at n = 625: two round trips instead of four)
#ifndef U32_Q_EARLY
#define U32_Q_EARLY 0               // (A/B switch, tools/build_variant.sh: 1 = request the finished r x r operand behind the slabs' first batch as the K-slice forms do)
#endif
// QS: K slices the r x r operand arrives in (H update; 1: the finished matrix)
// NS (H update of nsNMF, AlgorithmNonSmoothNMF.h:174-218): the smoothing matrix S = ns_a I + ns_b 1 1^T (r x r) goes AROUND the r x r product instead of over the
// panels -- the launch before this one multiplied the UNSMOOTHED W image against V and took the unsmoothed Gram matrix, so
//   num = S D (sum of the slabs)      ((W D S)^T V = S D (W^T V)),
//   den = S (D G D) (S h)             ((W D S)^T (W D S) h),
// and what leaves for the next product is the split image of S h_new (the operand of V (S H)^T, :194); the panel itself keeps h_new.  S x = ns_a x + ns_b sum(x):
// three sums over the 64 rank rows of a column (shuffles over a column's 16 lanes in the linear pass; per-wave partials through LDS for the MFMA's C/D map).
template <bool IS_W, int U = 7, int QS = 1, bool NS = false>
__global__ __launch_bounds__(256) void k_mu64_update32(
	float* __restrict__ P, const float* __restrict__ slabs, int S, long slab_stride,
	const float* __restrict__ Q, const float* __restrict__ scale, float eps,
	float* __restrict__ ps, int len_valid, const float* __restrict__ Gprev, int compute_error, bf16x8* __restrict__ x3_out, int x3_ks, PeerSlabs peers,
	float* __restrict__ colsq_part, int qsplit, float* __restrict__ q_out, float ns_a_arg, float ns_b_arg, int ns_r) {
	static_assert(!(NS && IS_W), "the smoothing matrix rides the H update only");
	typedef float f32x4v __attribute__((ext_vector_type(4)));
	__shared__ __attribute__((aligned(16))) float s_num[32][68];   // reduced numerator, later the new values
	__shared__ __attribute__((aligned(16))) float s_old[32][68];   // old values (scaled for the W update)
	__shared__ __attribute__((aligned(16))) float s_sm[NS ? 32 : 1][68];   // NS: S h, the B operand of the r x r product
	__shared__ float s_ps[4][32];
	__shared__ float s_dsum[NS ? 4 : 1][32], s_osum[NS ? 4 : 1][32];      // NS: per-wave parts of a column's sums over the rank rows (denominator, new values)
	const float ns_a = NS ? in_vgpr(ns_a_arg) : 1.f, ns_b = NS ? in_vgpr(ns_b_arg) : 0.f;   // (VGPRs: a packed multiply must not take a scalar source, split3.h)
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int q = lane >> 4, l15 = lane & 15;
	const long tile = (long)blockIdx.x * 32 * 64;
	const int c4 = (4 * tid) & 63;          // this thread's four panel rows in the linear pass
	const int yl0 = tid >> 4;               // its panel column in step j is yl0 + 16 j

	// ---- linear pass: slab sum (slab order), pending scale, into LDS ---------------------------
	// peers.count > 0: the "slabs" are the exchange panels of the ranks of a column-sharded run, read where they lie (this device or a peer's memory,
	// comm.h exchange_publish) and added in rank order -- the all-reduce of SURVEY 8(e) happens in this kernel's prologue
	if (peers.count > 0) S = peers.count;
	// panels in OTHER devices' memory: a system-scope acquire before the first read of them -- whatever fence scope the runtime gave this launch, lines of a
	// peer's buffer this device cached two iterations ago (the exchange slots alternate) must not be served again.  Never executed by a single-GPU run.
	if (peers.count > 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
	auto slab_at = [&](int k) -> const float* { return peers.count > 0 ? peers.p[k] : slabs + (long)k * slab_stride; };
	f32x4v qa[4];
	f32x4v qs[QS > 1 ? QS - 1 : 1][4];
	f32x4v nl[2], ol[2];
	{
		f32x4v t[U][2];
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const long e = tile + 4 * (tid + 256 * j);
			nl[j] = *reinterpret_cast<const f32x4v*>(slab_at(0) + e);
			ol[j] = *reinterpret_cast<const f32x4v*>(P + e);
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int k = 1 + u < S ? 1 + u : 0;   // clamped duplicate, discarded below
				t[u][j] = *reinterpret_cast<const f32x4v*>(slab_at(k) + e);
			}
		}
		if (QS > 1 || U32_Q_EARLY) {
			// Q arrives as QS unscaled K slices of Wu^T Wu (gram_image.h, K-split form): requested right BEHIND the slabs' first batch -- the wait for that batch then
			// leaves these in flight and they arrive while the slabs are summed and parked in LDS (in front of the batch they made every workgroup of the launch wait
			// for the same few lines first: 7.4 -> 10 us with four slices; behind the slab loop they cost a round trip of their own)
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int u = 0; u < 4; ++u) qa[u] = *reinterpret_cast<const f32x4v*>(Q + (long)(16 * wave + l15) * 64 + 16 * q + 4 * u);
#pragma unroll
			for (int k = 1; k < QS; ++k)
#pragma unroll
				for (int u = 0; u < 4; ++u) qs[k - 1][u] = *reinterpret_cast<const f32x4v*>(Q + (long)k * 4096 + (long)(16 * wave + l15) * 64 + 16 * q + 4 * u);
			__builtin_amdgcn_sched_barrier(0);
		}
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (1 + u < S) {
#pragma unroll
				for (int j = 0; j < 2; ++j) nl[j] += t[u][j];
			}
		for (int k0 = 1 + U; k0 < S; k0 += U) {      // more slabs than one batch: further batches of U
#pragma unroll
			for (int j = 0; j < 2; ++j)
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const int k = k0 + u < S ? k0 + u : 0;
					t[u][j] = *reinterpret_cast<const f32x4v*>(slab_at(k) + tile + 4 * (tid + 256 * j));
				}
#pragma unroll
			for (int u = 0; u < U; ++u)
				if (k0 + u < S) {
#pragma unroll
					for (int j = 0; j < 2; ++j) nl[j] += t[u][j];
				}
		}
	}
	const f32x4v sc = *reinterpret_cast<const f32x4v*>(scale + c4);
	// A operand of the r x r product: Q(c = 16 wave + l15, c' = 16 q + t), t = 0 .. 15 (Q is symmetric: a row is a column); one finished matrix: in flight during the LDS pass
	if (QS == 1 && !U32_Q_EARLY) {
#pragma unroll
		for (int u = 0; u < 4; ++u) qa[u] = *reinterpret_cast<const f32x4v*>(Q + (long)(16 * wave + l15) * 64 + 16 * q + 4 * u);
	}
	if (!IS_W && QS > 1) {
		// ... then D (.) D as the one-slice passengers do: (v * d(column)) * d(row)
#pragma unroll
		for (int k = 1; k < QS; ++k)
#pragma unroll
			for (int u = 0; u < 4; ++u) qa[u] += qs[k - 1][u];
		const float srow = scale[16 * wave + l15];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const f32x4v scol = *reinterpret_cast<const f32x4v*>(scale + 16 * q + 4 * u);
			qa[u] = (qa[u] * scol) * srow;
			if (blockIdx.x == 0 && q_out != nullptr) *reinterpret_cast<f32x4v*>(q_out + (long)(16 * wave + l15) * 64 + 16 * q + 4 * u) = qa[u];
		}
	}
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		// the pending column scale of W goes on the numerator W^T V (H update) or on W itself (W update)
		if (IS_W) ol[j] *= sc; else nl[j] *= sc;
		if (NS) {
			// S over the rank rows of column yl0 + 16 j: its 64 rows lie in the 16 lanes tid & 15 of this wave (rows past r: zero in, zero out)
			float sn = (nl[j][0] + nl[j][1]) + (nl[j][2] + nl[j][3]);
			float so = (ol[j][0] + ol[j][1]) + (ol[j][2] + ol[j][3]);
#pragma unroll
			for (int w = 1; w < 16; w <<= 1) { sn += __shfl_xor(sn, w); so += __shfl_xor(so, w); }
			f32x4v sm;
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				const bool in = c4 + g < ns_r;
				nl[j][g] = in ? ns_a * nl[j][g] + ns_b * sn : 0.f;
				sm[g] = in ? ns_a * ol[j][g] + ns_b * so : 0.f;
			}
			*reinterpret_cast<f32x4v*>(&s_sm[yl0 + 16 * j][c4]) = sm;
		}
		*reinterpret_cast<f32x4v*>(&s_num[yl0 + 16 * j][c4]) = nl[j];
		*reinterpret_cast<f32x4v*>(&s_old[yl0 + 16 * j][c4]) = ol[j];
	}
	__syncthreads();

	// ---- den = Q * old on the matrix pipe, two independent 16 x 16 accumulators per wave -------
	f32x4v ob[2][4];
#pragma unroll
	for (int yt = 0; yt < 2; ++yt)
#pragma unroll
		for (int u = 0; u < 4; ++u) ob[yt][u] = *reinterpret_cast<const f32x4v*>(NS ? &s_sm[16 * yt + l15][16 * q + 4 * u] : &s_old[16 * yt + l15][16 * q + 4 * u]);
	f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
	for (int u = 0; u < 4; ++u)
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[u][g], ob[0][u][g], acc[0], 0, 0, 0);
			acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[u][g], ob[1][u][g], acc[1], 0, 0, 0);
		}
	// C/D map: register g of lane (q, l15) is row c = 16 wave + 4 q + g, column y = 16 yt + l15
	f32x4v sq = {0.f, 0.f, 0.f, 0.f};
	if (NS) {
		// den = S t: the sum of t over all 64 rows of a column = over g, the four lane groups and the four waves
#pragma unroll
		for (int yt = 0; yt < 2; ++yt) {
			float v = (acc[yt][0] + acc[yt][1]) + (acc[yt][2] + acc[yt][3]);
			v += __shfl_xor(v, 16);
			v += __shfl_xor(v, 32);
			if (q == 0) s_dsum[wave][16 * yt + l15] = v;
		}
		__syncthreads();
#pragma unroll
		for (int yt = 0; yt < 2; ++yt) {
			const int y = 16 * yt + l15;
			const float tsum = ((s_dsum[0][y] + s_dsum[1][y]) + s_dsum[2][y]) + s_dsum[3][y];
#pragma unroll
			for (int g = 0; g < 4; ++g) acc[yt][g] = ns_a * acc[yt][g] + ns_b * tsum;     // (rows past r: old value 0, the quotient is never used)
		}
	}
#pragma unroll
	for (int yt = 0; yt < 2; ++yt) {
		const int y = 16 * yt + l15;
		const f32x4v oldv = *reinterpret_cast<const f32x4v*>(&s_old[y][16 * wave + 4 * q]);
		const f32x4v numv = *reinterpret_cast<const f32x4v*>(&s_num[y][16 * wave + 4 * q]);
		f32x4v o;
		float psum = 0.f;
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			o[g] = oldv[g] * numv[g] / (acc[yt][g] + eps);
			psum += o[g] * numv[g];
			if (IS_W) sq[g] += o[g] * o[g];
		}
		// each (column, four rows) of s_num is read and then overwritten by exactly one lane
		*reinterpret_cast<f32x4v*>(&s_num[y][16 * wave + 4 * q]) = o;
		if (NS) {
			float v = (o[0] + o[1]) + (o[2] + o[3]);
			v += __shfl_xor(v, 16);
			v += __shfl_xor(v, 32);
			if (q == 0) s_osum[wave][y] = v;
		}
		if (!IS_W && compute_error) {
			psum += __shfl_xor(psum, 16);
			psum += __shfl_xor(psum, 32);
			if (q == 0) s_ps[wave][y] = psum;
		}
	}
	if (IS_W && colsq_part != nullptr) {
		// this workgroup's 32 new rows: their sums of squares per factor column (kernel::normalizeColumns' sums, KernelNormalizeColumns.cu:37-49, in parts; rows
		// past the valid length are exactly 0): over the 16 lanes of a quarter, then one 16-byte store per quarter
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			float v = sq[g];
			v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
			sq[g] = v;
		}
		if (l15 == 0) *reinterpret_cast<f32x4v*>(colsq_part + (long)blockIdx.x * 64 + 16 * wave + 4 * q) = sq;
	}
	if (IS_W && compute_error && wave == 0) {
		// r terms of tr(H H^T W^T W): ps(d) = sum_i (H H^T)(d, i) (W^T W)(i, d)   (AlgorithmMultiplicativeFrobenius.h:212) -- one term per workgroup (round 3:
		// all 64 in workgroup 0, sixteen dependent round trips per wave: that workgroup ran 16 us on every error iteration while the others took 6)
		for (int d = blockIdx.x; d < 64; d += gridDim.x) {
			float v = Q[(long)lane * 64 + d] * Gprev[(long)d * 64 + lane];
			for (int w = 32; w > 0; w >>= 1) v += __shfl_xor(v, w);
			if (lane == 0) ps[d] = v;
		}
	}
	__syncthreads();
	if (!IS_W && compute_error && tid < 32) {
		// per-column terms of tr(H^T W^T V) (kernel::traceMultiplication, AlgorithmMultiplicativeFrobenius.h:194-197)
		const int ycol = blockIdx.x * 32 + tid;
		if (ycol < len_valid) ps[ycol] = ((s_ps[0][tid] + s_ps[1][tid]) + s_ps[2][tid]) + s_ps[3][tid];
	}
	// ---- result out (coalesced) and the split image of the two K-steps this tile is --------------
#pragma unroll
	for (int j = 0; j < 2; ++j)
		*reinterpret_cast<f32x4v*>(P + tile + 4 * (tid + 256 * j)) = *reinterpret_cast<const f32x4v*>(&s_num[yl0 + 16 * j][c4]);
	{
		const int r = tid & 31, h = (tid >> 5) & 1, nb = (tid >> 6) & 1, kk = tid >> 7;
		const long ks = 2l * blockIdx.x + kk;
		if (ks < x3_ks) {
			float v[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const int yl = 16 * kk + 8 * h + j;
				float x = s_num[yl][32 * nb + r];
				// NS: the image is that of S h_new, the operand of V (S H)^T (rows past the rank stay zero)
				if (NS) x = 32 * nb + r < ns_r ? ns_a * x + ns_b * (((s_osum[0][yl] + s_osum[1][yl]) + s_osum[2][yl]) + s_osum[3][yl]) : 0.f;
				v[j] = blockIdx.x * 32 + yl < len_valid ? x : 0.f;
			}
			store_split3(x3_out, ks, 2, nb, h, r, v);
		}
	}
}

template <bool IS_W, int U, int QS, bool NS = false>
static void launch_update32_inst(dim3 grid, hipStream_t stream, float* P, const float* slabs, int S, long slab_stride, const float* Q, const float* scale, float eps, float* ps,
                                 int len_valid, const float* Gprev, int compute_error, bf16x8* xo, int x3_ks, const PeerSlabs& peers, float* colsq_part, float* q_out,
                                 const SmoothAround* ns = nullptr) {
	hipLaunchKernelGGL((k_mu64_update32<IS_W, U, QS, NS>), grid, dim3(256), 0, stream, P, slabs, S, slab_stride, Q, scale, eps, ps, len_valid, Gprev, compute_error, xo, x3_ks, peers,
	                   colsq_part, QS, q_out, NS ? ns->diag - ns->off : 1.f, NS ? ns->off : 0.f, NS ? ns->r : 0);
}

hipError_t launch_mu64_update32(int is_w, float* P, const float* slabs, int S, long slab_stride, const float* Q, const float* scale,
                                float eps, float* ps, int len_valid, int len_pad, const float* Gprev, int compute_error, hipStream_t stream,
                                void* x3_out, int x3_ks, const PeerSlabs* peers, float* colsq_part, int qsplit, float* q_out, const SmoothAround* ns) {
	if (x3_out == nullptr || len_pad % 32 != 0 || (qsplit > 1 && is_w) || (qsplit > 1 && qsplit != 2 && qsplit != 4 && qsplit != 8)) return hipErrorInvalidValue;
	if (ns != nullptr && (is_w || ns->r < 1 || ns->r > 64)) return hipErrorInvalidValue;
	PeerSlabs pa = {};
	if (peers != nullptr) {
		if (peers->count < 1 || peers->count > PEER_SLABS_MAX) return hipErrorInvalidValue;
		pa = *peers;
		S = peers->count;
	}
	dim3 grid(len_pad / 32);
	bf16x8* xo = reinterpret_cast<bf16x8*>(x3_out);
#define NMFAMD_U32(ISW, UU, QQ) launch_update32_inst<ISW, UU, QQ>(grid, stream, P, slabs, S, slab_stride, Q, scale, eps, ps, len_valid, Gprev, compute_error, xo, x3_ks, pa, colsq_part, q_out)
#define NMFAMD_U32NS(UU, QQ) launch_update32_inst<false, UU, QQ, true>(grid, stream, P, slabs, S, slab_stride, Q, scale, eps, ps, len_valid, Gprev, compute_error, xo, x3_ks, pa, colsq_part, q_out, ns)
	if (ns != nullptr) {
		if (S > 8) { if (qsplit == 2) NMFAMD_U32NS(13, 2); else if (qsplit == 4) NMFAMD_U32NS(13, 4); else if (qsplit == 8) NMFAMD_U32NS(13, 8); else NMFAMD_U32NS(13, 1); }
		else { if (qsplit == 2) NMFAMD_U32NS(7, 2); else if (qsplit == 4) NMFAMD_U32NS(7, 4); else if (qsplit == 8) NMFAMD_U32NS(7, 8); else NMFAMD_U32NS(7, 1); }
#undef NMFAMD_U32NS
		return hipGetLastError();
	}
	if (is_w) { if (S > 8) NMFAMD_U32(true, 13, 1); else NMFAMD_U32(true, 7, 1); }
	else if (S > 8) { if (qsplit == 2) NMFAMD_U32(false, 13, 2); else if (qsplit == 4) NMFAMD_U32(false, 13, 4); else if (qsplit == 8) NMFAMD_U32(false, 13, 8); else NMFAMD_U32(false, 13, 1); }
	else { if (qsplit == 2) NMFAMD_U32(false, 7, 2); else if (qsplit == 4) NMFAMD_U32(false, 7, 4); else if (qsplit == 8) NMFAMD_U32(false, 7, 8); else NMFAMD_U32(false, 7, 1); }
#undef NMFAMD_U32
	return hipGetLastError();
}

// out[i] = sum over k (ascending) of src.p[k][i], i < count (a multiple of 4): the r x r part of the exchange (H_g H_g^T of every rank) before the W update
__global__ __launch_bounds__(256) void k_sum_peers(PeerSlabs src, float* __restrict__ out, int count) {
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // (peers' memory: see k_mu64_update32)
	const int i = 4 * (blockIdx.x * 256 + threadIdx.x);
	if (i >= count) return;
	f32x4 s = *reinterpret_cast<const f32x4*>(src.p[0] + i);
	for (int k = 1; k < src.count; ++k) s += *reinterpret_cast<const f32x4*>(src.p[k] + i);
	*reinterpret_cast<f32x4*>(out + i) = s;
}

hipError_t launch_sum_peers(const PeerSlabs& src, float* out, int count, hipStream_t stream) {
	if (src.count < 1 || src.count > PEER_SLABS_MAX || count % 4 != 0) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_sum_peers, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, stream, src, out, count);
	return hipGetLastError();
}

// Stand-alone form of the Gram-from-image passengers (callers that have no product launch to ride in)
__global__ __launch_bounds__(256) void k_gram_image(GramReduceArgs rg) {
	__shared__ __attribute__((aligned(16))) float lds[GRAM_IMAGE_LDS_FLOATS];
	gram_image_block(rg, blockIdx.x, lds);
}

// every field of rg as given (the K-split form included: 10 * ksplit blocks)
hipError_t launch_gram_image_args(const GramReduceArgs& rg, hipStream_t stream) {
	if (rg.image == nullptr || rg.ksplit > GRAM_KSPLIT_MAX || (rg.ksplit > 1 && rg.normalize != 0 && rg.colsq_part == nullptr)) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_gram_image, dim3(rg.ksplit > 1 ? GRAM_IMAGE_TILES * rg.ksplit : GRAM_REDUCE_BLOCKS), dim3(256), 0, stream, rg);
	return hipGetLastError();
}

hipError_t launch_gram_from_image(const void* image, int image_ks, float* G, float* scale, int normalize, hipStream_t stream, const float* colsq_part, int colsq_parts) {
	GramReduceArgs rg = {nullptr, 0, G, scale, normalize};
	rg.image = image; rg.image_ks = image_ks;
	rg.colsq_part = colsq_part; rg.colsq_parts = colsq_parts;
	hipLaunchKernelGGL(k_gram_image, dim3(GRAM_REDUCE_BLOCKS), dim3(256), 0, stream, rg);
	return hipGetLastError();
}

// Partial Gram matrices of an existing panel in the layout k_mu64_update produces (one per 64
// panel columns): used once after W has been (re)initialised.
__global__ __launch_bounds__(256) void k_mu64_gram_partials(const float* __restrict__ P, float* __restrict__ partial) {
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const int ab = wave >> 1, bb = wave & 1;
	const float* p = P + (long)blockIdx.x * 64 * 64;
	f32x16 g;
#pragma unroll
	for (int i = 0; i < 16; ++i) g[i] = 0.f;
#pragma unroll 8
	for (int x = 0; x < 64; x += 2) {
		const float a = p[(long)(x + half) * 64 + ab * 32 + l31];
		const float b = p[(long)(x + half) * 64 + bb * 32 + l31];
		g = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, g, 0, 0, 0);
	}
	float* out = partial + (long)blockIdx.x * 4096;
#pragma unroll
	for (int q = 0; q < 4; ++q)
#pragma unroll
		for (int gi = 0; gi < 4; ++gi) out[(long)(ab * 32 + gi + 8 * q + 4 * half) * 64 + bb * 32 + l31] = g[4 * q + gi];
}

hipError_t launch_mu64_gram_partials(const float* P, int len_pad, float* partial, hipStream_t stream) {
	hipLaunchKernelGGL(k_mu64_gram_partials, dim3(len_pad / 64), dim3(256), 0, stream, P, partial);
	return hipGetLastError();
}

// Stand-alone Gram reduction with the same arithmetic (and summation order) as the passenger
// workgroups of the factor-product launch.
__global__ __launch_bounds__(512) void k_mu64_gram_reduce(GramReduceArgs rg) {
	__shared__ float lds[64 + 512];
	const int tid = threadIdx.x, blk = blockIdx.x;
	float* s_scale = lds;
	float* s_tmp = lds + 64;
	const int parts = rg.parts;
	if (rg.normalize) {
		if (tid < 128) {
			const int c = tid & 63, g = tid >> 6;
			const int p0 = (parts * g) / 2, p1 = (parts * (g + 1)) / 2;
			float sum = 0.f;
			// (eight loads in flight, added in order: one load per round trip made this launch a chain of parts / 2 dependent loads -- 20 us at 64 parts)
			for (int p = p0; p < p1; p += 8) {
				float v[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) v[u] = rg.partials[(long)(p + u < p1 ? p + u : p0) * 4096 + c * 65];
#pragma unroll
				for (int u = 0; u < 8; ++u)
					if (p + u < p1) sum += v[u];
			}
			s_tmp[g * 64 + c] = sum;
		}
		__syncthreads();
		if (tid < 64) {
			const float d = s_tmp[tid] + s_tmp[64 + tid];
			s_scale[tid] = d > 0.f ? 1.0f / sqrtf(d) : 1.0f;
		}
	} else if (tid < 64) {
		s_scale[tid] = 1.0f;
	}
	__syncthreads();
	{
		const int el = tid & 255, g = tid >> 8;
		const int e = blk * 256 + el;
		const int p0 = (parts * g) / 2, p1 = (parts * (g + 1)) / 2;
		float sum = 0.f;
		for (int p = p0; p < p1; p += 8) {
			float v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = rg.partials[(long)(p + u < p1 ? p + u : p0) * 4096 + e];
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (p + u < p1) sum += v[u];
		}
		s_tmp[g * 256 + el] = sum;
	}
	__syncthreads();
	if (tid < 256) {
		const int e = blk * 256 + tid;
		const float v = s_tmp[tid] + s_tmp[256 + tid];
		rg.G[e] = (v * s_scale[e & 63]) * s_scale[e >> 6];
	}
	if (blk == 0 && tid < 64 && rg.scale) rg.scale[tid] = s_scale[tid];
}

hipError_t launch_mu64_gram_reduce(const GramReduceArgs& rg, hipStream_t stream) {
	hipLaunchKernelGGL(k_mu64_gram_reduce, dim3(GRAM_REDUCE_BLOCKS), dim3(512), 0, stream, rg);
	return hipGetLastError();
}

// Stand-alone form of the reduction for callers that cannot hide it behind a product launch: the wide
// k_reduce_partials (64 workgroups, 4 us) sums the partial matrices; this one-workgroup kernel then turns the diagonal
// into the column scales and scales G on both sides -- the passenger form takes 31 us when nothing runs beside it.
__global__ __launch_bounds__(256) void k_scale_gram64(float* __restrict__ G, float* __restrict__ scale) {
	__shared__ float s_scale[64];
	const int tid = threadIdx.x;
	if (tid < 64) {
		const float d = G[tid * 65];
		const float sc = d > 0.f ? 1.0f / sqrtf(d) : 1.0f;
		s_scale[tid] = sc;
		scale[tid] = sc;
	}
	__syncthreads();
	for (int e = tid; e < 4096; e += 256) G[e] = (G[e] * s_scale[e & 63]) * s_scale[e >> 6];
}

hipError_t launch_gram64_from_partials(const float* partials, int parts, float* G, float* scale, hipStream_t stream) {
	hipError_t e = launch_reduce_partials<float>(partials, parts, 4096, G, 4096, stream);
	if (e != hipSuccess || scale == nullptr) return e;
	hipLaunchKernelGGL(k_scale_gram64, dim3(1), dim3(256), 0, stream, G, scale);
	return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_mu64_apply_scale(float* __restrict__ P, long count4, const float* __restrict__ scale) {
	const long e = (long)blockIdx.x * 256 + threadIdx.x;
	if (e >= count4) return;
	f32x4 v = *reinterpret_cast<f32x4*>(P + 4 * e);
	const f32x4 s = *reinterpret_cast<const f32x4*>(scale + (4 * e) % 64);
	v *= s;
	*reinterpret_cast<f32x4*>(P + 4 * e) = v;
}

// The same pass also writing the split (3 x bf16) image of the scaled panel for the next factor product
// (kernels_x3.hip): a workgroup is one K-step of 16 panel rows.
__global__ __launch_bounds__(256) void k_mu64_apply_scale_x3(float* __restrict__ P, const float* __restrict__ scale, bf16x8* __restrict__ x3_out, int x3_ks) {
	__shared__ __attribute__((aligned(16))) float s_v[16][68];
	const int tid = threadIdx.x;
	const long e = (long)blockIdx.x * 256 + tid;
	f32x4 v = *reinterpret_cast<f32x4*>(P + 4 * e);
	v *= *reinterpret_cast<const f32x4*>(scale + (4 * tid) % 64);
	*reinterpret_cast<f32x4*>(P + 4 * e) = v;
	*reinterpret_cast<f32x4*>(&s_v[tid >> 4][4 * (tid & 15)]) = v;
	__syncthreads();
	if (tid < 128 && (int)blockIdx.x < x3_ks) {
		const int r = tid & 31, h = (tid >> 5) & 1, nb = tid >> 6;
		float w[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) w[j] = s_v[8 * h + j][32 * nb + r];
		store_split3(x3_out, blockIdx.x, 2, nb, h, r, w);
	}
}

// normalize_w of the rank-64 least-squares path in ONE launch after the reduction of the partial Gram matrices (k_scale_gram64 + k_mu64_apply_scale_x3:
// 4.8 + 5.4 us of launch floor): every workgroup takes the column scales from the diagonal of the reduced, still unscaled matrix itself (64 elements, L2 hits),
// scales its 16 panel rows and writes their split image; one extra workgroup publishes the scales and G = D Graw D.  Same arithmetic per element as the two kernels.
__global__ __launch_bounds__(256) void k_mu64_scale_all_x3(float* __restrict__ P, const float* __restrict__ Graw, float* __restrict__ G, float* __restrict__ scale,
                                                          bf16x8* __restrict__ x3_out, int x3_ks, int row_blocks) {
	__shared__ __attribute__((aligned(16))) float s_v[16][68];
	__shared__ __attribute__((aligned(16))) float s_scale[64];
	const int tid = threadIdx.x;
	if (tid < 64) {
		const float d = Graw[tid * 65];
		s_scale[tid] = d > 0.f ? 1.0f / sqrtf(d) : 1.0f;
	}
	__syncthreads();
	if ((int)blockIdx.x == row_blocks) {
		if (tid < 64) scale[tid] = s_scale[tid];
		for (int e = tid; e < 4096; e += 256) G[e] = (Graw[e] * s_scale[e & 63]) * s_scale[e >> 6];
		return;
	}
	const long e = (long)blockIdx.x * 256 + tid;
	f32x4 v = *reinterpret_cast<f32x4*>(P + 4 * e);
	v *= *reinterpret_cast<const f32x4*>(s_scale + (4 * tid) % 64);
	*reinterpret_cast<f32x4*>(P + 4 * e) = v;
	*reinterpret_cast<f32x4*>(&s_v[tid >> 4][4 * (tid & 15)]) = v;
	__syncthreads();
	if (tid < 128 && (int)blockIdx.x < x3_ks) {
		const int r = tid & 31, h = (tid >> 5) & 1, nb = tid >> 6;
		float w[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) w[j] = s_v[8 * h + j][32 * nb + r];
		store_split3(x3_out, blockIdx.x, 2, nb, h, r, w);
	}
}

// Round 4: the reduction of the partial Gram matrices AND the scaling pass in ONE launch (every launch costs ~5 us whatever it does).  What made round 3's attempt
// slow (13 us) was the diagonal: every workgroup gathered it from the partial matrices, one cache line per element and part.  Here the column scales come from the
// per-workgroup sums of squares the update kernel writes anyway (`sq_parts` vectors of 64, contiguous): every workgroup adds them in ONE batch of loads (four groups
// of parts, added in order: the same bits everywhere).  Grid: row_blocks workgroups scale 16 panel rows each and write their split image; 64 more reduce one row of
// the Gram matrix each over `parts` partial matrices (four groups of parts) and scale it, D (sum) D; the first of them publishes the scales.
__global__ __launch_bounds__(256) void k_mu64_reduce_scale_all_x3(float* __restrict__ P, const float* __restrict__ gram_part, int parts, const float* __restrict__ sumsq_part, int sq_parts,
                                                                 float* __restrict__ G, float* __restrict__ scale, bf16x8* __restrict__ x3_out, int x3_ks, int row_blocks) {
	__shared__ __attribute__((aligned(16))) float s_v[16][68];
	__shared__ __attribute__((aligned(16))) float s_scale[64];
	__shared__ float s_grp[4][64];
	const int tid = threadIdx.x;
	{
		// sums of squares of the 64 columns: thread (c, g) takes the parts g, g + 4, ... -- up to 48 of them in flight (config 5: 158 parts, 40 per thread)
		const int c = tid & 63, g = tid >> 6;
		float sum = 0.f;
		for (int p = g; p < sq_parts; p += 4 * 48) {
			float v[48];
#pragma unroll
			for (int u = 0; u < 48; ++u) v[u] = p + 4 * u < sq_parts ? sumsq_part[(long)(p + 4 * u) * 64 + c] : 0.f;
#pragma unroll
			for (int u = 0; u < 48; ++u) sum += v[u];
		}
		s_grp[g][c] = sum;
	}
	__syncthreads();
	if (tid < 64) {
		const float d = ((s_grp[0][tid] + s_grp[1][tid]) + s_grp[2][tid]) + s_grp[3][tid];
		s_scale[tid] = d > 0.f ? 1.0f / sqrtf(d) : 1.0f;
	}
	__syncthreads();
	if ((int)blockIdx.x >= row_blocks) {
		// one row of G: element (row, c) = sum over the partial matrices, four groups of parts, groups added in order
		const int row = (int)blockIdx.x - row_blocks, c = tid & 63, g = tid >> 6;
		const int p0 = (parts * g) / 4, p1 = (parts * (g + 1)) / 4;
		const float* src = gram_part + (long)row * 64 + c;
		float sum = 0.f;
		for (int p = p0; p < p1; p += 48) {
			float v[48];
#pragma unroll
			for (int u = 0; u < 48; ++u) v[u] = p + u < p1 ? src[(long)(p + u) * 4096] : 0.f;
#pragma unroll
			for (int u = 0; u < 48; ++u) sum += v[u];
		}
		__syncthreads();
		s_grp[g][c] = sum;
		__syncthreads();
		if (tid < 64) {
			const float v = ((s_grp[0][tid] + s_grp[1][tid]) + s_grp[2][tid]) + s_grp[3][tid];
			G[(long)row * 64 + tid] = (v * s_scale[tid]) * s_scale[row];
			if (row == 0) scale[tid] = s_scale[tid];
		}
		return;
	}
	const long e = (long)blockIdx.x * 256 + tid;
	f32x4 v = *reinterpret_cast<f32x4*>(P + 4 * e);
	v *= *reinterpret_cast<const f32x4*>(s_scale + (4 * tid) % 64);
	*reinterpret_cast<f32x4*>(P + 4 * e) = v;
	*reinterpret_cast<f32x4*>(&s_v[tid >> 4][4 * (tid & 15)]) = v;
	__syncthreads();
	if (tid < 128 && (int)blockIdx.x < x3_ks) {
		const int r = tid & 31, h = (tid >> 5) & 1, nb = tid >> 6;
		float w[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) w[j] = s_v[8 * h + j][32 * nb + r];
		store_split3(x3_out, blockIdx.x, 2, nb, h, r, w);
	}
}

hipError_t launch_gram64_reduce_scale_all(const float* gram_part, int parts, const float* sumsq_part, int sq_parts, float* G, float* scale, float* P, int len_pad,
                                          void* x3_out, int x3_ks, hipStream_t stream) {
	if (parts < 1 || sq_parts < 1 || len_pad % 16 != 0) return hipErrorInvalidValue;
	const int row_blocks = len_pad / 16;
	hipLaunchKernelGGL(k_mu64_reduce_scale_all_x3, dim3((unsigned)row_blocks + 64), dim3(256), 0, stream, P, gram_part, parts, sumsq_part, sq_parts, G, scale,
	                   reinterpret_cast<bf16x8*>(x3_out), x3_ks, row_blocks);
	return hipGetLastError();
}

// partials -> Graw (unscaled sum), then the launch above: G, scale, the scaled panel and its split image
hipError_t launch_gram64_normalize_all(const float* partials, int parts, float* Graw, float* G, float* scale, float* P, int len_pad, void* x3_out, int x3_ks, hipStream_t stream) {
	hipError_t e = launch_reduce_partials<float>(partials, parts, 4096, Graw, 4096, stream);
	if (e != hipSuccess) return e;
	const int row_blocks = len_pad / 16;
	hipLaunchKernelGGL(k_mu64_scale_all_x3, dim3((unsigned)row_blocks + 1), dim3(256), 0, stream, P, Graw, G, scale, reinterpret_cast<bf16x8*>(x3_out), x3_ks, row_blocks);
	return hipGetLastError();
}

// x3_out (optional): split image of the scaled panel, x3_ks K-steps of 16 rows (len_pad is a multiple of 128)
hipError_t launch_mu64_apply_scale(float* P, int len_pad, const float* scale, hipStream_t stream, void* x3_out, int x3_ks) {
	if (x3_out != nullptr) {
		hipLaunchKernelGGL(k_mu64_apply_scale_x3, dim3((unsigned)(len_pad / 16)), dim3(256), 0, stream, P, scale, reinterpret_cast<bf16x8*>(x3_out), x3_ks);
		return hipGetLastError();
	}
	const long count4 = (long)len_pad * 64 / 4;
	hipLaunchKernelGGL(k_mu64_apply_scale, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, stream, P, count4, scale);
	return hipGetLastError();
}

} // namespace nmfamd
