// mu64_update32.h -- body of the rank-64 multiplicative update on 32-column tiles (k_mu64_update32, kernels_mu64.hip), as a device function: the stand-alone
// launch runs it once per workgroup; the split-operand product launches of the fused iteration run it as their TAIL (kernels_x3.hip, UpdateTail) -- same
// instructions on the same values, so both forms leave the same bits.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.h"
#include "split3.h"

namespace nmfamd {

// The update on 32-column tiles, Gram matrices taken elsewhere (gram_image.h).
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
// QS = 0: the slice count is the run-time argument qsplit (1 = the finished matrix)
// unit: which 32-column tile; nunits: how many there are (the r error terms of the W update are dealt over them)
// LDS: s_num[32][68] (reduced numerator, later the new values), s_old[32][68] (old values, scaled for the W update), s_ps[4][32]
template <bool IS_W, int U, int QS>
__device__ __forceinline__ void mu64_update32_body(
	const int unit, const int nunits, float (*s_num)[68], float (*s_old)[68], float (*s_ps)[32],
	float* __restrict__ P, const float* __restrict__ slabs, int S, long slab_stride,
	const float* __restrict__ Q, const float* __restrict__ scale, float eps,
	float* __restrict__ ps, int len_valid, const float* __restrict__ Gprev, int compute_error, bf16x8* __restrict__ x3_out, int x3_ks, const PeerSlabs& peers,
	float* __restrict__ colsq_part, int qsplit, float* __restrict__ q_out) {
	typedef float f32x4v __attribute__((ext_vector_type(4)));
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int q = lane >> 4, l15 = lane & 15;
	const long tile = (long)unit * 32 * 64;
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
	const int nq = QS > 0 ? QS : qsplit;      // (QS == 0: the slice count is a run-time value -- the tail of a product launch, kernels_x3.hip)
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
		if (QS != 1 || U32_Q_EARLY) {
			// Q arrives as QS unscaled K slices of Wu^T Wu (gram_image.h, K-split form): requested right BEHIND the slabs' first batch -- the wait for that batch then
			// leaves these in flight and they arrive while the slabs are summed and parked in LDS (in front of the batch they made every workgroup of the launch wait
			// for the same few lines first: 7.4 -> 10 us with four slices; behind the slab loop they cost a round trip of their own)
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int u = 0; u < 4; ++u) qa[u] = *reinterpret_cast<const f32x4v*>(Q + (long)(16 * wave + l15) * 64 + 16 * q + 4 * u);
			if (QS == 0) {
				// (same sums in the same order as the compile-time form below: slice after slice)
				for (int k = 1; k < nq; ++k)
#pragma unroll
					for (int u = 0; u < 4; ++u) qa[u] += *reinterpret_cast<const f32x4v*>(Q + (long)k * 4096 + (long)(16 * wave + l15) * 64 + 16 * q + 4 * u);
			}
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
	if (!IS_W && (QS > 1 || (QS == 0 && nq > 1))) {
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
			if (unit == 0 && q_out != nullptr) *reinterpret_cast<f32x4v*>(q_out + (long)(16 * wave + l15) * 64 + 16 * q + 4 * u) = qa[u];
		}
	}
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		// the pending column scale of W goes on the numerator W^T V (H update) or on W itself (W update)
		if (IS_W) ol[j] *= sc; else nl[j] *= sc;
		*reinterpret_cast<f32x4v*>(&s_num[yl0 + 16 * j][c4]) = nl[j];
		*reinterpret_cast<f32x4v*>(&s_old[yl0 + 16 * j][c4]) = ol[j];
	}
	__syncthreads();

	// ---- den = Q * old on the matrix pipe, two independent 16 x 16 accumulators per wave -------
	f32x4v ob[2][4];
#pragma unroll
	for (int yt = 0; yt < 2; ++yt)
#pragma unroll
		for (int u = 0; u < 4; ++u) ob[yt][u] = *reinterpret_cast<const f32x4v*>(&s_old[16 * yt + l15][16 * q + 4 * u]);
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
		if (l15 == 0) *reinterpret_cast<f32x4v*>(colsq_part + (long)unit * 64 + 16 * wave + 4 * q) = sq;
	}
	if (IS_W && compute_error && wave == 0) {
		// r terms of tr(H H^T W^T W): ps(d) = sum_i (H H^T)(d, i) (W^T W)(i, d)   (AlgorithmMultiplicativeFrobenius.h:212) -- one term per workgroup (round 3:
		// all 64 in workgroup 0, sixteen dependent round trips per wave: that workgroup ran 16 us on every error iteration while the others took 6)
		for (int d = unit; d < 64; d += nunits) {
			float v = Q[(long)lane * 64 + d] * Gprev[(long)d * 64 + lane];
			for (int w = 32; w > 0; w >>= 1) v += __shfl_xor(v, w);
			if (lane == 0) ps[d] = v;
		}
	}
	__syncthreads();
	if (!IS_W && compute_error && tid < 32) {
		// per-column terms of tr(H^T W^T V) (kernel::traceMultiplication, AlgorithmMultiplicativeFrobenius.h:194-197)
		const int ycol = unit * 32 + tid;
		if (ycol < len_valid) ps[ycol] = ((s_ps[0][tid] + s_ps[1][tid]) + s_ps[2][tid]) + s_ps[3][tid];
	}
	// ---- result out (coalesced) and the split image of the two K-steps this tile is --------------
#pragma unroll
	for (int j = 0; j < 2; ++j)
		*reinterpret_cast<f32x4v*>(P + tile + 4 * (tid + 256 * j)) = *reinterpret_cast<const f32x4v*>(&s_num[yl0 + 16 * j][c4]);
	{
		const int r = tid & 31, h = (tid >> 5) & 1, nb = (tid >> 6) & 1, kk = tid >> 7;
		const long ks = 2l * unit + kk;
		if (ks < x3_ks) {
			float v[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const int yl = 16 * kk + 8 * h + j;
				v[j] = unit * 32 + yl < len_valid ? s_num[yl][32 * nb + r] : 0.f;
			}
			store_split3(x3_out, ks, 2, nb, h, r, v);
		}
	}
}

} // namespace nmfamd
