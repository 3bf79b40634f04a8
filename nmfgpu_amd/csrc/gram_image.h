// gram_image.h -- G = P P^T (64 x 64) of a rank-64 factor panel, computed from the panel's split (3 x bf16) image.
//
// The rank-64 multiplicative update needs W^T W (with the pending column scale folded in) before the H update and H H^T
// before the W update (reference: syrk at source/nmf/AlgorithmMultiplicativeFrobenius.h:171,224).  The split image of the
// panel exists anyway -- it is the factor operand the next product streams -- so the Gram matrix is taken from it, by
// sixteen workgroups that ride in the product launch (kernels_x3.hip) on CUs the product grid leaves idle: the same
// six-term product as kernels_x3.hip (fp32-level accuracy), on the same bytes, from the same L2.
// Round 1 had the update kernel emit one partial 64 x 64 Gram matrix per 64 panel rows (2.5 MB per W update, plus 32
// fp32 MFMAs per wave in the update's critical path) and the passengers reduce them.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.h"
#include "split3.h"

namespace nmfamd {

// One workgroup of 256 threads = one 16 x 16 tile (ti, tj), ti <= tj, of the upper triangle (blocks with ti > tj return at
// once); the mirrored tile is written from the same values, so G is exactly symmetric.  The four waves split the K-steps;
// partial tiles are added in wave order.  With rg.normalize the tile is scaled on both sides by 1 / sqrt(sum of squares of
// the column) (1 where that is 0: kernel::normalizeColumns' `sum > 0` guard, KernelNormalizeColumns.cu:37-58).
// Where the sums of squares come from:
//   rg.colsq_part != nullptr (round 4): the update kernel that wrote the panel left one vector of 64 partial sums per workgroup
//     (k_mu64_update32<true>); every block adds the `colsq_parts` vectors of its 32 columns in a fixed order -- fp32 sums of the
//     squares of the fp32 values, which is what the reference's kernel forms;
//   else: the diagonal of the six-term product itself (rounds 2-3): an off-diagonal block also accumulates the two diagonal tiles
//     -- 18 dependent MFMAs per pair of K-steps instead of 6, a 23 us chain for config 2's W (10 112 rows) that a 42 us product hides
//     and a column shard's 12 us product does not (profiles/r04_shard_trace.md).
// K-split form (rg.ksplit > 1, column shards -- round 4): the chain above is a chain of memory round trips (four pairs in flight per wave, ~1.2 us each under
// the product's stream: 24 us for 10 112 rows whatever the MFMA count), so the K range is cut into rg.ksplit slices and the grid holds 10 * ksplit blocks
// (block = slice * 10 + upper-triangle tile).  A block writes its UNSCALED partial tile (and its mirror image) into rg.G + slice * 4096; the slices are added, in
// order, and scaled by the consumer (k_mu64_update32<false>, qsplit) -- no atomics, no fence, same bits run after run.  Slice 0's diagonal blocks publish
// the column scales (from rg.colsq_part, or ones).
// lds: GRAM_IMAGE_LDS_FLOATS floats.
constexpr int GRAM_IMAGE_LDS_FLOATS = 3456 + 256;
#ifndef GRAM_IMAGE_RING
#define GRAM_IMAGE_RING 4
#endif
// One tile (ti, tj), ti <= tj, over the pairs of K-steps [p0, p1) of the calling wave (the four waves' ranges make up the segment).  partial: the tile goes out
// unscaled into rg.G + slice * 4096 (K-split and spread forms), else scaled into rg.G.  seg: how many segments this workgroup ran before (LDS reuse).
__device__ inline void gram_image_segment(const GramReduceArgs& rg, const int ti, const int tj, const int slice, const bool partial, const int p0, const int p1, float* lds, const int seg) {
	typedef float f32x4v __attribute__((ext_vector_type(4)));
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, l15 = lane & 15;
	const bf16x8* F = reinterpret_cast<const bf16x8*>(rg.image);
	const int KS = rg.image_ks;                       // K-steps of 16 panel rows; step KS is the all-zero step that closes the image
	const bool diag_block = ti == tj;
	const bool from_parts = rg.normalize != 0 && rg.colsq_part != nullptr && slice == 0;
	const bool need_diag = rg.normalize != 0 && !diag_block && rg.colsq_part == nullptr && !partial;
	// slot of (K-step ks, column c, plane, half h): ((ks * 2 + (c >> 5)) * 3 + plane) * 64 + h * 32 + (c & 31)
	const int ci = 16 * ti + l15, cj = 16 * tj + l15;
	const long offi = (long)(ci >> 5) * 192 + (q & 1) * 32 + (ci & 31);
	const long offj = (long)(cj >> 5) * 192 + (q & 1) * 32 + (cj & 31);
	f32x4v aij = {0.f, 0.f, 0.f, 0.f}, aii = aij, ajj = aij;
	// RD pairs of K-steps in flight per wave (6 sixteen-byte loads each).  Round 3 kept ONE pair ahead of the MFMAs.  What bounds the block is not that
	// latency but what ONE CU can pull from L2: 1.9 MB of fragments for the 632 K-steps of config 2's W = 24 us at ~80 GB/s, the same with four or
	// eight pairs in flight (RD = 8: slower, 36 us) -- hidden behind a 42 us product at n = 5 000, the critical path of W^T V for a column shard.  Hence
	// the K-split form (more CUs), see above.  The pairs are still summed in ascending order: same bits.
	constexpr int RD = GRAM_IMAGE_RING;
	bf16x8 a[RD][3], b[RD][3];
	auto fetch = [&](int p, bf16x8 (&fa)[3], bf16x8 (&fb)[3]) {
		int ks = 2 * p + (q >> 1);
		ks = (ks < KS && p < p1) ? ks : KS;             // (past this wave's range: the all-zero step -- adds +0, keeps the loop branch-free)
		const bf16x8* base = F + (long)ks * 384;
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) { fa[pl] = base[offi + pl * 64]; fb[pl] = base[offj + pl * 64]; }      // (no branch in here: the waits must stay counted)
	};
	auto six = [&](const bf16x8 (&x)[3], const bf16x8 (&y)[3], f32x4v acc) -> f32x4v {
		// smallest terms first, as in k_factor_product_x3
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[2], y[0], acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], y[2], acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], y[1], acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], y[0], acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], y[1], acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], y[0], acc, 0, 0, 0);
		return acc;
	};
#pragma unroll
	for (int d = 0; d < RD; ++d) fetch(p0 + d, a[d], b[d]);
	float* s_sq = lds + 3456;                            // [8][32]: partial sums of squares of the tile's 16 row- and 16 column-indices, eight groups of parts
	if (seg > 0) __syncthreads();                        // (the first wave is done with the LDS of the workgroup's previous segment)
	if (from_parts) {
		// thread (g = tid >> 5, c = tid & 31) adds the parts g, g + 8, ... of column c -- up to 40 of them requested together (ONE round trip for the 316 parts
		// of config 2's W: five batches of eight took 5 us of a column shard's 12 us launch), further batches beyond; the groups are added in order below
		const int c = tid & 31, g = tid >> 5;
		const float* src = rg.colsq_part + (c < 16 ? 16 * ti + c : 16 * tj + c - 16);
		float sum = 0.f;
		for (int p = g; p < rg.colsq_parts; p += 320) {
			float v[40];
#pragma unroll
			for (int u = 0; u < 40; ++u) v[u] = p + 8 * u < rg.colsq_parts ? src[(long)(p + 8 * u) * 64] : 0.f;
#pragma unroll
			for (int u = 0; u < 40; ++u) sum += v[u];
		}
		s_sq[g * 32 + c] = sum;
	}
	for (int p = p0; p < p1; p += RD) {
		// RD pairs per turn; a slot is refilled (pair p + d + RD) as soon as its MFMAs are issued
#pragma unroll
		for (int d = 0; d < RD; ++d) {
			aij = six(a[d], b[d], aij);
			if (need_diag) { aii = six(a[d], a[d], aii); ajj = six(b[d], b[d], ajj); }
			fetch(p + d + RD, a[d], b[d]);
		}
	}
	// partial tiles of the four waves, added in wave order.  C/D map of the 16 x 16 MFMA: register g of lane l is
	// row 4 (l >> 4) + g, column l & 15.
	f32x4v* part = reinterpret_cast<f32x4v*>(lds);          // [wave][tile][lane]
	float* s_tile = lds + 3072;                          // [16][16] (diagonal blocks: symmetrisation)   -- 3072 = 4 * 3 * 64 * 4
	float* s_di = lds + 3072 + 64;                       // (off-diagonal blocks reuse the tile area for the two diagonals)
	part[(wave * 3 + 0) * 64 + lane] = aij;
	part[(wave * 3 + 1) * 64 + lane] = aii;
	part[(wave * 3 + 2) * 64 + lane] = ajj;
	__syncthreads();
	if (tid >= 64) return;
	f32x4v vij = part[lane], vii = part[64 + lane], vjj = part[128 + lane];
#pragma unroll
	for (int w = 1; w < 4; ++w) { vij += part[(w * 3) * 64 + lane]; vii += part[(w * 3 + 1) * 64 + lane]; vjj += part[(w * 3 + 2) * 64 + lane]; }
	// (one wave from here on: LDS traffic inside it is ordered by the waits the compiler places)
	float* s_d = diag_block ? s_tile + 256 : s_tile;     // scale of the tile's rows [0..16) and columns [16..32)
	if (diag_block) {
		// G(r, c) and G(c, r) sum the six terms in different orders: keep the upper triangle, mirror it
#pragma unroll
		for (int g = 0; g < 4; ++g) s_tile[(4 * q + g) * 16 + l15] = vij[g];
		__builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int g = 0; g < 4; ++g) { const int r = 4 * q + g, c = l15; vij[g] = r <= c ? s_tile[r * 16 + c] : s_tile[c * 16 + r]; }
		if (lane < 16) {
			const float d = s_tile[lane * 17];
			float s = rg.normalize ? (d > 0.f ? 1.0f / sqrtf(d) : 1.0f) : 1.0f;
			if (from_parts) {
				float q2 = s_sq[lane];
#pragma unroll
				for (int g = 1; g < 8; ++g) q2 += s_sq[g * 32 + lane];
				s = q2 > 0.f ? 1.0f / sqrtf(q2) : 1.0f;
			}
			s_d[lane] = s; s_d[16 + lane] = s;
			if (rg.scale && slice == 0) rg.scale[16 * ti + lane] = s;
		}
	} else {
		if (need_diag) {
			if ((l15 >> 2) == q) {
				// (register picked by a select chain: a run-time index into a vector value goes through scratch)
				const int g = l15 & 3;
				s_di[l15] = g == 0 ? vii[0] : g == 1 ? vii[1] : g == 2 ? vii[2] : vii[3];
				s_di[16 + l15] = g == 0 ? vjj[0] : g == 1 ? vjj[1] : g == 2 ? vjj[2] : vjj[3];
			}
			__builtin_amdgcn_s_waitcnt(0xc07f);
			__builtin_amdgcn_wave_barrier();
			if (lane < 32) { const float d = s_di[lane]; s_d[lane] = d > 0.f ? 1.0f / sqrtf(d) : 1.0f; }
		} else if (from_parts) {
			if (lane < 32) {
				float q2 = s_sq[lane];
#pragma unroll
				for (int g = 1; g < 8; ++g) q2 += s_sq[g * 32 + lane];
				s_d[lane] = q2 > 0.f ? 1.0f / sqrtf(q2) : 1.0f;
			}
		} else if (lane < 32) s_d[lane] = 1.0f;
	}
	__builtin_amdgcn_s_waitcnt(0xc07f);
	__builtin_amdgcn_wave_barrier();
	if (partial) {
		// this slice's partial tile as it is; the consumer adds the slices and scales.  The scales: slice 0, off-diagonal blocks hold them too but only the
		// diagonal ones publish (above)
		float* Gk = rg.G + (long)slice * 4096;
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const int r = 4 * q + g, c = l15;
			Gk[(long)(16 * ti + r) * 64 + 16 * tj + c] = vij[g];
			if (!diag_block) Gk[(long)(16 * tj + c) * 64 + 16 * ti + r] = vij[g];
		}
		return;
	}
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		const int r = 4 * q + g, c = l15;
		const float v = (vij[g] * s_d[16 + c]) * s_d[r];
		rg.G[(long)(16 * ti + r) * 64 + 16 * tj + c] = v;
		if (!diag_block) rg.G[(long)(16 * tj + c) * 64 + 16 * ti + r] = v;
	}
}

// tile t of the upper triangle, row by row: (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
__device__ inline void gram_image_tile(int t, int* ti, int* tj) {
	*ti = t < 4 ? 0 : t < 7 ? 1 : t < 9 ? 2 : 3;
	*tj = t < 4 ? t : t < 7 ? t - 3 : t < 9 ? t - 5 : 3;
}

__device__ inline void gram_image_block(const GramReduceArgs& rg, int blk, float* lds) {
	const int wave = threadIdx.x >> 6;
	const int pairs = (rg.image_ks + 2) / 2;           // 32 k per MFMA = two K-steps
	if (rg.spread > 0) {
		// Spread form (round 5, the sixteen passengers of a whole problem's W^T V): the one-slice form leaves six of the sixteen workgroups idle (the lower-triangle
		// tiles) and makes the ten others the LAST workgroups of the launch (config 2: 37 - 40 us under the product's stream against 33 - 36 for the product blocks).
		// Here the ten tiles' K ranges, end to end, are dealt evenly to all sixteen: a workgroup runs 5 / 8 of a tile's range -- the end of one tile and the
		// start of the next -- and a tile arrives in up to GRAM_SPREAD_SLICES unscaled pieces (piece = how many workgroups before this one worked on the tile);
		// the last workgroup to finish adds them in order and scales the sum (below): long before the product's own workgroups are done.
		const long total = (long)GRAM_IMAGE_TILES * pairs;
		const long w0 = total * blk / GRAM_REDUCE_BLOCKS, w1 = total * (blk + 1) / GRAM_REDUCE_BLOCKS;
		int seg = 0;
		for (int t = (int)(w0 / pairs); t < GRAM_IMAGE_TILES && (long)t * pairs < w1; ++t) {
			const long a = w0 > (long)t * pairs ? w0 : (long)t * pairs, b = w1 < (long)(t + 1) * pairs ? w1 : (long)(t + 1) * pairs;
			if (b <= a) continue;
			int first = blk;                         // the first workgroup whose range reaches into tile t
			while (first > 0 && total * first / GRAM_REDUCE_BLOCKS > (long)t * pairs) --first;
			int ti, tj;
			gram_image_tile(t, &ti, &tj);
			const int sb = (int)(a - (long)t * pairs), len = (int)(b - a);
			gram_image_segment(rg, ti, tj, blk - first, true, sb + (int)(((long)len * wave) / 4), sb + (int)(((long)len * (wave + 1)) / 4), lds, seg++);
		}
		if (rg.spread_out == nullptr || rg.spread_counter == nullptr) return;
		// hand-over (the scheme of tri_gram_tile.h): this workgroup's pieces are out -- each wave waits for its own stores, ONE release for the workgroup, one count;
		// the last of the sixteen finishes the matrix (nobody waits for anybody)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		int* s_last = reinterpret_cast<int*>(lds);
		if (threadIdx.x == 0) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			const unsigned old = __hip_atomic_fetch_add(rg.spread_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const int last = old == (unsigned)(GRAM_REDUCE_BLOCKS - 1);
			if (last) {
				__hip_atomic_store(rg.spread_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the next launch finds zero)
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
			*s_last = last;
		}
		__syncthreads();
		if (*s_last == 0) return;
		for (int e = threadIdx.x; e < 4096; e += 256) {
			const int r = e >> 6, c = e & 63;
			float v = rg.G[e];
#pragma unroll
			for (int k = 1; k < GRAM_SPREAD_SLICES; ++k) v += rg.G[(long)k * 4096 + e];
			rg.spread_out[e] = rg.scale != nullptr ? (v * rg.scale[c]) * rg.scale[r] : v;
		}
		return;
	}
	const int ksplit = rg.ksplit > 1 ? rg.ksplit : 1;
	int ti = blk >> 2, tj = blk & 3, slice = 0;
	if (ksplit > 1) {
		slice = blk / GRAM_IMAGE_TILES;
		if (slice >= ksplit) return;
		gram_image_tile(blk % GRAM_IMAGE_TILES, &ti, &tj);
	}
	if (ti > tj) return;
	const int piece = slice * 4 + wave, pieces = 4 * ksplit;
	gram_image_segment(rg, ti, tj, slice, ksplit > 1, (int)(((long)pairs * piece) / pieces), (int)(((long)pairs * (piece + 1)) / pieces), lds, 0);
}

} // namespace nmfamd
