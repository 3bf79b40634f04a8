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
// partial tiles are added in wave order.  With rg.normalize the tile is scaled on both sides by 1 / sqrt(diag) (1 where
// the diagonal is 0: kernel::normalizeColumns' `sum > 0` guard): an off-diagonal block also accumulates the two diagonal
// tiles it needs the diagonal of -- the same instructions on the same data as the diagonal blocks run, hence the same bits.
// lds: 3456 floats.
__device__ inline void gram_image_block(const GramReduceArgs& rg, int blk, float* lds) {
	typedef float f32x4v __attribute__((ext_vector_type(4)));
	const int ti = blk >> 2, tj = blk & 3;
	if (ti > tj) return;
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, l15 = lane & 15;
	const bf16x8* F = reinterpret_cast<const bf16x8*>(rg.image);
	const int KS = rg.image_ks;                       // K-steps of 16 panel rows; step KS is the all-zero step that closes the image
	const bool diag_block = ti == tj;
	const bool need_diag = rg.normalize != 0 && !diag_block;
	// slot of (K-step ks, column c, plane, half h): ((ks * 2 + (c >> 5)) * 3 + plane) * 64 + h * 32 + (c & 31)
	const int ci = 16 * ti + l15, cj = 16 * tj + l15;
	const long offi = (long)(ci >> 5) * 192 + (q & 1) * 32 + (ci & 31);
	const long offj = (long)(cj >> 5) * 192 + (q & 1) * 32 + (cj & 31);
	const int pairs = (KS + 2) / 2;                    // 32 k per MFMA = two K-steps
	const int p0 = (pairs * wave) / 4, p1 = (pairs * (wave + 1)) / 4;
	f32x4v aij = {0.f, 0.f, 0.f, 0.f}, aii = aij, ajj = aij;
	bf16x8 a[2][3], b[2][3];
	auto fetch = [&](int p, bf16x8 (&fa)[3], bf16x8 (&fb)[3]) {
		int ks = 2 * p + (q >> 1);
		ks = ks < KS ? ks : KS;
		const bf16x8* base = F + (long)ks * 384;
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) { fa[pl] = base[offi + pl * 64]; fb[pl] = base[offj + pl * 64]; }
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
	if (p0 < p1) fetch(p0, a[0], b[0]);
	for (int p = p0; p < p1; p += 2) {
		// two pairs per turn, the fetch of each one a pair ahead of its MFMAs
		if (p + 1 < p1) fetch(p + 1, a[1], b[1]);
		aij = six(a[0], b[0], aij);
		if (need_diag) { aii = six(a[0], a[0], aii); ajj = six(b[0], b[0], ajj); }
		if (p + 2 < p1) fetch(p + 2, a[0], b[0]);
		if (p + 1 < p1) {
			aij = six(a[1], b[1], aij);
			if (need_diag) { aii = six(a[1], a[1], aii); ajj = six(b[1], b[1], ajj); }
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
			const float s = rg.normalize ? (d > 0.f ? 1.0f / sqrtf(d) : 1.0f) : 1.0f;
			s_d[lane] = s; s_d[16 + lane] = s;
			if (rg.scale) rg.scale[16 * ti + lane] = s;
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
		} else if (lane < 32) s_d[lane] = 1.0f;
	}
	__builtin_amdgcn_s_waitcnt(0xc07f);
	__builtin_amdgcn_wave_barrier();
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		const int r = 4 * q + g, c = l15;
		const float v = (vij[g] * s_d[16 + c]) * s_d[r];
		rg.G[(long)(16 * ti + r) * 64 + 16 * tj + c] = v;
		if (!diag_block) rg.G[(long)(16 * tj + c) * 64 + 16 * ti + r] = v;
	}
}

} // namespace nmfamd
