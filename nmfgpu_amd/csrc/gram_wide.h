// gram_wide.h -- one K slice of one 128 x 128 super-block of a wide panel's Gram matrix (device code shared by k_gram_wide_x3, kernels_wide.hip, and the passenger
// workgroups of the split-operand product at padded ranks 128 ... 512, kernels_x3.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "split3.h"

namespace nmfamd {

typedef float gw_f32x16 __attribute__((ext_vector_type(16)));

// The same Gram matrix at fp32 accuracy on the bf16 matrix pipe (kernels_x3.hip): K-step = 16 panel rows, a lane
// (c = l & 31, h = l >> 5) gathers P(16 s + 8 h + j, block + c), j = 0..7, with eight coalesced 4-byte loads and splits
// them exactly into three bf16 terms; six 32x32x16 MFMAs per tile and K-step replace eight 32x32x2 fp32 ones at a quarter of
// their cycles each.  Rows past len are zero up to the padded length (a multiple of 128).
// MIRROR = false (the fused sequence, k_gram_reduce_x3 behind it): only 32 x 32 blocks ON the diagonal are mirrored here; the reduction mirrors the blocks above it
// once instead of every slice doing it with 4-byte stores a row apart (measured with parts of the kernel compiled out: of 15.6 us per launch at padded rank 256 the mirrored stores were 6.4, the direct
// ones 4.1, the MFMAs 3.4, the operand split 0.7; a launch without stores takes 5.0).
// slice / super_block: what blockIdx.x / blockIdx.y are in the stand-alone launch (parts slices; super-blocks (I, J), I <= J, row by row).  256 threads, no LDS.
template <int D, bool MIRROR>
__device__ __forceinline__ void gram_wide_slice(const float* __restrict__ P, int RP, int len, int parts, float* __restrict__ partial, int slice, int super_block) {
	typedef gw_f32x16 f32x16;
	const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const int nb = RP / 128;
	int I = 0, rem = super_block;
	while (rem >= nb - I) { rem -= nb - I; ++I; }
	const int J = I + rem;
	const int wi = wave >> 1, wj = wave & 1;
	const int ca = 128 * I + 64 * wi, cb = 128 * J + 64 * wj;
	const int steps_total = (len + 15) / 16;
	const int s0 = (int)(((long)steps_total * slice) / parts);
	const int s1 = (int)(((long)steps_total * (slice + 1)) / parts);
	const int steps = s1 - s0;

	f32x16 acc[2][2];
#pragma unroll
	for (int a = 0; a < 2; ++a)
#pragma unroll
		for (int b = 0; b < 2; ++b)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

	if (steps > 0) {
		const float* pa = P + ((long)16 * s0 + 8 * half) * RP + ca + l31;
		const float* pb = P + ((long)16 * s0 + 8 * half) * RP + cb + l31;
		const int last = steps - 1;
		float va[D][2][8], vb[D][2][8];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int t = d < last ? d : last;
#pragma unroll
			for (int k = 0; k < 2; ++k)
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					va[d][k][j] = pa[((long)16 * t + j) * RP + 32 * k];
					vb[d][k][j] = pb[((long)16 * t + j) * RP + 32 * k];
				}
		}
		__builtin_amdgcn_sched_barrier(0);
		int t = 0;
		for (; t + D <= steps; t += D) {
#pragma unroll
			for (int d = 0; d < D; ++d) {
				bf16x8 ah[2][3], bh[2][3];
#pragma unroll
				for (int k = 0; k < 2; ++k) {
					split3(va[d][k], ah[k][0], ah[k][1], ah[k][2]);
					split3(vb[d][k], bh[k][0], bh[k][1], bh[k][2]);
				}
				int tn = t + D + d;
				tn = tn < last ? tn : last;
#pragma unroll
				for (int k = 0; k < 2; ++k)
#pragma unroll
					for (int j = 0; j < 8; ++j) {
						va[d][k][j] = pa[((long)16 * tn + j) * RP + 32 * k];
						vb[d][k][j] = pb[((long)16 * tn + j) * RP + 32 * k];
					}
#pragma unroll
				for (int a = 0; a < 2; ++a)
#pragma unroll
					for (int b = 0; b < 2; ++b) {
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][2], bh[b][0], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][0], bh[b][2], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][1], bh[b][1], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][1], bh[b][0], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][0], bh[b][1], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][0], bh[b][0], acc[a][b], 0, 0, 0);
					}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		const int remn = steps - t;
#pragma unroll
		for (int d = 0; d < D; ++d) {
			if (d < remn) {
				bf16x8 ah[2][3], bh[2][3];
#pragma unroll
				for (int k = 0; k < 2; ++k) {
					split3(va[d][k], ah[k][0], ah[k][1], ah[k][2]);
					split3(vb[d][k], bh[k][0], bh[k][1], bh[k][2]);
				}
#pragma unroll
				for (int a = 0; a < 2; ++a)
#pragma unroll
					for (int b = 0; b < 2; ++b) {
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][2], bh[b][0], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][0], bh[b][2], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][1], bh[b][1], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][1], bh[b][0], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][0], bh[b][1], acc[a][b], 0, 0, 0);
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a][0], bh[b][0], acc[a][b], 0, 0, 0);
					}
			}
		}
	}
	// Only the upper triangle (r <= c) is taken from the accumulators and mirrored: the six-term sum of (r, c) and of
	// (c, r) adds the same products in a different order, and G must be exactly symmetric (the reference computes one
	// triangle, syrk, and reads it through symm).
	float* out = partial + (long)slice * RP * RP;
#pragma unroll
	for (int a = 0; a < 2; ++a)
#pragma unroll
		for (int b = 0; b < 2; ++b)
#pragma unroll
			for (int g = 0; g < 16; ++g) {
				const int r = ca + 32 * a + (g & 3) + 8 * (g >> 2) + 4 * half;
				const int c = cb + 32 * b + l31;
				if (r <= c) {
					out[(long)r * RP + c] = acc[a][b][g];
					if (r != c && (MIRROR || ca + 32 * a == cb + 32 * b)) out[(long)c * RP + r] = acc[a][b][g];
				}
			}
}

} // namespace nmfamd
