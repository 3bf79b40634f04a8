// kernels_bf16.hip -- bf16-operand form of the factor product (extension: the reference computes in
// float / double only, include/nmfgpu.h:298-299; selected by Parameter "precision" = 1).
//
// The streamed matrix and the factor panel are ROUNDED TO bf16 as MFMA operands; products are exact
// in fp32 and accumulate in fp32 (v_mfma_f32_32x32x16_bf16); the factors themselves, the update,
// the Gram matrices and the error terms stay fp32.  At 16x the fp32 MFMA rate the product is no
// longer MFMA-bound but HBM-bound (half the bytes of V per pass).
//
// Operand storage follows the MFMA fragment order so that every operand load is 16 B per lane over
// consecutive lanes (1 KiB per wave instruction) and a wave streams one sequential region:
//   A (streamed, x-tiled by 128):  Ab[(((xt*KS + ks)*4 + b)*2 + h)*32 + r][8]   = A(x = 128xt+32b+r, y = 16ks+8h+j)
//   F (factor panel, 64 rows):     Fb[((ks*2 + nb)*2 + h)*32 + r][8]            = F(c = 32nb+r,      y = 16ks+8h+j)
// (lane l = 32h + r holds A[row r][k = 8h + j] / B[k = 8h + j][col r], j = 0..7:
//  /opt/skills/guides/cdna_hip_programming.md, "A/B operand lane maps, bf16").
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kernels.h"

namespace nmfamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BF_WAVES = 8;

// src: column-major fp32.  transposed = 0: A(x, y) = src[y*ld + x];  1: A(x, y) = src[x*ld + y].
__global__ __launch_bounds__(256) void k_pack_stream_bf16(const float* __restrict__ src, long ld, int X, int Y, int transposed,
                                                          bf16x8* __restrict__ dst, int KS, long frags) {
	const long f = (long)blockIdx.x * 256 + threadIdx.x;   // one 16-byte fragment row per thread
	if (f >= frags) return;
	const int r = (int)(f & 31), h = (int)((f >> 5) & 1), b = (int)((f >> 6) & 3);
	const long t = f >> 8;
	const int ks = (int)(t % KS);
	const long xt = t / KS;
	const long x = xt * 128 + 32 * b + r;
	bf16x8 o;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const long y = 16l * ks + 8 * h + j;
		float v = 0.f;
		if (x < X && y < Y) v = transposed ? src[x * ld + y] : src[y * ld + x];
		o[j] = (__bf16)v;
	}
	dst[f] = o;
}

hipError_t launch_pack_stream_bf16(const float* src, long ld, int X, int Y, bool transposed, void* dst, int xtiles, int KS, hipStream_t stream) {
	const long frags = (long)xtiles * KS * 256;
	hipLaunchKernelGGL(k_pack_stream_bf16, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, stream, src, ld, X, Y, transposed ? 1 : 0,
	                   reinterpret_cast<bf16x8*>(dst), KS, frags);
	return hipGetLastError();
}

// P: fp32 panel [y][RP]; len = valid panel rows (y); NBT = RP / 32 column blocks.
__global__ __launch_bounds__(256) void k_pack_panel_bf16(const float* __restrict__ P, int RP, int NBT, int len, bf16x8* __restrict__ dst, long frags) {
	const long f = (long)blockIdx.x * 256 + threadIdx.x;
	if (f >= frags) return;
	const int r = (int)(f & 31), h = (int)((f >> 5) & 1);
	const long t = f >> 6;
	const int nb = (int)(t % NBT);
	const long ks = t / NBT;
	bf16x8 o;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const long y = 16 * ks + 8 * h + j;
		o[j] = (__bf16)(y < len ? P[y * RP + 32 * nb + r] : 0.f);
	}
	dst[f] = o;
}

hipError_t launch_pack_panel_bf16(const float* P, int RP, int len, void* dst, int KS, hipStream_t stream) {
	const int NBT = RP / 32;
	const long frags = (long)KS * NBT * 64;
	hipLaunchKernelGGL(k_pack_panel_bf16, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, stream, P, RP, NBT, len, reinterpret_cast<bf16x8*>(dst), frags);
	return hipGetLastError();
}

// Passenger Gram reduction, as in k_factor_product_f32 (kernels.hip); duplicated here because the two
// kernels live in different translation units.
__device__ inline void gram_reduce_block_bf(const GramReduceArgs& rg, int blk, float* lds) {
	const int tid = threadIdx.x;
	float* s_scale = lds;
	float* s_tmp = lds + 64;
	const int parts = rg.parts;
	if (rg.normalize) {
		if (tid < 128) {
			const int c = tid & 63, g = tid >> 6;
			const int p0 = (parts * g) / 2, p1 = (parts * (g + 1)) / 2;
			float sum = 0.f;
			for (int p = p0; p < p1; ++p) sum += rg.partials[(long)p * 4096 + c * 65];
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
		int p = p0;
		for (; p + 8 <= p1; p += 8) {
			float v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = rg.partials[(long)(p + u) * 4096 + e];
#pragma unroll
			for (int u = 0; u < 8; ++u) sum += v[u];
		}
		for (; p < p1; ++p) sum += rg.partials[(long)p * 4096 + e];
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

// Workgroup = 8 waves = one 128-row x-tile times one slice of the reduction range (K-steps of 16 y)
// times CH chunks of 64 panel columns; the slice is cut into KP = 8 / CH wave pieces.  A wave keeps a
// 128 x 64 accumulator block (8 tiles), streams its piece through a D-deep register ring, and the KP
// pieces of a chunk are summed through LDS in piece order; one fp32 slab per slice.
//   CH = 1: padded rank 64 (all eight waves cut K);  CH = 2 / 4: 128 / 256 panel columns per pass over
//   A -- the CH waves that share a K piece read the same A fragments (one HBM fetch, L1/L2 hits).
template <int D, int CH>
__global__ __launch_bounds__(512, 2) void k_factor_product_bf16(
	const bf16x8* __restrict__ A, long tile_frags,      // 16-byte fragments per x-tile = KS * 256
	const bf16x8* __restrict__ F, int NBT,              // factor fragments, NBT = RP / 32 column blocks per K-step
	float* __restrict__ slabs, long slab_stride, int RP,
	int steps_total, int splits, GramReduceArgs rg) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	constexpr int KP = BF_WAVES / CH;
	if (blockIdx.y == (unsigned)splits) {
		if (blockIdx.x < GRAM_REDUCE_BLOCKS) gram_reduce_block_bf(rg, blockIdx.x, lds);
		return;
	}
	const int xt = blockIdx.x, sp = blockIdx.y, grp = blockIdx.z;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const int chunk = wave % CH, kp = wave / CH;
	const int cg = grp * CH + chunk;                    // 64-column chunk of the panel
	const int b0 = (int)(((long)steps_total * sp) / splits);
	const int b1 = (int)(((long)steps_total * (sp + 1)) / splits);
	const int s0 = b0 + (int)(((long)(b1 - b0) * kp) / KP);
	const int s1 = b0 + (int)(((long)(b1 - b0) * (kp + 1)) / KP);
	const int steps = s1 - s0;
	const long fstep = (long)NBT * 64;                  // factor fragments per K-step
	f32x16 acc[4][2];
#pragma unroll
	for (int b = 0; b < 4; ++b)
#pragma unroll
		for (int nb = 0; nb < 2; ++nb)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[b][nb][g] = 0.f;

	if (steps > 0) {
		const bf16x8* ap = A + (long)xt * tile_frags + (long)s0 * 256 + lane;   // + b*64 per M-block, + 256 per K-step
		const bf16x8* fp = F + (long)s0 * fstep + (long)cg * 128 + lane;        // + nb*64 per N-block, + fstep per K-step
		const int last = steps - 1;
		bf16x8 va[D][4], fb[D][2];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int st = d < last ? d : last;
#pragma unroll
			for (int b = 0; b < 4; ++b) va[d][b] = ap[(long)st * 256 + b * 64];
#pragma unroll
			for (int nb = 0; nb < 2; ++nb) fb[d][nb] = fp[(long)st * fstep + nb * 64];
		}
		__builtin_amdgcn_sched_barrier(0);
		int t = 0;
		for (; t + D <= steps; t += D) {
#pragma unroll
			for (int d = 0; d < D; ++d) {
#pragma unroll
				for (int b = 0; b < 4; ++b)
#pragma unroll
					for (int nb = 0; nb < 2; ++nb)
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[d][b], fb[d][nb], acc[b][nb], 0, 0, 0);
				int st = t + D + d;
				st = st < last ? st : last;
#pragma unroll
				for (int b = 0; b < 4; ++b) va[d][b] = ap[(long)st * 256 + b * 64];
#pragma unroll
				for (int nb = 0; nb < 2; ++nb) fb[d][nb] = fp[(long)st * fstep + nb * 64];
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		const int rem = steps - t;
#pragma unroll
		for (int d = 0; d < D; ++d) {
			if (d < rem) {
#pragma unroll
				for (int b = 0; b < 4; ++b)
#pragma unroll
					for (int nb = 0; nb < 2; ++nb)
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[d][b], fb[d][nb], acc[b][nb], 0, 0, 0);
			}
		}
	}

	// in-workgroup sum through LDS, two M-blocks (four tiles) per round; C/D map: register g of lane l
	// is row (g&3) + 8*(g>>2) + 4*(l>>5), column l&31.  A round leaves CH * 16 (chunk, tile, q) slices
	// of 64 x 16 B, each the sum of KP pieces in piece order; every wave sums 2 * CH of them.
	f32x4* l4 = reinterpret_cast<f32x4*>(lds);
	float* slab = slabs + (long)sp * slab_stride;
#pragma unroll
	for (int rd = 0; rd < 2; ++rd) {
		if (rd > 0) __syncthreads();
#pragma unroll
		for (int tl = 0; tl < 4; ++tl) {
			const int b = 2 * rd + (tl >> 1), nb = tl & 1;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				f32x4 v;
				v[0] = acc[b][nb][4 * q + 0]; v[1] = acc[b][nb][4 * q + 1];
				v[2] = acc[b][nb][4 * q + 2]; v[3] = acc[b][nb][4 * q + 3];
				l4[((wave * 4 + tl) * 4 + q) * 64 + lane] = v;
			}
		}
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 2 * CH; ++i) {
			const int sl = wave * 2 * CH + i;           // slice = (chunk, tile, q)
			const int q = sl & 3, tl = (sl >> 2) & 3, ch = sl >> 4;
			const int b = 2 * rd + (tl >> 1), nb = tl & 1;
			f32x4 s = l4[((ch * 4 + tl) * 4 + q) * 64 + lane];
#pragma unroll
			for (int p = 1; p < KP; ++p) s += l4[(((p * CH + ch) * 4 + tl) * 4 + q) * 64 + lane];
			const int c = 64 * (grp * CH + ch) + 32 * nb + l31;
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) {
				const int x = xt * 128 + 32 * b + gi + 8 * q + 4 * half;
				slab[(long)x * RP + c] = s[gi];
			}
		}
	}
}

template <int D, int CH>
static hipError_t launch_fp_bf16(const FactorProductPlan& p, const void* A, int KS, const void* F, int RP,
                                 float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg) {
	GramReduceArgs none = {nullptr, 0, nullptr, nullptr, 0};
	const bool with_reduce = rg != nullptr && rg->partials != nullptr && p.xtiles >= GRAM_REDUCE_BLOCKS;
	if (rg != nullptr && rg->partials != nullptr && (!with_reduce || CH != 1)) return hipErrorInvalidValue;
	dim3 grid(p.xtiles, p.splits + (with_reduce ? 1 : 0), RP / (64 * CH)), block(512);
	const size_t lds_bytes = 8 * 4 * 4 * 64 * sizeof(f32x4);
	static bool attr_done = false;
	if (!attr_done) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_product_bf16<D, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		if (e != hipSuccess) return e;
		attr_done = true;
	}
	hipLaunchKernelGGL((k_factor_product_bf16<D, CH>), grid, block, lds_bytes, stream,
	                   reinterpret_cast<const bf16x8*>(A), (long)KS * 256, reinterpret_cast<const bf16x8*>(F), RP / 32,
	                   slabs, slab_stride, RP, KS, p.splits, with_reduce ? *rg : none);
	return hipGetLastError();
}

// Every wave piece gets at least 8 K-steps (the ring is 4 deep); as many slices as fill the chip.
int plan_splits_bf16(int xtiles, int KS, int RP, int num_cus) {
	const int KP = RP == 64 ? 8 : (RP % 256 == 0 ? 2 : 4);
	const int by_fill = std::max(1, num_cus / std::max(1, xtiles));
	const int by_depth = std::max(1, KS / (8 * KP));
	return std::max(1, std::min(by_fill, by_depth));
}

// RP: padded rank of the panel (64, or a multiple of 128).  256 columns per pass over A when RP is a
// multiple of 256, else 128 (64 for RP = 64); wider panels take RP / 256 (RP / 128) passes (grid.z).
hipError_t launch_factor_product_bf16(const FactorProductPlan& p, const void* A, int KS, const void* F, int RP,
                                      float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg) {
	constexpr int D = 4;
	if (RP == 64) return launch_fp_bf16<D, 1>(p, A, KS, F, RP, slabs, slab_stride, stream, rg);
	if (RP % 256 == 0) return launch_fp_bf16<D, 4>(p, A, KS, F, RP, slabs, slab_stride, stream, rg);
	if (RP % 128 == 0) return launch_fp_bf16<D, 2>(p, A, KS, F, RP, slabs, slab_stride, stream, rg);
	return hipErrorInvalidValue;
}

} // namespace nmfamd
