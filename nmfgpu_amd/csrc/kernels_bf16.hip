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
#include <cstdlib>

#include "kernels.h"
#include "tri_gram_tile.h"
#include "tuning.h"

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

// ---- 256 panel columns per pass: A through LDS --------------------------------------------------
// With CH = 4 the four waves that share a K piece request the SAME A fragments: the bytes they keep in
// flight are counted four times against the registers that hold them, and the kernel above stalls at
// ~3.5 TB/s (Little's law: 32 KiB of distinct A in flight per CU).  Here every wave loads DISTINCT A
// blocks (register staged, SETS - 1 stages in flight), parks them in a two-buffer LDS image in fragment
// order (three buffers), and all waves of a piece read their operands from there (ds_read_b128, lane-linear, conflict
// free).  F fragments are private to a wave and stay on the direct global -> register ring.
//   stage = G = 2 K-steps of both pieces = 16 blocks of 1 KiB; wave w loads blocks 2w, 2w + 1:
//   block ((pk * G + g) * 4 + b) = fragments of piece pk, K-step g of the stage, M-block b.
//   stage q: loads issued in iteration q - SETS - 1, written to LDS buffer q % 3 in iteration q - 2 (after the
//   barrier that retires the reads of stage q - 3), visible after the barrier of iteration q - 1, so that the
//   operand reads of a K-step can be issued one K-step ahead of its MFMAs, across the stage boundary.
template <int SETS>
__global__ __launch_bounds__(512, 2) void k_factor_product_bf16_staged(
	const bf16x8* __restrict__ A, long tile_frags, const bf16x8* __restrict__ F, int NBT,
	float* __restrict__ slabs, long slab_stride, int RP, int steps_total, int splits) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	constexpr int CH = 4, KP = 2, G = 2;
	const int xt = blockIdx.x, sp = blockIdx.y, grp = blockIdx.z;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const int chunk = wave % CH, kp = wave / CH;
	const int cg = grp * CH + chunk;
	const int b0 = (int)(((long)steps_total * sp) / splits);
	const int b1 = (int)(((long)steps_total * (sp + 1)) / splits);
	const int mid = b0 + (b1 - b0) / 2;
	const int s0 = kp == 0 ? b0 : mid, steps = kp == 0 ? mid - b0 : b1 - mid;      // this wave's piece
	const int nst = ((b1 - mid) + G - 1) / G;                                       // stage count (piece 1 is the longer one)
	const long fstep = (long)NBT * 64;

	// loader role: blocks 2w, 2w + 1 of every stage
	const int lpk = wave >> 2, lg = (wave >> 1) & 1, lb = 2 * (wave & 1);
	const int ls0 = lpk == 0 ? b0 : mid, lsteps = lpk == 0 ? mid - b0 : b1 - mid;
	const int llast = lsteps > 0 ? lsteps - 1 : 0;
	const bf16x8* lap = A + (long)xt * tile_frags + (long)ls0 * 256 + lb * 64 + lane;
	bf16x8* l8 = reinterpret_cast<bf16x8*>(lds);
	const int lblk = ((lpk * G + lg) * 4 + lb) * 64 + lane;      // + 1024 per buffer, + 64 for the second block

	f32x16 acc[4][2];
#pragma unroll
	for (int b = 0; b < 4; ++b)
#pragma unroll
		for (int nb = 0; nb < 2; ++nb)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[b][nb][g] = 0.f;

	// The loop body is branch-free (hipcc then counts vmcnt exactly instead of draining the queue): K-steps past
	// the end of a piece are ZERO blocks in LDS (the loader writes zeros), and the stage count is padded to a
	// multiple of SETS.
	const int nst_pad = ((nst + SETS - 1) / SETS) * SETS;
	const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
	// prologue: stages 0 .. SETS - 1 requested, stages 0 and 1 parked in LDS, stage SETS requested
	bf16x8 st[SETS][2];
#pragma unroll
	for (int q = 0; q < SETS; ++q) {
		int k = q * G + lg;
		k = k < llast ? k : llast;
		st[q][0] = lap[(long)k * 256];
		st[q][1] = lap[(long)k * 256 + 64];
	}
	const bf16x8* fp = F + (long)s0 * fstep + (long)cg * 128 + lane;
	const int flast = steps > 0 ? steps - 1 : 0;
	bf16x8 fb[2][G][2];
#pragma unroll
	for (int q = 0; q < 2; ++q)
#pragma unroll
		for (int g = 0; g < G; ++g) {
			int k = q * G + g;
			k = k < flast ? k : flast;
			fb[q][g][0] = fp[(long)k * fstep];
			fb[q][g][1] = fp[(long)k * fstep + 64];
		}
	{
		const bool v0 = lg < lsteps, v1 = G + lg < lsteps;
		l8[lblk] = v0 ? st[0][0] : zero;
		l8[lblk + 64] = v0 ? st[0][1] : zero;
		l8[1024 + lblk] = v1 ? st[1][0] : zero;
		l8[1024 + lblk + 64] = v1 ? st[1][1] : zero;
		int k = SETS * G + lg;
		k = k < llast ? k : llast;
		st[0][0] = lap[(long)k * 256];
		st[0][1] = lap[(long)k * 256 + 64];
	}
	__syncthreads();
	const bf16x8* rbase = l8 + (kp * G) * 256 + lane;      // this wave's piece inside a buffer
	// operands of K-step j live in va[j & 1]; the reads of K-step j + 1 are issued ahead of the MFMAs of K-step j
	bf16x8 va[2][4];
#pragma unroll
	for (int b = 0; b < 4; ++b) va[0][b] = rbase[b * 64];
	__builtin_amdgcn_sched_barrier(0);

	int buf = 0;      // LDS buffer of stage t (three buffers)
	for (int t0 = 0; t0 < nst_pad; t0 += SETS) {
#pragma unroll
		for (int u = 0; u < SETS; ++u) {
			const int t = t0 + u;
			const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
			__syncthreads();      // stage t + 1 visible; buffer of stage t + 2 (= stage t - 1) free
			// stage t + 2 (set (u + 2) % SETS) -> LDS
			{
				const bool valid = (t + 2) * G + lg < lsteps;
				bf16x8* dst = l8 + buf2 * 1024 + lblk;
				dst[0] = valid ? st[(u + 2) % SETS][0] : zero;
				dst[64] = valid ? st[(u + 2) % SETS][1] : zero;
			}
			// loads of stage t + SETS + 1 -> set (u + 1) % SETS (stage t + 1 left it for LDS one iteration ago)
			{
				int k = (t + SETS + 1) * G + lg;
				k = k < llast ? k : llast;
				st[(u + 1) % SETS][0] = lap[(long)k * 256];
				st[(u + 1) % SETS][1] = lap[(long)k * 256 + 64];
			}
			// K-step 2t: read 2t + 1 (same stage), multiply 2t
#pragma unroll
			for (int b = 0; b < 4; ++b) va[1][b] = rbase[buf * 1024 + (4 + b) * 64];
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int b = 0; b < 4; ++b)
#pragma unroll
				for (int nb = 0; nb < 2; ++nb)
					acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[0][b], fb[u & 1][0][nb], acc[b][nb], 0, 0, 0);
			__builtin_amdgcn_sched_barrier(0);
			// K-step 2t + 1: read 2t + 2 (first K-step of stage t + 1), multiply 2t + 1
#pragma unroll
			for (int b = 0; b < 4; ++b) va[0][b] = rbase[buf1 * 1024 + b * 64];
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int b = 0; b < 4; ++b)
#pragma unroll
				for (int nb = 0; nb < 2; ++nb)
					acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[1][b], fb[u & 1][1][nb], acc[b][nb], 0, 0, 0);
			// F of stage t + 2 -> the ring slot just consumed
#pragma unroll
			for (int g = 0; g < G; ++g) {
				int k = (t + 2) * G + g;
				k = k < flast ? k : flast;
				fb[u & 1][g][0] = fp[(long)k * fstep];
				fb[u & 1][g][1] = fp[(long)k * fstep + 64];
			}
			buf = buf1;
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	__syncthreads();      // the staging buffers become the epilogue's exchange area

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
			const int sl = wave * 2 * CH + i;
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

// ---- 256 panel columns per pass, round 2 ----------------------------------------------------------------------------------
// The staged kernel above keeps eight waves at 256 registers (4 column chunks x 2 K pieces of a 128-row tile) and pays a
// barrier per two K-steps: 182 us per launch at config 4's shard against 100 us of HBM time and 83 us of matrix-pipe time,
// MFMA pipe busy 45 % of the waves' life (profiles/r01_pmc_factor_product_bf16_f64.md).  Here:
//   * workgroup = 4 waves, ONE per SIMD, = NRB = 7 row blocks of 32 (224 rows) x 256 columns x one K slice; wave w owns the
//     64 columns 64 w .. 64 w + 63 of all seven blocks: 14 accumulator tiles = 224 AGPRs, 14 MFMAs per K-step.  Seven
//     blocks, not four: the factor fragments of a K-step (8 KiB) are fetched once per 7 KiB of A instead of once per 4 KiB,
//     and config 4's 1 563 row blocks make 224 workgroups -- one round on 256 CUs -- instead of 391 in two rounds;
//   * A: every wave loads two of the tile's blocks per K-step into a D-deep register ring, parks a landed step in a
//     three-slot LDS ring (8 ds_write_b128 per step and workgroup), all four waves read their operands from there;
//   * F: a wave's two factor blocks go straight into its own D-deep register ring (no LDS at all);
//   * one barrier per K-step; the loop body is branch-free (hipcc then counts vmcnt exactly): K-steps past the end of the
//     slice load a block of zeros; every memory instruction sits BETWEEN two MFMAs (sched_group_barrier): with the groups in
//     a row the same kernel ran 188 us, interleaved 159, with scalar-base addressing and no per-step selects 155.
// Tried first and dropped: both operands by LDS-DMA (global_load_lds_dwordx4 into an 11 + 10 slot ring, 136 - 157 KiB of LDS,
// counted vmcnt per loader wave, raw s_barrier): 165 us, and 156 us with the MFMAs removed -- 15 KiB per K-step and CU through
// the DMA path top out near 37 GB/s per CU whatever the depth of the request ring (4 / 8 / 9 steps).
// Tried last and dropped: the same tile with EIGHT waves (two per SIMD, 128 accumulator registers each: columns by wave & 3, row blocks
// 0..3 / 3..6 by wave >> 2, one A load and two F loads per wave and K-step so that a wave keeps 16 - 20 loads in flight, the number a
// streaming probe tolerates): 183 - 194 us for ring depths (4,4) .. (12,4) against this kernel's 162 on the same box.
constexpr int BFD_NRB = 7;            // row blocks per workgroup
#ifndef BFD_R2_D
#define BFD_R2_D 8                    // register-ring depth of the A operand (4 .. 12 measure the same)
#endif
#ifndef BFD_R2_DF
#define BFD_R2_DF 8                   // ... of the factor fragments (divides BFD_R2_D): 4 -> 166 us, 3 -> 184, 2 -> 274 (their slice of the image streams through L2 once)
#endif
__device__ bf16x8 g_bf_zero_block[64];      // one all-zero fragment block: the A operand of K-steps past the end of a slice
#ifdef NMFAMD_DIAG_BUILD
// measurement build: per-wave life stamps of the NEXT launch of the round-2 kernel (100 MHz ticks: entry, loop start, loop end, exit; [block][wave][4]) -- tools/stamp_bf16.py
static thread_local unsigned long long* t_bf_stamps = nullptr;
void set_factor_product_bf16_stamps(unsigned long long* stamps) { t_bf_stamps = stamps; }
#define BF_STAMP_ARG , unsigned long long* __restrict__ stamps
#define BF_STAMP(i) do { if (stamps != nullptr && lane == 0) stamps[4 * ((long)blockIdx.x * 4 + wave) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BF_STAMP_ARG
#define BF_STAMP(i) do { } while (0)
#endif
// VAR (measurement builds, NMFAMD_BF_VARIANT; results void, timing only -- profiles/r06_c4_loop.md): the SHIPPED loop with parts taken out --
//   1 no MFMAs (loads, LDS ring and barriers as they are)   2 only the streamed operand's path (A loads, park, reads; no F loads, no MFMAs)
//   3 only the factor fragments' loads (no A loads, no LDS traffic, no MFMAs)   4 the whole loop, no slab stores   5 MFMAs and LDS reads only (no global loads in the loop)
//   6 only the A loads (no LDS ring, no barrier, no F loads, no MFMAs)   7 A and F loads, nothing else
//   8 / 9 (results valid) the slab epilogue as whole 1 KB rows through LDS / with non-temporal stores
template <int NRB, int D, int DF, int VAR = 0>
__global__ __launch_bounds__(256, 1) void k_factor_product_bf16_r2(
	const bf16x8* __restrict__ A, long tile_frags, int total_blocks, const bf16x8* __restrict__ F, int NBT,
	float* __restrict__ slabs, long slab_stride, int RP, int steps_total, int splits, int tiles, GramReduceArgs rg BF_STAMP_ARG) {
	static_assert(D % 2 == 0 && D % DF == 0 && DF >= 2 && NRB <= 8, "ring depth even (two operand sets) and a multiple of the factor ring's, at most eight row blocks");
#ifndef BFD_PAIR
#define BFD_PAIR 0
#endif
	// BFD_PAIR = 1: one barrier per TWO K-steps -- a step is parked four steps ahead into a six-slot ring (the pair being read, the pair
	// that is visible and read next, the pair being written).  Measured the same as one barrier per step (156 us either way), as did
	// ring depths 4 .. 12: with the MFMAs removed the kernel takes 155 us, with A served from cache instead of HBM 125 us, with F
	// from cache 158 us (profiles/r02_c4_kernel_experiments.md) -- the loop is bound by its own memory skeleton, not by the barrier.
	// Neither by the image layout: with the image K-step-major ([K-step][tile][block][lane]: what all workgroups read at one time is ONE
	// contiguous window instead of one stream per tile) 162 us against 164 tile-major, both with a run-time K-step stride (which costs
	// the scalar-base form of the loads: 155 -> 163).
	constexpr int AHEAD = BFD_PAIR ? 4 : 2, SLOTS = BFD_PAIR ? 6 : 3;
	__shared__ __attribute__((aligned(16))) bf16x8 l8[SLOTS * 512];      // [slot][block 0..7][lane]
	const int nblk = tiles * splits;
	if (blockIdx.x >= (unsigned)nblk) {
		// passengers behind the product's workgroups: the Gram matrix of a factor panel, one tile per workgroup (tri_gram_tile.h)
		static_assert(sizeof(l8) >= TRI_RIDE_LDS_BYTES, "the passengers' LDS overlays the product's ring");
#ifdef NMFAMD_DIAG_BUILD
		{ const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63; BF_STAMP(0); }
#endif
		if (blockIdx.y == 0 && rg.tri_frags != nullptr)
			tri_gram_passenger(reinterpret_cast<const bf16x8*>(rg.tri_frags), rg.tri_ks, (int)blockIdx.x - nblk, rg.tri_partial, rg.tri_counters, rg.G,
			                   reinterpret_cast<bf16x8*>(rg.tri_x3), rg.tri_diag, l8);
#ifdef NMFAMD_DIAG_BUILD
		{ const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BF_STAMP(3); }
#endif
		return;
	}
	int vb = blockIdx.x;
	{
		const int q8 = nblk / 8, r8 = nblk % 8, xcd = vb % 8, idx = vb / 8;      // XCD-aware placement, as in kernels_x3.hip
		vb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
	}
	const int t = vb % tiles, sp = vb / tiles, grp = blockIdx.y;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int half = lane >> 5, l31 = lane & 31;
	BF_STAMP(0);
	const int gb0 = t * NRB;
	const int b0 = (int)(((long)steps_total * sp) / splits);
	const int b1 = (int)(((long)steps_total * (sp + 1)) / splits);
	const int n = b1 - b0;
	const int last = n > 0 ? n - 1 : 0;
	const long fstep = (long)NBT * 64;

	// loader role: blocks 2 wave, 2 wave + 1 of the tile (beyond NRB - 1: a repeat of the last block, parked in the slot's
	// spare eighth block), clamped to the image.  Wave-uniform bases + the lane's fixed offset: the loads take the scalar-base
	// form and cost no address arithmetic on the vector ALU.  Steps past the end of the K slice read a block of zeros
	// (g_bf_zero_block) instead of the image, so that nothing has to be selected when a step is parked.
	const bf16x8* abase[2];
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		const int u = 2 * wave + j;
		int gb = gb0 + (u < NRB ? u : NRB - 1);
		gb = gb < total_blocks ? gb : total_blocks - 1;
		abase[j] = A + (long)(gb >> 2) * tile_frags + (long)b0 * 256 + (gb & 3) * 64;
	}
	const bf16x8* fbase = F + ((long)b0 * NBT + grp * 8 + 2 * wave) * 64;
	const bf16x8* zbase = g_bf_zero_block;
	// (the spare eighth block of a slot is nobody's operand: it is filled from the zero block -- an L1 hit -- instead of a second copy of
	//  block NRB - 1 from the image)
	const bool spare[2] = {2 * wave >= NRB, 2 * wave + 1 >= NRB};
	auto a_src = [&](int j, int k) -> const bf16x8* { return (k < n && !spare[j]) ? abase[j] + (long)k * 256 : zbase; };      // scalar select
	auto f_src = [&](int k) -> const bf16x8* { return fbase + (long)(k < last ? k : last) * fstep; };
	const int pblk = 2 * wave * 64 + lane;                    // this wave's two blocks inside a slot (+ 64 for the second)

	f32x16 acc[NRB][2];
#pragma unroll
	for (int b = 0; b < NRB; ++b)
#pragma unroll
		for (int nb = 0; nb < 2; ++nb)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[b][nb][g] = 0.f;

	bf16x8 stA[D][2], stF[DF][2];
#pragma unroll
	for (int q = 0; q < D; ++q) { stA[q][0] = a_src(0, q)[lane]; stA[q][1] = a_src(1, q)[lane]; }
#pragma unroll
	for (int q = 0; q < DF; ++q) { stF[q][0] = f_src(q)[lane]; stF[q][1] = f_src(q)[64 + lane]; }
	// steps 0 .. AHEAD - 1 parked, their ring slots reloaded with steps D ...
#pragma unroll
	for (int q = 0; q < AHEAD; ++q) {
		l8[q * 512 + pblk] = stA[q][0];
		l8[q * 512 + pblk + 64] = stA[q][1];
		stA[q][0] = a_src(0, D + q)[lane]; stA[q][1] = a_src(1, D + q)[lane];
	}
	__syncthreads();
	bf16x8 va[2][NRB];
#pragma unroll
	for (int b = 0; b < NRB; ++b) va[0][b] = l8[b * 64 + lane];
	__builtin_amdgcn_sched_barrier(0);

	const int n_pad = ((n + D - 1) / D) * D;
	BF_STAMP(1);
	int rd = 1, wr = AHEAD;                              // LDS slots of step s + 1 (to read) and s + AHEAD (to park)
	for (int t0 = 0; t0 < n_pad; t0 += D) {
#pragma unroll
		for (int u = 0; u < D; ++u) {
			const int s = t0 + u;
			if ((!BFD_PAIR || (u & 1) == 0) && VAR != 6 && VAR != 7) __syncthreads();      // step s + 1 visible; the slot of step s + AHEAD free
			// Everything below is ONE scheduling region: the fourteen MFMAs of step s, and between them (one wave per SIMD
			// overlaps nothing but its own instruction order; PMC of the first version with the groups in a row: matrix pipe
			// busy 45 % of the wave's life, 24 % issue stalls outside it) the park of step s + 2, the A loads of step
			// s + 2 + D, the operand reads of step s + 1 and the F loads of step s + D - 1 (into the ring slot step s - 1 has
			// just left).
			if (VAR != 3 && VAR != 6 && VAR != 7) {
				l8[wr * 512 + pblk] = stA[(u + AHEAD) % D][0];
				l8[wr * 512 + pblk + 64] = stA[(u + AHEAD) % D][1];
			}
#if defined(__HIP_DEVICE_COMPILE__)
			if (VAR == 6 || VAR == 7) asm volatile("" :: "v"(stA[(u + AHEAD) % D][0]), "v"(stA[(u + AHEAD) % D][1]));
#endif
			if (VAR != 3 && VAR != 5) {
				stA[(u + AHEAD) % D][0] = a_src(0, s + AHEAD + D)[lane];
				stA[(u + AHEAD) % D][1] = a_src(1, s + AHEAD + D)[lane];
			}
			if (VAR != 3 && VAR != 6 && VAR != 7) {
#pragma unroll
				for (int b = 0; b < NRB; ++b) va[(u + 1) & 1][b] = l8[rd * 512 + b * 64 + lane];
			}
			if (VAR != 2 && VAR != 5 && VAR != 6) {
				stF[(u + DF - 1) % DF][0] = f_src(s + DF - 1)[lane];
				stF[(u + DF - 1) % DF][1] = f_src(s + DF - 1)[64 + lane];
			}
			if (VAR == 0 || VAR == 4 || VAR == 5 || VAR >= 8) {
#pragma unroll
				for (int b = 0; b < NRB; ++b) {
					acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[u & 1][b], stF[u % DF][0], acc[b][0], 0, 0, 0);
					acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[u & 1][b], stF[u % DF][1], acc[b][1], 0, 0, 0);
				}
			} else {
				// (what was loaded and read stays "used": the compiler keeps the instructions)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
				for (int b = 0; b < NRB; ++b) asm volatile("" :: "v"(va[u & 1][b]));
				asm volatile("" :: "v"(stF[u % DF][0]), "v"(stF[u % DF][1]));
#endif
			}
			// MFMA, then one memory instruction, fourteen times: 2 LDS writes, 2 + 2 loads, 7 (NRB) LDS reads
#pragma unroll
			for (int i = 0; i < 2 * NRB; ++i) {
				__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // MFMA
				if (i < 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                        // DS write
				else if (i < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                   // VMEM read (A)
				else if (i < 4 + NRB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // DS read
				else if (i < 6 + NRB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);             // VMEM read (F)
			}
			rd = rd == SLOTS - 1 ? 0 : rd + 1;
			wr = wr == SLOTS - 1 ? 0 : wr + 1;
			__builtin_amdgcn_sched_barrier(0);
		}
	}

	// epilogue: C/D map of the 32 x 32 MFMA: register g of lane l is row (g & 3) + 8 (g >> 2) + 4 (l >> 5), column l & 31
	BF_STAMP(2);
	float* slab = slabs + (long)sp * slab_stride;
	if (VAR == 8) {
		// sixteen rows of the tile at a time through the (now free) LDS ring: every store instruction of a wave writes one whole row of the 256-column group (1 KB)
		float* st = reinterpret_cast<float*>(l8);
		typedef float f32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
		for (int b = 0; b < NRB; ++b) {
			const int gb = gb0 + b;
#pragma unroll
			for (int h2 = 0; h2 < 2; ++h2) {
				__syncthreads();
#pragma unroll
				for (int nb = 0; nb < 2; ++nb)
#pragma unroll
					for (int gq = 0; gq < 2; ++gq)
#pragma unroll
						for (int gl = 0; gl < 4; ++gl) {
							const int g = 4 * (2 * h2 + gq) + gl;
							st[(gl + 8 * gq + 4 * half) * 256 + 64 * wave + 32 * nb + l31] = acc[b][nb][g];
						}
				__syncthreads();
				if (gb < total_blocks) {
#pragma unroll
					for (int k = 0; k < 4; ++k) {
						const int idx = threadIdx.x + 256 * k, row = idx >> 6, c4 = idx & 63;
						*reinterpret_cast<f32x4v*>(slab + (long)(32 * gb + 16 * h2 + row) * RP + 256 * grp + 4 * c4) = *reinterpret_cast<const f32x4v*>(st + row * 256 + 4 * c4);
					}
				}
			}
		}
	} else
#pragma unroll
	for (int b = 0; b < NRB; ++b) {
		const int gb = gb0 + b;
#if defined(__HIP_DEVICE_COMPILE__)
		if (VAR == 4) { asm volatile("" :: "v"(acc[b][0]), "v"(acc[b][1])); continue; }
#endif
		if (gb < total_blocks) {
#pragma unroll
			for (int nb = 0; nb < 2; ++nb) {
				const int c = 256 * grp + 64 * wave + 32 * nb + l31;
#pragma unroll
				for (int g = 0; g < 16; ++g) {
					const int x = 32 * gb + (g & 3) + 8 * (g >> 2) + 4 * half;
					if (VAR == 9) __builtin_nontemporal_store(acc[b][nb][g], slab + (long)x * RP + c);
					else slab[(long)x * RP + c] = acc[b][nb][g];
				}
			}
		}
	}
#ifdef NMFAMD_DIAG_BUILD
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	BF_STAMP(3);
#endif
}

// (Round 4 built two more memory skeletons for this product -- the HBM stream straight into the MFMA operand registers with the factor fragments through LDS, and the
//  same with hand-written loads and counted waits: 158.5 / 157.7 us per launch against this kernel's 159.6, profiles/r04_c4_product_kernel.md.  Removed in round 5:
//  git history holds them -- k_factor_product_bf16_r3 / r3a.)

// workgroups along x and K slices of the round-2 kernel for `xtiles` 128-row tiles and KS K-steps
static void plan_bf16_dma(int xtiles, int KS, int num_cus, int* tiles, int* splits) {
	const int nrb = BFD_NRB;      // row blocks per workgroup
	*tiles = (4 * xtiles + nrb - 1) / nrb;
	const int by_fill = std::max(1, num_cus / std::max(1, *tiles));
	const int by_depth = std::max(1, KS / 48);               // at least 48 K-steps per slice: the rings are 8 deep
	*splits = std::max(1, std::min(by_fill, by_depth));
}

static hipError_t launch_fp_bf16_r2(const FactorProductPlan& p, const void* A, int KS, const void* F, int RP,
                                     float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg = nullptr) {
	// (the K slices are the caller's plan -- plan_splits_bf16 from the engine's CU count, which also sized the slabs; the workgroups
	//  along x follow from the tile count alone: no device query per launch)
	const int tiles = (4 * p.xtiles + BFD_NRB - 1) / BFD_NRB, splits = p.splits;
	if (splits < 1 || KS < 1) return hipErrorInvalidValue;
	GramReduceArgs none = {nullptr, 0, nullptr, nullptr, 0};
	const bool ride = rg != nullptr && rg->tri_frags != nullptr;
	if (ride && (RP != 256 || rg->G == nullptr || rg->tri_ks < 1 || rg->tri_partial == nullptr || rg->tri_counters == nullptr)) return hipErrorInvalidValue;
	dim3 grid(tiles * splits + (ride ? TRI_PASSENGERS : 0), RP / 256), block(256);
#ifdef NMFAMD_DIAG_BUILD
	if (const char* ve = tuning_env("NMFAMD_BF_VARIANT")) {
		const int v = std::atoi(ve);
#define NMFAMD_BF_VAR(V) case V: hipLaunchKernelGGL((k_factor_product_bf16_r2<BFD_NRB, BFD_R2_D, BFD_R2_DF, V>), grid, block, 0, stream, \
		reinterpret_cast<const bf16x8*>(A), (long)KS * 256, 4 * p.xtiles, reinterpret_cast<const bf16x8*>(F), RP / 32, slabs, slab_stride, RP, KS, splits, tiles, ride ? *rg : none, t_bf_stamps); \
		t_bf_stamps = nullptr; return hipGetLastError();
		switch (v) { NMFAMD_BF_VAR(1) NMFAMD_BF_VAR(2) NMFAMD_BF_VAR(3) NMFAMD_BF_VAR(4) NMFAMD_BF_VAR(5) NMFAMD_BF_VAR(6) NMFAMD_BF_VAR(7) NMFAMD_BF_VAR(8) NMFAMD_BF_VAR(9) default: break; }
#undef NMFAMD_BF_VAR
	}
#endif
	hipLaunchKernelGGL((k_factor_product_bf16_r2<BFD_NRB, BFD_R2_D, BFD_R2_DF>), grid, block, 0, stream,
	                   reinterpret_cast<const bf16x8*>(A), (long)KS * 256, 4 * p.xtiles, reinterpret_cast<const bf16x8*>(F), RP / 32,
	                   slabs, slab_stride, RP, KS, splits, tiles, ride ? *rg : none
#ifdef NMFAMD_DIAG_BUILD
	                   , t_bf_stamps
#endif
	                   );
#ifdef NMFAMD_DIAG_BUILD
	t_bf_stamps = nullptr;
#endif
	return hipGetLastError();
}

template <int SETS>
static hipError_t launch_fp_bf16_staged(const FactorProductPlan& p, const void* A, int KS, const void* F, int RP,
                                        float* slabs, long slab_stride, hipStream_t stream) {
	dim3 grid(p.xtiles, p.splits, RP / 256), block(512);
	const size_t lds_bytes = 8 * 4 * 4 * 64 * sizeof(f32x4);
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_factor_product_bf16_staged<SETS>), (int)lds_bytes, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL((k_factor_product_bf16_staged<SETS>), grid, block, lds_bytes, stream,
	                   reinterpret_cast<const bf16x8*>(A), (long)KS * 256, reinterpret_cast<const bf16x8*>(F), RP / 32,
	                   slabs, slab_stride, RP, KS, p.splits);
	return hipGetLastError();
}

template <int D, int CH>
static hipError_t launch_fp_bf16(const FactorProductPlan& p, const void* A, int KS, const void* F, int RP,
                                 float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg) {
	GramReduceArgs none = {nullptr, 0, nullptr, nullptr, 0};
	const bool with_reduce = rg != nullptr && rg->partials != nullptr && p.xtiles >= GRAM_REDUCE_BLOCKS;
	if (rg != nullptr && rg->partials != nullptr && (!with_reduce || CH != 1)) return hipErrorInvalidValue;
	dim3 grid(p.xtiles, p.splits + (with_reduce ? 1 : 0), RP / (64 * CH)), block(512);
	const size_t lds_bytes = 8 * 4 * 4 * 64 * sizeof(f32x4);
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_factor_product_bf16<D, CH>), (int)lds_bytes, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL((k_factor_product_bf16<D, CH>), grid, block, lds_bytes, stream,
	                   reinterpret_cast<const bf16x8*>(A), (long)KS * 256, reinterpret_cast<const bf16x8*>(F), RP / 32,
	                   slabs, slab_stride, RP, KS, p.splits, with_reduce ? *rg : none);
	return hipGetLastError();
}

// Every wave piece gets at least 8 K-steps (the ring is 4 deep); as many slices as fill the chip.
int plan_splits_bf16(int xtiles, int KS, int RP, int num_cus) {
	if (RP % 256 == 0 && tuning_env("NMFAMD_BF_STAGED") == nullptr && tuning_env("NMFAMD_BF_UNSTAGED") == nullptr) {
		int tiles = 0, splits = 0;
		plan_bf16_dma(xtiles, KS, num_cus, &tiles, &splits);
		return splits;
	}
	const int KP = RP == 64 ? 8 : (RP % 256 == 0 ? 2 : 4);
	const int by_fill = std::max(1, num_cus / std::max(1, xtiles));
	const int by_depth = std::max(1, KS / (8 * KP));
	return std::max(1, std::min(by_fill, by_depth));
}

int bf16_product_workgroups(const FactorProductPlan& p) { return ((4 * p.xtiles + BFD_NRB - 1) / BFD_NRB) * p.splits; }

// RP: padded rank of the panel (64, or a multiple of 128).  256 columns per pass over A when RP is a
// multiple of 256, else 128 (64 for RP = 64); wider panels take RP / 256 (RP / 128) passes (grid.z).
hipError_t launch_factor_product_bf16(const FactorProductPlan& p, const void* A, int KS, const void* F, int RP,
                                      float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg) {
	constexpr int D = 4;
	if (RP == 64) return launch_fp_bf16<D, 1>(p, A, KS, F, RP, slabs, slab_stride, stream, rg);
	if (RP % 256 == 0) {
		static const bool unstaged = tuning_env("NMFAMD_BF_UNSTAGED") != nullptr;      // A/B switch for measurements
		if (unstaged) return launch_fp_bf16<D, 4>(p, A, KS, F, RP, slabs, slab_stride, stream, rg);
		if (rg != nullptr && rg->partials != nullptr) return hipErrorInvalidValue;
		const bool tri_ride = rg != nullptr && rg->tri_frags != nullptr;
		static const bool staged = tuning_env("NMFAMD_BF_STAGED") != nullptr;          // A/B switch: the round-1 kernel
		if (tri_ride && staged) return hipErrorInvalidValue;      // (only the shipped kernel carries passengers)
		if (staged) return launch_fp_bf16_staged<6>(p, A, KS, F, RP, slabs, slab_stride, stream);
		return launch_fp_bf16_r2(p, A, KS, F, RP, slabs, slab_stride, stream, rg);
	}
	if (RP % 128 == 0) return launch_fp_bf16<D, 2>(p, A, KS, F, RP, slabs, slab_stride, stream, rg);
	return hipErrorInvalidValue;
}

} // namespace nmfamd
