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
	if (RP % 256 == 0) {
		static const bool unstaged = tuning_env("NMFAMD_BF_UNSTAGED") != nullptr;      // A/B switch for measurements
		if (unstaged) return launch_fp_bf16<D, 4>(p, A, KS, F, RP, slabs, slab_stride, stream, rg);
		if (rg != nullptr && rg->partials != nullptr) return hipErrorInvalidValue;
		return launch_fp_bf16_staged<6>(p, A, KS, F, RP, slabs, slab_stride, stream);
	}
	if (RP % 128 == 0) return launch_fp_bf16<D, 2>(p, A, KS, F, RP, slabs, slab_stride, stream, rg);
	return hipErrorInvalidValue;
}

} // namespace nmfamd
