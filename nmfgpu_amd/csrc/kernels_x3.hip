// kernels_x3.hip -- fp32 factor product on the bf16 matrix pipe by exact operand splitting.
//
// Every fp32 operand is cut into three bf16 terms, a = a1 + a2 + a3 EXACTLY (8 + 8 + 8 significand
// bits, round-to-nearest residuals), and the product keeps the six cross terms of order <= 2^-16:
//     a b  ~=  a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1)          |dropped| <= ~2^-23 |a b|
// Each term is exact in fp32 (8 x 8 bit significands) and accumulates in the fp32 accumulators of
// v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs cost 6/16 of the fp32 MFMA work they replace, so the product
// of a 10 000 x 5 000 fp32 matrix stops being MFMA-bound and becomes HBM-bound on the fp32 image of V.
//
// The streamed matrix is the SAME x-tiled fp32 image the fp32 kernel reads (kernels.hip): a lane's
// 16-byte load is four consecutive rows of one column, i.e. one k value of four M-blocks, and the
// eight loads of a K-step (k = 8 h + j) are exactly the eight-k bf16 operand of the 32x32x16 MFMA --
// no transposition, splitting is three conversions and two subtractions per element in registers.
// The factor panel is split once per product into fragment order (k_pack_panel_x3).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "tuning.h"
#include "split3.h"
#include "inverse_gj64.h"
#include "gram_image.h"

namespace nmfamd {

#ifndef NMFAMD_XCD_REMAP
#define NMFAMD_XCD_REMAP 1
#endif
constexpr bool XCD_REMAP = NMFAMD_XCD_REMAP != 0;

// Ring depth of the rank-64 instantiations: K-steps in flight per wave, 8 loads of V + 6 of factor fragments each.  Fewer is faster (a wave
// with more than ~20 loads in flight loses HBM bandwidth, profiles/r02_c4_kernel_experiments.md): config 2, us per iteration, depth of
// the x-tiled / y-tiled form: 3/3 98.1, 2/2 96.1, 4/4 99.9, 2/1 see DESIGN.  Depths 1 and 2 deal the K-steps out in the same units,
// so all of them give the same bits.
#ifndef X3_RING_X
#define X3_RING_X 2                   // K-steps in flight per wave (8 + 6 loads each), x-tiled form
#endif
#ifndef X3_RING_Y
#define X3_RING_Y 2                   // ... y-tiled form (its landing ring; two more steps sit in the LDS slots)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Factor fragments: Fx[(((ks * NBT + nb) * 3 + plane) * 2 + h) * 32 + r][8] = plane of F(c = 32 nb + r, y = 16 ks + 8 h + j)
__global__ __launch_bounds__(256) void k_pack_panel_x3(const float* __restrict__ P, int RP, int NBT, int len, bf16x8* __restrict__ dst, long frags) {
	const long f = (long)blockIdx.x * 256 + threadIdx.x;   // one (ks, nb, h, r) per thread, three 16-byte fragments out
	if (f >= frags) return;
	const int r = (int)(f & 31), h = (int)((f >> 5) & 1);
	const long t = f >> 6;
	const int nb = (int)(t % NBT);
	const long ks = t / NBT;
	float v[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const long y = 16 * ks + 8 * h + j;
		v[j] = y < len ? P[y * RP + 32 * nb + r] : 0.f;
	}
	store_split3(dst, ks, NBT, nb, h, r, v);
}

// KS K-steps cover len; the image has KS + 1: the last one is all zero (see k_factor_product_x3).
hipError_t launch_pack_panel_x3(const float* P, int RP, int len, void* dst, int KS, hipStream_t stream) {
	const int NBT = RP / 32;
	const long frags = (long)(KS + 1) * NBT * 64;
	hipLaunchKernelGGL(k_pack_panel_x3, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, stream, P, RP, NBT, len, reinterpret_cast<bf16x8*>(dst), frags);
	return hipGetLastError();
}

// Passenger Gram reduction (same contract as in kernels.hip / kernels_bf16.hip) for a block of NT threads.
// The block is latency-bound (every load is a miss under the product's HBM stream), so the work is cut for
// few round trips: NT / 64 groups each take a contiguous range of the partial matrices, a lane owns four
// consecutive elements (16-byte loads, eight in flight), groups are added in ascending order.
template <int NT>
__device__ inline void gram_reduce_block_x3(const GramReduceArgs& rg, int blk, float* lds) {
	constexpr int NG = NT / 64;
	const int tid = threadIdx.x;
	const int g = tid >> 6, l = tid & 63;
	float* s_scale = lds;                     // 64
	float* s_tmp = lds + 64;                  // NG * 256
	const int parts = rg.parts;
	const int p0 = (parts * g) / NG, p1 = (parts * (g + 1)) / NG;
	if (rg.normalize) {
		// diagonal of the full sum: column l, this group's range of parts
		const float* dp = rg.partials + l * 65;
		float sum = 0.f;
		int p = p0;
		for (; p + 8 <= p1; p += 8) {
			float v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = dp[(long)(p + u) * 4096];
#pragma unroll
			for (int u = 0; u < 8; ++u) sum += v[u];
		}
		for (; p < p1; ++p) sum += dp[(long)p * 4096];
		s_tmp[g * 64 + l] = sum;
		__syncthreads();
		if (tid < 64) {
			float d = s_tmp[tid];
#pragma unroll
			for (int k = 1; k < NG; ++k) d += s_tmp[k * 64 + tid];
			s_scale[tid] = d > 0.f ? 1.0f / sqrtf(d) : 1.0f;
		}
	} else if (tid < 64) {
		s_scale[tid] = 1.0f;
	}
	__syncthreads();
	{
		const float* ep = rg.partials + blk * 256 + 4 * l;
		f32x4 sum = {0.f, 0.f, 0.f, 0.f};
		int p = p0;
		for (; p + 8 <= p1; p += 8) {
			f32x4 v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(ep + (long)(p + u) * 4096);
#pragma unroll
			for (int u = 0; u < 8; ++u) sum += v[u];
		}
		for (; p < p1; ++p) sum += *reinterpret_cast<const f32x4*>(ep + (long)p * 4096);
		*reinterpret_cast<f32x4*>(s_tmp + g * 256 + 4 * l) = sum;
	}
	__syncthreads();
	if (tid < 256) {
		const int e = blk * 256 + tid;
		float v = s_tmp[tid];
#pragma unroll
		for (int k = 1; k < NG; ++k) v += s_tmp[k * 256 + tid];
		rg.G[e] = (v * s_scale[e & 63]) * s_scale[e >> 6];
	}
	if (blk == 0 && tid < 64 && rg.scale) rg.scale[tid] = s_scale[tid];
}

// Workgroup = X3_WAVES waves (four: one per SIMD, 512 registers each) = one 128-row x-tile times one
// slice of the reduction range times one 64-column chunk of the panel (grid.y), the slice cut into
// wave pieces of K-steps (16 y each).  A wave keeps the 128 x 64 accumulator block (8 tiles of 32 x 32),
// streams its piece of the fp32 tile through a D-deep ring of K-steps (8 x 16 B per lane and step) and
// the matching factor fragments (6 x 16 B) through a ring of the same depth.  Row of M-block b held by
// lane l: 4 (l & 31) + b (the fp32 image interleaves four rows per lane).  Epilogue: the pieces are
// summed through LDS in piece order; one fp32 slab per slice.
// DIAG (measurement builds only, NMFAMD_X3_VARIANT 10..12): 1 = no ring refill (issue rate of the split + MFMA
// stream alone), 2 = refill A only, 3 = refill F only, 4 = the production loop; all of them stamp the main loop
// (shader cycles, 100 MHz ticks, K-steps per wave).  The production instantiation has DIAG = 0.
// R32 (y-tiled form on 16-row tiles, loads straight to registers): MFMA row r of M-block b is tile row 32 b + r (as in the YLDS form) instead of 4 r + b: consecutive lanes read
// consecutive 64-byte column chunks -- 16 cache lines per wave instruction instead of 32
template <int D, int X3_WAVES, int DIAG = 0, int NBW = 2, bool TR = false, int IMG = 128, bool YLDS = false, bool R32 = false>
__global__ __launch_bounds__(64 * X3_WAVES, NBW == 1 ? 2 : 1) void k_factor_product_x3(
	const float* __restrict__ A, long tile_stride,
	const bf16x8* __restrict__ F, int NBT,              // NBT = RP / 32 column blocks per K-step
	float* __restrict__ slabs, long slab_stride, int RP,
	int steps_total, int xtiles, int splits, GramReduceArgs rg, unsigned long long* __restrict__ stamps) {
	constexpr int TH = 128;
	unsigned long long t_loop0 = 0, t_loop1 = 0, r_loop0 = 0, r_loop1 = 0, r_entry = 0, r_tail = 0;
	if (DIAG != 0) r_entry = __builtin_amdgcn_s_memrealtime();
	extern __shared__ __attribute__((aligned(16))) float lds[];
	// grid.x = xtiles * splits product blocks (x-tile fastest) followed by the GRAM_REDUCE_BLOCKS passenger blocks,
	// and no more: every block of this kernel claims a whole CU (512 registers per lane), so an idle block would
	// still wait for a CU to drain and be launched there before the kernel can end
	// NBW = 1 (padded rank 64 only): the two 32-column chunks of an x-tile are NEIGHBOURS in grid.x (same XCD: the tile of A is fetched into one L2) and two workgroups share
	// a CU (180 registers per lane): two waves per SIMD, one splitting operands or waiting for memory while the other feeds the matrix pipe
	constexpr bool FOLD = NBW == 1;
	const int pblocks = xtiles * splits * (FOLD ? 2 : 1);
	if (blockIdx.x >= (unsigned)pblocks) {
		if (blockIdx.y != 0) return;                            // (one set of passengers, whatever the number of column chunks)
		// the 64 x 64 inverse of the least-squares algorithms rides as ONE block right behind the product's last one
		if (rg.inv_a != nullptr) inverse_gj64_body<float, X3_WAVES>(rg.inv_a, 64, rg.inv_r, rg.inv_out, rg.inv_offdiag, rg.inv_diag);
		else if (rg.image != nullptr) { if (X3_WAVES == 4) gram_image_block(rg, blockIdx.x - pblocks, lds); }
		else gram_reduce_block_x3<64 * X3_WAVES>(rg, blockIdx.x - pblocks, lds);
		return;
	}
	// XCD-aware placement (speed only): blocks b and b + 8 share an XCD and its L2; each XCD takes a CONTIGUOUS range of
	// (slice, x-tile) pairs, so the workgroups that stream the same factor fragments -- one K slice -- sit on one or two XCDs
	// and that slice of the factor image is fetched into one or two L2s instead of all eight
	int vb = blockIdx.x;
	if (XCD_REMAP) {
		const int q8 = pblocks / 8, r8 = pblocks % 8, xcd = vb % 8, idx = vb / 8;
		vb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
	}
	const int chunk = FOLD ? (vb & 1) : (int)blockIdx.y;
	if (FOLD) vb >>= 1;
	const int xt = vb % xtiles, sp = vb / xtiles;
	const int coff = 32 * NBW * chunk;
	const long fstep = (long)NBT * 192;                 // factor fragments per K-step
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const int nw = splits * X3_WAVES;
	const int widx = sp * X3_WAVES + wave;
	// The reduction range is dealt out in units of D K-steps, so that every wave runs whole turns of the ring and
	// the loop needs no tail.  Steps past the end of the range (at most D - 1, in the last unit) re-read the last
	// valid step of A against the all-zero K-step that closes the factor image (index steps_total).
	// (the unit is 2 for ring depths 1 and 2, so that every instantiation with such a ring deals the K-steps out the same way and
	//  the forms stay bit-identical to each other whatever their depth)
	constexpr int DU = D <= 2 ? 2 : D;
	const int units = (steps_total + DU - 1) / DU;
	const int s0 = DU * (int)(((long)units * widx) / nw);
	const int s1 = DU * (int)(((long)units * (widx + 1)) / nw);
	const int steps = s1 - s0;

	f32x16 acc[4][NBW];
#pragma unroll
	for (int b = 0; b < 4; ++b)
#pragma unroll
		for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[b][nb][g] = 0.f;

	if (steps > 0) {
		// TR = false: A is tiled along x (the output index): tile xt, K-step s, k = 8 half + j: ap + (16 s + j) TH, and
		//   the four rows 4 l31 + b of an M-block quartet are one 16-byte load  (va[..][j] = rows b = 0..3 at k = 8 half + j).
		// TR = true: A is tiled along y (the reduction index) -- the image the OTHER product streams along x: tile s / 8,
		//   row (128 xt + 4 l31 + b) of that tile holds the tile's 128 y contiguously, so the 16-byte load is four
		//   consecutive k of ONE row  (va[..][2 b + q] = row b at k = 8 half + 4 q .. + 3).  Same bytes per instruction, but
		//   64 rows = 32 cache lines per wave instruction instead of 8 (each line serves two K-steps).
		// IMG = tile height of the image (rows of the tiled index per tile).  128: as described above.  16 (one resident
		//   image, see Engine): a tile is 16 rows, so that BOTH forms stream contiguous memory -- x-tiled: the 128 output rows
		//   are eight tiles, a lane quartet (l31 >> 2) per tile, 1 KiB contiguous per tile and K-step; y-tiled: a K-step IS a
		//   tile, the 16 k of a row are 64 contiguous bytes and the 128 rows of the wave 8 KiB contiguous (with 128-row tiles
		//   a K-step takes 64 bytes out of each of 128 rows 512 bytes apart and every row is revisited eight times).
		const float* ap = IMG == 16 ? (TR ? A + ((long)xt * TH + (R32 ? l31 : 4 * l31)) * 16 + 8 * half
		                                  : A + ((long)xt * 8 + (l31 >> 2)) * tile_stride + (8 * half) * 16 + 4 * (l31 & 3))
		                            : (TR ? A + ((long)xt * TH + 4 * l31) * TH + 8 * half
		                                  : A + (long)xt * tile_stride + (8 * half) * TH + 4 * l31);
		auto a_addr = [&](int step, int i) -> const float* {
			if (IMG == 16) {
				if (TR) return ap + (long)step * tile_stride + (i >> 1) * (R32 ? 32 * 16 : 16) + 4 * (i & 1);
				return ap + ((long)step * 16 + i) * 16;
			}
			if (TR) return ap + (long)(step >> 3) * tile_stride + (step & 7) * 16 + (i >> 1) * TH + 4 * (i & 1);
			return ap + ((long)step * 16 + i) * TH;
		};
		// YLDS (y-tiled form on 16-row tiles): the K-step -- 128 rows x 64 B, 8 KiB contiguous -- is loaded lane-linear
		//   (instruction i, lane l: bytes 1024 i + 16 l = row 16 i + (l >> 2), chunk l & 3: the access pattern of the x-tiled
		//   form) into the landing ring va[], parked in one of two wave-private LDS slots as [row][16 + 4] floats, and every
		//   lane reads its own row back (MFMA row r of M-block b = tile row 32 b + r; row stride 80 B: conflict-free
		//   ds_read_b128).
		const float* gp = A + (long)xt * TH * 16 + 4 * lane;
		auto g_addr = [&](int step, int i) -> const float* { return gp + (long)step * tile_stride + i * 256; };
		float* lw = lds + wave * (2 * 128 * 20);                                           // two slots of 128 x 20 floats
		const int wofs = (lane >> 2) * 20 + 4 * (lane & 3);                                // + i * 320, + slot * 2560
		const int rofs = l31 * 20 + 8 * half;                                              // + b * 640, + slot * 2560
		const bf16x8* fp = F + (long)chunk * (NBW * 192) + lane;                      // + step * fstep + (nb * 3 + plane) * 64
		const int last = s1 - 1, kend = steps_total - 1;
		f32x4 va[D][8];
		bf16x8 fb[D][NBW][3];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int st = s0 + d;                                  // steps >= D here
			const int sa = st < kend ? st : kend, sf = st <= kend ? st : steps_total;
#pragma unroll
			for (int j = 0; j < 8; ++j) va[d][j] = *reinterpret_cast<const f32x4*>(YLDS ? g_addr(sa, j) : a_addr(sa, j));
#pragma unroll
			for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
				for (int pl = 0; pl < 3; ++pl) fb[d][nb][pl] = fp[(long)sf * fstep + (nb * 3 + pl) * 64];
		}
		__builtin_amdgcn_sched_barrier(0);
		// One wave per SIMD: nothing but this wave's own instruction order overlaps the operand splitting
		// (VALU) with the matrix pipe.  The body is a chain of phases (K-step d, M-block b): the twelve
		// MFMAs of the phase are interleaved four-to-one with the 44 VALU instructions that split the
		// NEXT phase's operand, and in the last phase of a step with the refill of the step's ring slot.
		// Branch-free: the ring is refilled with clamped step indices.
		if (DIAG != 0) { t_loop0 = __builtin_amdgcn_s_memtime(); r_loop0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
		bf16x8 op[2][3];
		f32x4 raw[2][2];
		if (YLDS) {
			// steps 0 and 1 go to LDS slots 0 and 1; their landing registers take steps D and D + 1
#pragma unroll
			for (int d = 0; d < 2; ++d) {
#pragma unroll
				for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(lw + d * 2560 + j * 320 + wofs) = va[d % D][j];
				int st = s0 + D + d;
				st = st < last ? st : last;
				const int sa = st < kend ? st : kend;
#pragma unroll
				for (int j = 0; j < 8; ++j) va[d % D][j] = *reinterpret_cast<const f32x4*>(g_addr(sa, j));
			}
			// LDS reads run two phases ahead of the MFMAs that use them: raw[q & 1] holds the operand of phase q, read during
			// phase q - 2 and split during phase q - 1 (one wave per SIMD: a read consumed in the phase that issues it would
			// expose the LDS latency four times per K-step)
			raw[0][0] = *reinterpret_cast<const f32x4*>(lw + rofs); raw[0][1] = *reinterpret_cast<const f32x4*>(lw + rofs + 4);
			raw[1][0] = *reinterpret_cast<const f32x4*>(lw + 640 + rofs); raw[1][1] = *reinterpret_cast<const f32x4*>(lw + 640 + rofs + 4);
			float v[8];
#pragma unroll
			for (int j = 0; j < 4; ++j) { v[j] = raw[0][0][j]; v[4 + j] = raw[0][1][j]; }
			split3(v, op[0][0], op[0][1], op[0][2]);
		} else {
			float v[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) v[j] = TR ? va[0][j >> 2][j & 3] : va[0][j][0];
			split3(v, op[0][0], op[0][1], op[0][2]);
		}
		int t = 0;
		for (; t < steps; t += D) {
#pragma unroll
			for (int d = 0; d < D; ++d) {
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const int cur = (d * 4 + b) & 1, nxt = cur ^ 1;
					const int nd = b == 3 ? (d + 1) % D : d, nbk = (b + 1) & 3;
					if (YLDS) {
						// split the operand of the next phase (read from LDS one phase ago), then read the one after it
						float v[8];
#pragma unroll
						for (int j = 0; j < 4; ++j) { v[j] = raw[nxt][0][j]; v[4 + j] = raw[nxt][1][j]; }
						split3(v, op[nxt][0], op[nxt][1], op[nxt][2]);
						const int b2 = (b + 2) & 3;
						const int slot = (t + d + (b >= 2 ? 1 : 0)) & 1;            // K-step of phase p + 2: this one or the next
						const float* rp = lw + slot * 2560 + b2 * 640 + rofs;
						raw[cur][0] = *reinterpret_cast<const f32x4*>(rp); raw[cur][1] = *reinterpret_cast<const f32x4*>(rp + 4);
					} else {
						float v[8];
#pragma unroll
						for (int j = 0; j < 8; ++j) v[j] = TR ? va[nd][2 * nbk + (j >> 2)][j & 3] : va[nd][j][nbk];
						split3(v, op[nxt][0], op[nxt][1], op[nxt][2]);
					}
#pragma unroll
					for (int nb = 0; nb < NBW; ++nb) {
						// smallest terms first
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][2], fb[d][nb][0], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][0], fb[d][nb][2], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][1], fb[d][nb][1], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][1], fb[d][nb][0], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][0], fb[d][nb][1], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][0], fb[d][nb][0], acc[b][nb], 0, 0, 0);
					}
					if (b == 3) {
						int st = s0 + t + D + d;
						st = st < last ? st : last;                         // past this wave's piece: a harmless re-read
						const int sa = st < kend ? st : kend, sf = st <= kend ? st : steps_total;
						if (YLDS) {
							// K-step t + d is done with its LDS slot: park step t + d + 2 there (it landed in ring slot (d + 2) % D,
							// loaded D steps ago) and send that ring slot for step t + d + 2 + D
							const int rs = (d + 2) % D;
							const int slot = (t + d) & 1;
#pragma unroll
							for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(lw + slot * 2560 + j * 320 + wofs) = va[rs][j];
							int s2 = s0 + t + d + 2 + D;
							s2 = s2 < last ? s2 : last;
							const int sa2 = s2 < kend ? s2 : kend;
#pragma unroll
							for (int j = 0; j < 8; ++j) va[rs][j] = *reinterpret_cast<const f32x4*>(g_addr(sa2, j));
						} else if (DIAG == 0 || DIAG == 2 || DIAG == 4) {
#pragma unroll
							for (int j = 0; j < 8; ++j) va[d][j] = *reinterpret_cast<const f32x4*>(a_addr(sa, j));
						} else {
#pragma unroll
							for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(va[d][j]));
						}
						if (DIAG == 0 || DIAG == 3 || DIAG == 4) {
#pragma unroll
							for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
								for (int pl = 0; pl < 3; ++pl) fb[d][nb][pl] = fp[(long)sf * fstep + (nb * 3 + pl) * 64];
						} else {
#pragma unroll
							for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
								for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(fb[d][nb][pl]));
						}
						// 6 NBW MFMAs against 44 VALU (+ 8 + 3 NBW loads in the last phase of a step)
						constexpr int NM = 6 * NBW, VPM = (44 + NM - 1) / NM, NL = 8 + 3 * NBW;
#pragma unroll
						for (int g = 0; g < NM; ++g) {
							__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
							__builtin_amdgcn_sched_group_barrier(0x002, VPM, 0); // VALU
							if (g < NL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
						}
						if (NL > NM) __builtin_amdgcn_sched_group_barrier(0x020, NL - NM, 0);
					} else {
						constexpr int NM = 6 * NBW, VPM = (44 + NM - 1) / NM;
#pragma unroll
						for (int g = 0; g < NM; ++g) {
							__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
							__builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
			}
		}
		if (DIAG != 0) { __builtin_amdgcn_sched_barrier(0); t_loop1 = __builtin_amdgcn_s_memtime(); r_loop1 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
	}

	if (DIAG != 0) { __builtin_amdgcn_sched_barrier(0); r_tail = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
	// in-workgroup sum through LDS, two M-blocks (four tiles) per round; C/D map: register g of lane l
	// is MFMA row i = (g & 3) + 8 (g >> 2) + 4 (l >> 5), column l & 31; output row x = 128 xt + 4 i + b.
	if (YLDS) __syncthreads();            // the epilogue image overlays the staging slots of every wave
	f32x4* l4 = reinterpret_cast<f32x4*>(lds);
	float* slab = slabs + (long)sp * slab_stride;
#pragma unroll
	for (int rd = 0; rd < NBW; ++rd) {        // four accumulator tiles per round
		if (rd > 0) __syncthreads();
#pragma unroll
		for (int tl = 0; tl < 4; ++tl) {
			const int b = (4 * rd + tl) / NBW, nb = (4 * rd + tl) % NBW;
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
		for (int i = 0; i < 16 / X3_WAVES; ++i) {
			const int sl = wave * (16 / X3_WAVES) + i;  // slice = (tile, q)
			const int q = sl & 3, tl = sl >> 2;
			const int b = (4 * rd + tl) / NBW, nb = (4 * rd + tl) % NBW;
			f32x4 s = l4[((0 * 4 + tl) * 4 + q) * 64 + lane];
#pragma unroll
			for (int p = 1; p < X3_WAVES; ++p) s += l4[((p * 4 + tl) * 4 + q) * 64 + lane];
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) {
				const int mi = gi + 8 * q + 4 * half;
				const int x = xt * TH + ((YLDS || R32) ? 32 * b + mi : 4 * mi + b);
				slab[(long)x * RP + coff + 32 * nb + l31] = s[gi];
			}
		}
	}
	if (DIAG != 0 && stamps != nullptr) {
		__builtin_amdgcn_s_waitcnt(0);
		const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
		if (lane == 0) {
			// shader cycles and 100 MHz ticks in the main loop, K-steps run there; then 100 MHz stamps of the wave's life
			unsigned long long* o = stamps + 8 * ((long)blockIdx.x * X3_WAVES + wave);
			o[0] = t_loop1 - t_loop0; o[1] = r_loop1 - r_loop0; o[2] = (unsigned long long)steps;
			o[3] = r_entry; o[4] = r_loop0; o[5] = r_loop1; o[6] = r_tail; o[7] = r_end;
		}
	}
}

// Written-out loads and counted waits of k_factor_product_x3s (see the comment at its loop).  A load's destination counts as written when the statement ends, for
// hipcc; the data lands later: every consumer reads the OUTPUT of a wait statement that names the register.
template <bool TR, int IMG>
constexpr int x3s_a_off(int i) {       // byte offset of load i (0..7) of a group of four row blocks, from the group's address
	return IMG == 16 ? (TR ? ((i >> 1) * 256 + 4 * (i & 1)) * 4 : i * 64) : i * 128 * 4;
}
constexpr int x3s_f_off(int cb, int pl) { return (pl * 64 + 16 * (cb & 1)) * 16; }     // fragment (column block cb, plane pl) from the address of its 32-column block
template <int OFF> __device__ inline void x3s_load_v(f32x4& dst, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:%c2" : "=v"(dst) : "v"(p), "i"(OFF) : "memory"); }
template <int OFF> __device__ inline void x3s_load_v_tied(f32x4& dst, const float* p, bf16x8& tie) {
	asm volatile("global_load_dwordx4 %0, %2, off offset:%c3" : "=v"(dst), "+v"(tie) : "v"(p), "i"(OFF) : "memory");
}
template <int OFF> __device__ inline void x3s_load_a(u32x4& dst, const bf16x8* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:%c2" : "=a"(dst) : "v"(p), "i"(OFF) : "memory"); }
template <int OFF> __device__ inline void x3s_load_a_tied(u32x4& dst, const bf16x8* p, bf16x8& tie) {
	asm volatile("global_load_dwordx4 %0, %2, off offset:%c3" : "=a"(dst), "+v"(tie) : "v"(p), "i"(OFF) : "memory");
}
template <int N> __device__ inline void x3s_wait8(f32x4 (&g)[8]) {
	asm volatile("s_waitcnt vmcnt(%c8)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]) : "i"(N) : "memory");
}
template <int N> __device__ inline void x3s_wait2(f32x4& a, f32x4& b) { asm volatile("s_waitcnt vmcnt(%c2)" : "+v"(a), "+v"(b) : "i"(N) : "memory"); }
template <int N, int NC> __device__ inline void x3s_wait_frags(u32x4 (&f)[NC][3]) {
	if constexpr (NC == 4)
		asm volatile("s_waitcnt vmcnt(%c12)" : "+a"(f[0][0]), "+a"(f[0][1]), "+a"(f[0][2]), "+a"(f[1][0]), "+a"(f[1][1]), "+a"(f[1][2]),
		             "+a"(f[2][0]), "+a"(f[2][1]), "+a"(f[2][2]), "+a"(f[3][0]), "+a"(f[3][1]), "+a"(f[3][2]) : "i"(N) : "memory");
	else
		asm volatile("s_waitcnt vmcnt(%c6)" : "+a"(f[0][0]), "+a"(f[0][1]), "+a"(f[0][2]), "+a"(f[1][0]), "+a"(f[1][1]), "+a"(f[1][2]) : "i"(N) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// Round 5: the same product on v_mfma_f32_16x16x32_bf16.
// The loop of k_factor_product_x3 above runs at 35-37 cycles per 32-cycle MFMA -- and at 1.2-1.35 GHz: the chip holds its clock down under this load (power),
// so wall time per launch is set by energy per MFMA, not by the issue stream (tools/stamp_x3.py, profiles/r05_x3_shape.md).  The 16 x 16 x 32 shape does the
// same FLOP per cycle and the chip holds a ~20 % higher clock on it with the same operand-split work beside it (tools/probe/x3_shape_probe.hip: 2.09-2.25 GHz
// against 1.73-1.98).  Same six exact terms per product, smallest first; another summation order inside an instruction (32 k instead of 16), so other bits
// than the 32 x 32 x 16 kernel -- every form of THIS kernel (x-tiled / y-tiled, 16- / 128-row image tiles, 64 / 32 panel columns per workgroup) gives the
// same bits as every other.
//
// Wave tile: 128 rows x (32 NBW) panel columns = 8 row blocks x (2 NBW) column blocks of 16 x 16, reduction in DOUBLE steps of 32 k (two K-steps of the
// split image).  The factor fragments are the MFMA's A operand (lane l: panel column 16 cb + l % 16, k = 8 (l / 16) .. + 7 -- the image's 16-byte unit,
// whatever the shape), the split values of V its B operand (lane l: row l % 16 of the row block, the same k), so accumulator register e of lane l is
// panel column 16 cb + 4 (l / 16) + e of that row: four CONSECUTIVE panel columns per lane = one 16-byte store into the slab.
// Row of row block rb held by lane l (r = l % 16):
//   x-tiled (either image): 64 (rb / 4) + 4 r + rb % 4 -- a lane's 16-byte load is four consecutive rows of one column, one for each row block of a group of four
//   y-tiled (16-row tiles): 16 rb + r -- a lane's two 16-byte loads are the eight k of its row; consecutive lanes read consecutive 64-byte column chunks
// Ring: one double step of V in flight per wave (two groups of 8 loads, each refilled in the phase that follows the last split of its values) and the
// factor fragments of the next double step in a second register set (12 loads, issued with the first group's refill): at most 28 loads in flight, as above.
template <int NBW, bool TR, int IMG, int DIAG = 0>
__global__ __launch_bounds__(256, NBW == 1 ? 2 : 1) void k_factor_product_x3s(
	const float* __restrict__ A, long tile_stride,
	const bf16x8* __restrict__ F, int NBT,
	float* __restrict__ slabs, long slab_stride, int RP,
	int steps_total, int xtiles, int splits, GramReduceArgs rg, unsigned long long* stamps) {
	static_assert(NBW == 1 || NBW == 2, "one or two 32-column blocks per wave tile");
	static_assert(!TR || IMG == 16, "the y-tiled form reads 16-row tiles");
	constexpr int TH = 128, NC = 2 * NBW;
	unsigned long long t_loop0 = 0, t_loop1 = 0, r_loop0 = 0, r_loop1 = 0, r_entry = 0, r_tail = 0;
	if (DIAG != 0) r_entry = __builtin_amdgcn_s_memrealtime();
	extern __shared__ __attribute__((aligned(16))) float lds[];
	constexpr bool FOLD = NBW == 1;
	const int pblocks = xtiles * splits * (FOLD ? 2 : 1);
	if (blockIdx.x >= (unsigned)pblocks) {
		// passengers: as in k_factor_product_x3
		if (blockIdx.y != 0) return;
		if (rg.inv_a != nullptr) inverse_gj64_body<float, 4>(rg.inv_a, 64, rg.inv_r, rg.inv_out, rg.inv_offdiag, rg.inv_diag);
		else if (rg.image != nullptr) gram_image_block(rg, blockIdx.x - pblocks, lds);
		else gram_reduce_block_x3<256>(rg, blockIdx.x - pblocks, lds);
		return;
	}
	int vb = blockIdx.x;
	if (XCD_REMAP) {
		const int q8 = pblocks / 8, r8 = pblocks % 8, xcd = vb % 8, idx = vb / 8;
		vb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
	}
	const int chunk = FOLD ? (vb & 1) : (int)blockIdx.y;
	if (FOLD) vb >>= 1;
	const int xt = vb % xtiles, sp = vb / xtiles;
	const int coff = 32 * NBW * chunk;
	const long fstep = (long)NBT * 192;                 // factor fragments per K-step of 16
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int r16 = lane & 15, kq = lane >> 4;
	const int nw = splits * 4;
	const int widx = sp * 4 + wave;
	// the reduction range in double steps, dealt to the nw waves of the x-tile; a double step past an odd range's end meets the all-zero K-step that closes the
	// factor image (index steps_total) with a re-read of the last valid K-step of A
	const int units = (steps_total + 1) / 2;
	const int S0 = (int)(((long)units * widx) / nw), S1 = (int)(((long)units * (widx + 1)) / nw);
	const int kend = steps_total - 1;

	f32x4 acc[8][NC];
#pragma unroll
	for (int rb = 0; rb < 8; ++rb)
#pragma unroll
		for (int cb = 0; cb < NC; ++cb)
#pragma unroll
			for (int e = 0; e < 4; ++e) acc[rb][cb][e] = 0.f;

	if (S1 > S0) {
		// Every load of the loop is WRITTEN OUT (asm, x3s_* above), and so is every wait for one.  Why: the factor fragments must land in the accumulator half of
		// the register file (an MFMA reads its A / B operands from there as well; two sets of 12 NBW registers beside the 64 NBW accumulators leave the 256
		// architectural VGPRs to the ring of V and the operand split) and hipcc only loads into VGPRs and copies (78 v_accvgpr_write per double step); and a
		// wait hipcc counts for ITS loads waits for every written-out load in flight as well.  So: loads as asm statements placed where they are to issue --
		// each behind an MFMA, tied to the operand register the MFMAs on either side read, which is what keeps hipcc's scheduler from moving them --, counted
		// s_waitcnt statements that name the registers they release (the consumers read the statement's outputs: nothing is read before its wait), and a
		// drain at the loop's end that keeps the destination registers of the last (unused) requests live until they have landed.
		constexpr bool RA = DIAG == 0 || DIAG == 2 || DIAG == 4, RF = DIAG == 0 || DIAG == 3 || DIAG == 4;       // refill V / the fragments (measurement forms switch them off)
		// lane part of the streamed operand's address; + the lane's K-step (2 S + kq / 2, clamped) times its stride, + the group's offset; loads at constant offsets
		const float* ap = IMG == 16 ? (TR ? A + ((long)xt * TH + r16) * 16 + 8 * (kq & 1)
		                                  : A + ((long)xt * 8 + (r16 >> 2)) * tile_stride + (8 * (kq & 1)) * 16 + 4 * (r16 & 3))
		                            : A + (long)xt * tile_stride + (8 * (kq & 1)) * TH + 4 * r16;
		const long kstride = IMG == 16 ? (TR ? tile_stride : 256) : 16 * TH;            // floats per K-step of 16
		const long gstride = IMG == 16 ? (TR ? 64 * 16 : 4 * tile_stride) : 64;           // floats between the two groups of four row blocks
		auto a_step = [&](int S) -> const float* {
			int so = 2 * S + (kq >> 1);
			so = so < kend ? so : kend;
			return ap + (long)so * kstride;
		};
		const bf16x8* fp = F + (long)(kq >> 1) * fstep + (kq & 1) * 32 + r16 + (long)chunk * (NBW * 192);      // + 2 S * fstep + (cb / 2) * 192 + plane * 64 + 16 (cb % 2)
		f32x4 va[2][8];
		u32x4 fb[2][NC][3];
		bf16x8 op[2][3];
#define X3S_IC(n) std::integral_constant<int, (n)>{}
		{
			const float* s_ = a_step(S0);
			const float* s1_ = s_ + gstride;
			const bf16x8* f_ = fp + (long)(2 * S0) * fstep;
			const bf16x8* f1_ = f_ + 192;
#define X3S_PRO(i) x3s_load_v<x3s_a_off<TR, IMG>(i)>(va[0][i], s_); x3s_load_v<x3s_a_off<TR, IMG>(i)>(va[1][i], s1_);
			X3S_PRO(0) X3S_PRO(1) X3S_PRO(2) X3S_PRO(3) X3S_PRO(4) X3S_PRO(5) X3S_PRO(6) X3S_PRO(7)
#undef X3S_PRO
#define X3S_PRO(cb, pl) if constexpr (cb < NC) { x3s_load_a<x3s_f_off(cb, pl)>(fb[0][cb][pl], cb < 2 ? f_ : f1_); if (!RF) x3s_load_a<x3s_f_off(cb, pl)>(fb[1][cb][pl], cb < 2 ? f_ : f1_); }
			X3S_PRO(0, 0) X3S_PRO(0, 1) X3S_PRO(0, 2) X3S_PRO(1, 0) X3S_PRO(1, 1) X3S_PRO(1, 2) X3S_PRO(2, 0) X3S_PRO(2, 1) X3S_PRO(2, 2) X3S_PRO(3, 0) X3S_PRO(3, 1) X3S_PRO(3, 2)
#undef X3S_PRO
			x3s_wait8<0>(va[0]); x3s_wait8<0>(va[1]);
			x3s_wait_frags<0, NC>(fb[0]);
			if (!RF) x3s_wait_frags<0, NC>(fb[1]);
		}
		__builtin_amdgcn_sched_barrier(0);
		if (DIAG != 0) { t_loop0 = __builtin_amdgcn_s_memtime(); r_loop0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
		auto operand = [&](int rb, float (&v)[8]) {
#pragma unroll
			for (int j = 0; j < 8; ++j) v[j] = TR ? va[rb >> 2][2 * (rb & 3) + (j >> 2)][j & 3] : va[rb >> 2][j][rb & 3];
		};
		{ float v[8]; operand(0, v); split3(v, op[0][0], op[0][1], op[0][2]); }
		// One double step = eight phases (row blocks).  Phase rb: the 6 NC MFMAs of its row block, interleaved with the split of row block rb + 1 (in phase 7: of the
		// NEXT double step).  Requests, in issue order -- x-tiled: phase 3 group 0 of V of the next step (its last split ran in phase 2), phase 4 the next step's
		// fragments into the other register set, phase 7 group 1; y-tiled: a row block's two loads in ITS phase (its split ran one phase ago: eight phases ahead of
		// their next use), the fragments in phase 4 behind that phase's two.  Waits (requests younger than the awaited one that may stay in flight) -- x-tiled: group 1
		// at phase 3: none; group 0 at phase 7: the fragments; the fragments at the step's head: group 1.  y-tiled: row block rb + 1 at phase rb: six phases' pairs
		// and the fragments (at phase 4: the pairs only); the fragments at the step's head: three pairs.
		constexpr int NFR = RF ? 3 * NC : 0;                  // fragment requests per double step
		auto step = [&](auto U, int S) __attribute__((always_inline)) {
			constexpr int u = decltype(U)::value;
			int Sn = S + 1;
			Sn = Sn < S1 ? Sn : S1 - 1;                                         // past this wave's piece: a harmless re-read
			const float* sn0_ = a_step(Sn);
			const float* sn1_ = sn0_ + gstride;
			const bf16x8* fn_ = fp + (long)(2 * Sn) * fstep;
			const bf16x8* fn1_ = fn_ + 192;
			if (S != S0) x3s_wait_frags<(TR ? (RA ? 6 : 0) : (RA ? 8 : 0)), NC>(fb[u]);
			__builtin_amdgcn_sched_barrier(0);
			auto phase = [&](auto RB) __attribute__((always_inline)) {
				constexpr int rb = decltype(RB)::value, cur = rb & 1, nxt = cur ^ 1, nrb = (rb + 1) & 7;
				if constexpr (TR) x3s_wait2<(RA ? 12 : 0) + (rb == 4 ? 0 : NFR)>(va[nrb >> 2][2 * (nrb & 3)], va[nrb >> 2][2 * (nrb & 3) + 1]);
				else if constexpr (rb == 3) x3s_wait8<0>(va[1]);
				else if constexpr (rb == 7) x3s_wait8<NFR>(va[0]);
				{ float v[8]; operand(nrb, v); split3(v, op[nxt][0], op[nxt][1], op[nxt][2]); }
				// MFMA (t, cb), t = term (smallest first: planes (fragment, operand) = (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)), then the request slot behind it: slot = t NC + cb.
				// x-tiled: the eight loads of a group in slots 1, 3, .. 15 (NC = 2: 0 .. 7) of phases 3 / 7; y-tiled: the row block's pair in slots 1 and 3 of every phase;
				// the fragments in phase 4: NC = 4: slots 4 .. 15; NC = 2: slots 4, 6, .. 14
				auto issue = [&](auto SLOT, bf16x8& tie) __attribute__((always_inline)) {
					constexpr int slot = decltype(SLOT)::value;
					if constexpr (RA && TR && (slot == 1 || slot == 3)) {
						constexpr int i = 2 * (rb & 3) + (slot == 3 ? 1 : 0);
						x3s_load_v_tied<x3s_a_off<TR, IMG>(i)>(va[rb >> 2][i], rb < 4 ? sn0_ : sn1_, tie);
					}
					if constexpr (RA && !TR && (rb == 3 || rb == 7) && NC == 4 && (slot & 1) == 1 && slot < 16)
						x3s_load_v_tied<x3s_a_off<TR, IMG>(slot >> 1)>(va[rb >> 2][slot >> 1], rb == 3 ? sn0_ : sn1_, tie);
					if constexpr (RA && !TR && (rb == 3 || rb == 7) && NC == 2 && slot < 8)
						x3s_load_v_tied<x3s_a_off<TR, IMG>(slot)>(va[rb >> 2][slot], rb == 3 ? sn0_ : sn1_, tie);
					if constexpr (RF && rb == 4 && NC == 4 && slot >= 4 && slot < 16) {
						constexpr int fi = slot - 4, cb = fi / 3, pl = fi % 3;
						x3s_load_a_tied<x3s_f_off(cb, pl)>(fb[u ^ 1][cb][pl], cb < 2 ? fn_ : fn1_, tie);
					}
					if constexpr (RF && rb == 4 && NC == 2 && slot >= 2 && slot < 8) {
						constexpr int fi = slot - 2, cb = fi / 3, pl = fi % 3;
						x3s_load_a_tied<x3s_f_off(cb, pl)>(fb[u ^ 1][cb][pl], fn_, tie);
					}
				};
				auto term = [&](auto T, auto PF, auto PO) __attribute__((always_inline)) {
					constexpr int t = decltype(T)::value, pf = decltype(PF)::value, po = decltype(PO)::value;
					acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[u][0][pf]), op[cur][po], acc[rb][0], 0, 0, 0);
					issue(X3S_IC(t * NC + 0), op[cur][po]);
					acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[u][1][pf]), op[cur][po], acc[rb][1], 0, 0, 0);
					issue(X3S_IC(t * NC + 1), op[cur][po]);
					if constexpr (NC == 4) {
						acc[rb][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[u][2][pf]), op[cur][po], acc[rb][2], 0, 0, 0);
						issue(X3S_IC(t * NC + 2), op[cur][po]);
						acc[rb][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[u][3][pf]), op[cur][po], acc[rb][3], 0, 0, 0);
						issue(X3S_IC(t * NC + 3), op[cur][po]);
					}
				};
				term(X3S_IC(0), X3S_IC(0), X3S_IC(2));
				term(X3S_IC(1), X3S_IC(2), X3S_IC(0));
				term(X3S_IC(2), X3S_IC(1), X3S_IC(1));
				term(X3S_IC(3), X3S_IC(0), X3S_IC(1));
				term(X3S_IC(4), X3S_IC(1), X3S_IC(0));
				term(X3S_IC(5), X3S_IC(0), X3S_IC(0));
				constexpr int NM = 6 * NC;                      // MFMAs of the phase; 44 VALU of the split (+ a few of addressing)
				constexpr int VPM = (44 + NM - 1) / NM;
#pragma unroll
				for (int gi = 0; gi < NM; ++gi) {
					__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
					__builtin_amdgcn_sched_group_barrier(0x002, VPM, 0); // VALU
				}
				__builtin_amdgcn_sched_barrier(0);
			};
			phase(X3S_IC(0)); phase(X3S_IC(1)); phase(X3S_IC(2)); phase(X3S_IC(3)); phase(X3S_IC(4)); phase(X3S_IC(5)); phase(X3S_IC(6)); phase(X3S_IC(7));
		};
		int S = S0;
		for (; S + 2 <= S1; S += 2) {
			step(std::integral_constant<int, 0>{}, S);
			step(std::integral_constant<int, 1>{}, S + 1);
		}
		if (S < S1) step(std::integral_constant<int, 0>{}, S);
		// the last step's requests (re-reads that nobody uses) are still in flight and hipcc does not know of them: wait, and keep their destination registers
		// live until then
		x3s_wait8<0>(va[0]); x3s_wait8<0>(va[1]);
		x3s_wait_frags<0, NC>(fb[0]); x3s_wait_frags<0, NC>(fb[1]);
		__builtin_amdgcn_sched_barrier(0);
#undef X3S_IC
		if (DIAG != 0) { __builtin_amdgcn_sched_barrier(0); t_loop1 = __builtin_amdgcn_s_memtime(); r_loop1 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
	}

	if (DIAG != 0) { __builtin_amdgcn_sched_barrier(0); r_tail = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
	// in-workgroup sum of the four waves' tiles through LDS in wave order, sixteen 16 x 16 tiles per round; a lane's accumulator IS a 16-byte piece of the
	// slab (four consecutive panel columns of one row)
	f32x4* l4 = reinterpret_cast<f32x4*>(lds);
	float* slab = slabs + (long)sp * slab_stride;
	constexpr int ROUNDS = 8 * NC / 16;
#pragma unroll
	for (int rd = 0; rd < ROUNDS; ++rd) {
		if (rd > 0) __syncthreads();
#pragma unroll
		for (int tl = 0; tl < 16; ++tl) {
			const int t = 16 * rd + tl;
			l4[(wave * 16 + tl) * 64 + lane] = acc[t / NC][t % NC];
		}
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int tl = wave * 4 + i;
			const int t = 16 * rd + tl, rb = t / NC, cb = t % NC;
			f32x4 s = l4[(0 * 16 + tl) * 64 + lane];
#pragma unroll
			for (int p = 1; p < 4; ++p) s += l4[(p * 16 + tl) * 64 + lane];
			const int x = xt * TH + (TR ? 16 * rb + r16 : 64 * (rb >> 2) + 4 * r16 + (rb & 3));
			*reinterpret_cast<f32x4*>(slab + (long)x * RP + coff + 16 * cb + 4 * kq) = s;
		}
	}
	if (DIAG != 0 && stamps != nullptr) {
		__builtin_amdgcn_s_waitcnt(0);
		const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
		if (lane == 0) {
			// as k_factor_product_x3: shader cycles and 100 MHz ticks in the main loop, K-steps (of 16) run there; then 100 MHz stamps of the wave's life
			unsigned long long* o = stamps + 8 * ((long)blockIdx.x * 4 + wave);
			o[0] = t_loop1 - t_loop0; o[1] = r_loop1 - r_loop0; o[2] = (unsigned long long)(2 * (S1 - S0));
			o[3] = r_entry; o[4] = r_loop0; o[5] = r_loop1; o[6] = r_tail; o[7] = r_end;
		}
	}
}

// Every wave piece (four per slice) gets at least two turns of the 3-deep ring; as many slices as fill the chip.
// reserve: CUs kept free for passenger workgroups beyond the sixteen (K-split Gram passengers of a column shard's W^T V)
int plan_splits_x3(int xtiles, int KS, int num_cus, int reserve) {
	if (reserve > 0) {
		const int by_fill = std::max(1, (num_cus - reserve) / std::max(1, xtiles));
		const int by_depth = std::max(1, KS / (6 * 4));
		return std::max(1, std::min(by_fill, by_depth));
	}
	// (fewer than sixteen x-tiles -- a column shard's W^T V: leave the sixteen passenger workgroups a CU each, Engine::passengers_ride)
	const int by_fill = std::max(1, (xtiles < GRAM_REDUCE_BLOCKS ? num_cus - GRAM_REDUCE_BLOCKS : num_cus) / std::max(1, xtiles));
	const int by_depth = std::max(1, KS / (6 * 4));
	return std::max(1, std::min(by_fill, by_depth));
}

// events handed from launch_factor_product_x3 to the instantiation it dispatches to (per thread: rank threads launch concurrently)
static thread_local hipEvent_t t_ev_start = nullptr, t_ev_stop = nullptr;

template <int D, int WAVES, int DIAG = 0, int NBW = 2, bool TR = false, int IMG = 128, bool YLDS = false, bool R32 = false>
static hipError_t launch_fp_x3(const FactorProductPlan& p, const float* A, long tile_stride, const void* F, int RP,
                               float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg, unsigned long long* stamps = nullptr) {
	GramReduceArgs none = {nullptr, 0, nullptr, nullptr, 0};
	const bool wanted = rg != nullptr && (rg->partials != nullptr || rg->inv_a != nullptr || rg->image != nullptr);
	// (the passengers sit behind the product blocks in the grid, whatever the number of x-tiles; the caller sees to it that they find a CU
	//  while the product runs -- Engine::passengers_ride)
	const bool with_reduce = wanted && RP == 64;
	if (wanted && !with_reduce) return hipErrorInvalidValue;
	const int passengers = !with_reduce ? 0 : (rg->inv_a != nullptr ? 1 : (rg->image != nullptr && rg->ksplit > 1) ? GRAM_IMAGE_TILES * rg->ksplit : GRAM_REDUCE_BLOCKS);
	if (NBW == 1 && RP != 64) return hipErrorInvalidValue;
	dim3 grid(p.xtiles * p.splits * (NBW == 1 ? 2 : 1) + passengers, NBW == 1 ? 1 : RP / (32 * NBW), 1), block(64 * WAVES);
	const size_t lds_bytes = std::max<size_t>(std::max<size_t>(WAVES * 4 * 4 * 64 * sizeof(f32x4), 1024 * sizeof(float)), YLDS ? WAVES * 2 * 128 * 20 * sizeof(float) : 0);
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_factor_product_x3<D, WAVES, DIAG, NBW, TR, IMG, YLDS, R32>), (int)lds_bytes, lds_done); e != hipSuccess) return e;
	if (t_ev_start != nullptr && t_ev_stop != nullptr) {
		// (the caller wants this launch timed: its own start / stop timestamps, no event records around it)
		const hipEvent_t e0 = t_ev_start, e1 = t_ev_stop;
		t_ev_start = t_ev_stop = nullptr;
		hipExtLaunchKernelGGL((k_factor_product_x3<D, WAVES, DIAG, NBW, TR, IMG, YLDS, R32>), grid, block, (std::uint32_t)lds_bytes, stream, e0, e1, 0u,
		                      A, tile_stride, reinterpret_cast<const bf16x8*>(F), RP / 32, slabs, slab_stride, RP, p.steps_total, p.xtiles, p.splits, with_reduce ? *rg : none, stamps);
		return hipGetLastError();
	}
	hipLaunchKernelGGL((k_factor_product_x3<D, WAVES, DIAG, NBW, TR, IMG, YLDS, R32>), grid, block, lds_bytes, stream,
	                   A, tile_stride, reinterpret_cast<const bf16x8*>(F), RP / 32, slabs, slab_stride, RP, p.steps_total, p.xtiles, p.splits, with_reduce ? *rg : none, stamps);
	return hipGetLastError();
}

template <int NBW, bool TR, int IMG, int DIAG = 0>
static hipError_t launch_fp_x3s(const FactorProductPlan& p, const float* A, long tile_stride, const void* F, int RP,
                                float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg, unsigned long long* stamps = nullptr) {
	GramReduceArgs none = {nullptr, 0, nullptr, nullptr, 0};
	const bool wanted = rg != nullptr && (rg->partials != nullptr || rg->inv_a != nullptr || rg->image != nullptr);
	const bool with_reduce = wanted && RP == 64;
	if (wanted && !with_reduce) return hipErrorInvalidValue;
	const int passengers = !with_reduce ? 0 : (rg->inv_a != nullptr ? 1 : (rg->image != nullptr && rg->ksplit > 1) ? GRAM_IMAGE_TILES * rg->ksplit : GRAM_REDUCE_BLOCKS);
	if (NBW == 1 && RP != 64) return hipErrorInvalidValue;
	dim3 grid(p.xtiles * p.splits * (NBW == 1 ? 2 : 1) + passengers, NBW == 1 ? 1 : RP / (32 * NBW), 1), block(256);
	const size_t lds_bytes = 4 * 16 * 64 * sizeof(f32x4);          // the epilogue's exchange image (the passengers need less)
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_factor_product_x3s<NBW, TR, IMG, DIAG>), (int)lds_bytes, lds_done); e != hipSuccess) return e;
	if (t_ev_start != nullptr && t_ev_stop != nullptr) {
		const hipEvent_t e0 = t_ev_start, e1 = t_ev_stop;
		t_ev_start = t_ev_stop = nullptr;
		hipExtLaunchKernelGGL((k_factor_product_x3s<NBW, TR, IMG, DIAG>), grid, block, (std::uint32_t)lds_bytes, stream, e0, e1, 0u,
		                      A, tile_stride, reinterpret_cast<const bf16x8*>(F), RP / 32, slabs, slab_stride, RP, p.steps_total, p.xtiles, p.splits, with_reduce ? *rg : none, stamps);
		return hipGetLastError();
	}
	hipLaunchKernelGGL((k_factor_product_x3s<NBW, TR, IMG, DIAG>), grid, block, lds_bytes, stream,
	                   A, tile_stride, reinterpret_cast<const bf16x8*>(F), RP / 32, slabs, slab_stride, RP, p.steps_total, p.xtiles, p.splits, with_reduce ? *rg : none, stamps);
	return hipGetLastError();
}

// y_tiled: A is the image tiled along the REDUCTION index (128-row tiles of y, tile_stride apart, each holding all x as
// columns of 128 contiguous y) -- i.e. the x-tiled image of the transposed matrix; steps_total K-steps of 16 y, and the
// image must cover 128 * xtiles columns.  image_tile: rows of the tiled index per tile of the image, 128 (tile_stride = 128 *
// columns) or 16 (tile_stride = 16 * columns; both forms then read contiguous memory, see the kernel).
// A: x-tiled fp32 image (tile height 128, zero-filled up to a multiple of 16 columns); F: k_pack_panel_x3
// image of the RP-column panel (RP a multiple of 64; grid.y = RP / 64 passes over A); p.steps_total = K-steps
// of 16; p.th must be 128.  Passengers (Gram reduction, or the 64 x 64 inverse) ride only at RP = 64.
hipError_t launch_factor_product_x3(const FactorProductPlan& p, const float* A, long tile_stride, const void* F, int RP,
                                    float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg, unsigned long long* stamps,
                                    bool y_tiled, int image_tile, hipEvent_t ev_start, hipEvent_t ev_stop) {
	if (RP % 64 != 0 || p.th != 128 || (image_tile != 128 && image_tile != 16)) return hipErrorInvalidValue;
	t_ev_start = ev_start; t_ev_stop = ev_stop;
	struct Clear { ~Clear() { t_ev_start = t_ev_stop = nullptr; } } clear_on_exit;      // (a path that did not consume them -- an error return, grid.y > 1 -- leaves nothing behind)
	// Round 5: panels of 64 k columns (k odd) run on the 16 x 16 x 32 MFMA shape (k_factor_product_x3s); NMFAMD_X3_SHAPE32 (measurement builds) keeps the 32 x 32 x 16 kernel.
	// Whole multiples of 128 columns stay on the 128-column form of the 32 x 32 x 16 kernel (half the passes over A).
	static const bool shape32 = tuning_env("NMFAMD_X3_SHAPE32") != nullptr;
	// (the 128 x 32 form, two workgroups per CU at 256 registers each, does not fit that budget with its ring in written-out loads yet: it stays on the 32 x 32 x 16 kernel)
	if (!shape32 && RP % 128 != 0 && !(y_tiled && image_tile != 16) && !(RP == 64 && p.col_split == 2)) {
		const bool fold = false;
#ifdef NMFAMD_DIAG_BUILD
		if (image_tile == 16 && RP == 64 && stamps != nullptr && !fold) {
			// stamped forms (tools/stamp_x3.py): NMFAMD_X3_VARIANT = 10..13 the x-tiled form, 30..33 the y-tiled one; last digit 0 = no ring refill, 1 = A only, 2 = F only, 3 = the production loop
			static const int yv = [] { const char* e = tuning_env("NMFAMD_X3_VARIANT"); return e ? std::atoi(e) : 0; }();
			if (!y_tiled) switch (yv % 10) {
				case 0: return launch_fp_x3s<2, false, 16, 1>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
				case 1: return launch_fp_x3s<2, false, 16, 2>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
				case 2: return launch_fp_x3s<2, false, 16, 3>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
				default: return launch_fp_x3s<2, false, 16, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
			}
			switch (yv % 10) {
				case 0: return launch_fp_x3s<2, true, 16, 1>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
				case 1: return launch_fp_x3s<2, true, 16, 2>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
				case 2: return launch_fp_x3s<2, true, 16, 3>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
				default: return launch_fp_x3s<2, true, 16, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
			}
		}
#endif
		if (stamps != nullptr) return hipErrorNotSupported;
		if (image_tile == 16) {
			if (fold) return y_tiled ? launch_fp_x3s<1, true, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3s<1, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
			return y_tiled ? launch_fp_x3s<2, true, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3s<2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
		}
		if (fold) return launch_fp_x3s<1, false, 128>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
		return launch_fp_x3s<2, false, 128>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
	}
#ifdef NMFAMD_DIAG_BUILD
	if (image_tile == 16 && RP == 64 && stamps != nullptr) {
		// stamped diagnostic builds of the PRODUCTION forms on the one resident image (tools/stamp_x3.py): NMFAMD_X3_VARIANT = 10..13 the x-tiled form,
		// 30..33 the y-tiled form (rows 32 b + r per lane); last digit 0 = no ring refill, 1 = refill A only, 2 = refill F only, 3 = the production loop
		static const int yv = [] { const char* e = tuning_env("NMFAMD_X3_VARIANT"); return e ? std::atoi(e) : 0; }();
		if (!y_tiled) switch (yv % 10) {
			case 0: return launch_fp_x3<X3_RING_X, 4, 1, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
			case 1: return launch_fp_x3<X3_RING_X, 4, 2, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
			case 2: return launch_fp_x3<X3_RING_X, 4, 3, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
			default: return launch_fp_x3<X3_RING_X, 4, 4, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
		}
		switch (yv % 10) {
			case 0: return launch_fp_x3<X3_RING_X, 4, 1, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
			case 1: return launch_fp_x3<X3_RING_X, 4, 2, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
			case 2: return launch_fp_x3<X3_RING_X, 4, 3, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
			default: return launch_fp_x3<X3_RING_X, 4, 4, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps);
		}
	}
#else
	if (stamps != nullptr) return hipErrorNotSupported;      // stamped kernels exist in the diagnostic build only (tuning.h)
#endif
	if (image_tile == 16) {
		// y-tiled form on 16-row tiles (one resident image, config 2's W^T V).  Round 4: loads straight to registers with MFMA row r of M-block b = tile row 32 b + r, so that
		// consecutive lanes read consecutive 64-byte column chunks (16 cache lines per wave instruction): 42.5 us per launch against 46.3 for the LDS-staged form (rounds 2-3's
		// default) and 47.4 for direct loads with the x-tiled form's interleaved rows (4 r + b: 32 lines per instruction) on the same box; results bit-identical.
		// NMFAMD_X3_YLDS / NMFAMD_X3_YDIRECT (measurement builds) select the older forms.
		static const bool ylds = tuning_env("NMFAMD_X3_YLDS") != nullptr, ydirect = tuning_env("NMFAMD_X3_YDIRECT") != nullptr;
		if (RP % 128 == 0) {
			if (!y_tiled) return launch_fp_x3<2, 4, 0, 4, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
			return ydirect ? launch_fp_x3<2, 4, 0, 4, true, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg)
			               : launch_fp_x3<2, 4, 0, 4, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
		}
#ifdef NMFAMD_DIAG_BUILD
		if (RP == 64 && p.col_split == 2) {       // (measured slower at config 2: 41.9 -> 51.0 and 40.8 -> 45.4 us per launch; Engine::init)
			if (!y_tiled) return launch_fp_x3<X3_RING_X, 4, 0, 1, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
			return launch_fp_x3<X3_RING_X, 4, 0, 1, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
		}
#endif
		if (!y_tiled) return launch_fp_x3<X3_RING_X, 4, 0, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
		if (ylds) return launch_fp_x3<X3_RING_Y, 4, 0, 2, true, 16, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
		if (ydirect) return launch_fp_x3<X3_RING_X, 4, 0, 2, true, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
		return launch_fp_x3<X3_RING_X, 4, 0, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
	}
	if (y_tiled) {
		if (RP % 128 == 0) return launch_fp_x3<2, 4, 0, 4, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
#ifdef NMFAMD_DIAG_BUILD
		if (RP == 64 && p.col_split == 2) return launch_fp_x3<X3_RING_X, 4, 0, 1, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
#endif
		return launch_fp_x3<X3_RING_X, 4, 0, 2, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
	}
	static const int variant = [] { const char* e = tuning_env("NMFAMD_X3_VARIANT"); return e ? std::atoi(e) : 0; }();   // A/B switch for measurements
	// short reduction ranges (a column shard's V H^T): 128 x 32 per workgroup, two workgroups per x-tile (FactorProductPlan::col_split)
	if (RP == 64 && p.col_split == 2) return launch_fp_x3<X3_RING_X, 4, 0, 1>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
	// wide panels: 128 columns per pass over A (256 accumulator registers, ring depth 2) -- half the passes, MFMA-bound
	if (RP % 128 == 0 && variant != 20) return launch_fp_x3<2, 4, 0, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
#ifdef NMFAMD_DIAG_BUILD
	switch (variant) {
	case 1: return launch_fp_x3<2, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
	case 2: return launch_fp_x3<4, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
	default: break;
	}
#endif
	return launch_fp_x3<X3_RING_X, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
}

} // namespace nmfamd
