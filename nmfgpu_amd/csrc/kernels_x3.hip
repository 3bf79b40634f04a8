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
#include <type_traits>
#include <cstdlib>

#include "kernels.h"
#include "tuning.h"
#include "split3.h"
#include "inverse_gj64.h"
#include "gram_image.h"
#include "gram_wide.h"

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
#ifndef X3_EPILOGUE_ONE_ROUND
#define X3_EPILOGUE_ONE_ROUND 1       // (A/B switch: 0 = the in-workgroup sum in two rounds of four accumulator tiles, 64 KiB of LDS, as rounds 2-4 had it)
#endif
#ifndef X3_DEAL_TURNS
#define X3_DEAL_TURNS 0               // (A/B switch, tools/build_variant.sh: 1 = always deal the K-steps in whole turns of the ring, as rounds 2-4 did)
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
// ODD (ring depth 2): every wave piece is an ODD number of K-steps -- see the dealing below
template <int D, int X3_WAVES, int DIAG = 0, int NBW = 2, bool TR = false, int IMG = 128, bool YLDS = false, bool R32 = false, bool ODD = false>
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
		if constexpr (NBW == 4) {
			// padded ranks 128 ... 512: K slices of the panel's Gram matrix (gram_wide.h; ring of four K-steps -- under the product's memory stream a load comes back late)
			if (rg.wide_P != nullptr) {
				const int pid = (int)blockIdx.x - pblocks;
				gram_wide_slice<4, false>(rg.wide_P, RP, rg.wide_len, rg.wide_parts, rg.wide_partial, pid % rg.wide_parts, pid / rg.wide_parts);
			}
			return;
		}
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
	// ODD (round 5): whole turns made config 2's pieces 28 K-steps for the last wave of a launch's last K slice beside 26 for the others (313 steps over 12
	// pieces, 625 over 24) and the workgroups of that slice ended every launch ~2 us after the rest (per-workgroup stamps, tools/stamp_x3.py).  Pieces of an
	// odd count each -- one step out of ring slot 1 AHEAD of the loop, then whole turns -- can be 27 and 25 there: piece i = 2 a_i + 1 steps with the a_i dealt
	// evenly.  Every wave of the launch takes the same path (a piece-dependent branch in front of the loop made hipcc wait for vmcnt(0) inside it), the host
	// picks the form with the shorter longest piece (x3_odd_pieces), the same for every instantiation of a width: the forms stay bit-identical to each other.
	static_assert(!ODD || (D == 2 && !YLDS), "odd pieces: ring depth 2, loads straight to registers");
	constexpr int DU = D <= 2 ? 2 : D;
	const int units = (steps_total + DU - 1) / DU;
	const int pairs = (steps_total - nw + 1) / 2;               // ODD: whole turns to deal out beside one step per piece (the host made sure steps_total >= nw)
	const int s0 = ODD ? 2 * (int)(((long)pairs * widx) / nw) + widx : DU * (int)(((long)units * widx) / nw);
	const int s1 = ODD ? 2 * (int)(((long)pairs * (widx + 1)) / nw) + widx + 1 : DU * (int)(((long)units * (widx + 1)) / nw);
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
			int st = s0 + (ODD ? (d ^ 1) : d);                      // (ODD: step s0 waits in ring slot 1, step s0 + 1 in slot 0; a piece of one step re-reads it there)
			st = st < last ? st : last;
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
			for (int j = 0; j < 8; ++j) v[j] = TR ? va[ODD ? D - 1 : 0][j >> 2][j & 3] : va[ODD ? D - 1 : 0][j][0];
			split3(v, op[0][0], op[0][1], op[0][2]);
		}
		// one K-step out of ring slot d; t = the first step of the turn it belongs to
		auto kstep = [&](auto dtag, const int t) {
			constexpr int d = decltype(dtag)::value;
			{
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
		};
#define X3_KSTEP(dd) kstep(std::integral_constant<int, dd>{}, t)
		int t = 0;
		if constexpr (ODD) { t = -1; X3_KSTEP(1); t = 1; }      // the piece's first step: "turn -1", slot 1 -- its refill is the step slot 1 holds in the loop's first turn
		for (; t < steps; t += D) {
			X3_KSTEP(0);
			if constexpr (D > 1) X3_KSTEP(1);
			if constexpr (D > 2) X3_KSTEP(2);
			if constexpr (D > 3) X3_KSTEP(3);
			static_assert(D <= 4, "ring depth");
		}
#undef X3_KSTEP
		if (DIAG != 0) { __builtin_amdgcn_sched_barrier(0); t_loop1 = __builtin_amdgcn_s_memtime(); r_loop1 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
	}

	if (DIAG != 0) { __builtin_amdgcn_sched_barrier(0); r_tail = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
	// in-workgroup sum through LDS, two M-blocks (four tiles) per round; C/D map: register g of lane l
	// is MFMA row i = (g & 3) + 8 (g >> 2) + 4 (l >> 5), column l & 31; output row x = 128 xt + 4 i + b.
	if (YLDS) __syncthreads();            // the epilogue image overlays the staging slots of every wave
	f32x4* l4 = reinterpret_cast<f32x4*>(lds);
	float* slab = slabs + (long)sp * slab_stride;
	// TPR accumulator tiles per round: four (64 KiB of LDS for four waves), or -- X3_EPILOGUE_ONE_ROUND, the 64-column forms -- all eight at once (128 KiB: one
	// workgroup per CU has them), one barrier instead of three
	constexpr int TPR = (X3_EPILOGUE_ONE_ROUND && NBW == 2 && !YLDS) ? 8 : 4;
	constexpr int ROUNDS = 4 * NBW / TPR;
#pragma unroll
	for (int rd = 0; rd < ROUNDS; ++rd) {
		if (rd > 0) __syncthreads();
#pragma unroll
		for (int tl = 0; tl < TPR; ++tl) {
			const int b = (TPR * rd + tl) / NBW, nb = (TPR * rd + tl) % NBW;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				f32x4 v;
				v[0] = acc[b][nb][4 * q + 0]; v[1] = acc[b][nb][4 * q + 1];
				v[2] = acc[b][nb][4 * q + 2]; v[3] = acc[b][nb][4 * q + 3];
				l4[((wave * TPR + tl) * 4 + q) * 64 + lane] = v;
			}
		}
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 4 * TPR / X3_WAVES; ++i) {
			const int sl = wave * (4 * TPR / X3_WAVES) + i;  // slice = (tile, q)
			const int q = sl & 3, tl = sl >> 2;
			const int b = (TPR * rd + tl) / NBW, nb = (TPR * rd + tl) % NBW;
			f32x4 s = l4[((0 * TPR + tl) * 4 + q) * 64 + lane];
#pragma unroll
			for (int p = 1; p < X3_WAVES; ++p) s += l4[((p * TPR + tl) * 4 + q) * 64 + lane];
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

template <int D, int WAVES, int DIAG = 0, int NBW = 2, bool TR = false, int IMG = 128, bool YLDS = false, bool R32 = false, bool ODD = false>
static hipError_t launch_fp_x3(const FactorProductPlan& p, const float* A, long tile_stride, const void* F, int RP,
                               float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg, unsigned long long* stamps = nullptr) {
	GramReduceArgs none = {nullptr, 0, nullptr, nullptr, 0};
	const bool wide = rg != nullptr && rg->wide_P != nullptr;
	if (wide && (NBW != 4 || RP % 128 != 0 || rg->wide_parts < 1 || rg->wide_partial == nullptr)) return hipErrorInvalidValue;
	const bool wanted = rg != nullptr && !wide && (rg->partials != nullptr || rg->inv_a != nullptr || rg->image != nullptr);
	// (the passengers sit behind the product blocks in the grid, whatever the number of x-tiles; the caller sees to it that they find a CU
	//  while the product runs -- Engine::passengers_ride)
	const bool with_reduce = (wanted && RP == 64) || wide;
	if (wanted && !with_reduce) return hipErrorInvalidValue;
	const int passengers = !with_reduce ? 0 : wide ? rg->wide_parts * ((RP / 128) * (RP / 128 + 1) / 2)
	                                    : (rg->inv_a != nullptr ? 1 : (rg->image != nullptr && rg->ksplit > 1) ? GRAM_IMAGE_TILES * rg->ksplit : GRAM_REDUCE_BLOCKS);
	if (NBW == 1 && RP != 64) return hipErrorInvalidValue;
	dim3 grid(p.xtiles * p.splits * (NBW == 1 ? 2 : 1) + passengers, NBW == 1 ? 1 : RP / (32 * NBW), 1), block(64 * WAVES);
	const size_t lds_bytes = std::max<size_t>(std::max<size_t>(WAVES * ((X3_EPILOGUE_ONE_ROUND && NBW == 2 && !YLDS) ? 8 : 4) * 4 * 64 * sizeof(f32x4), 1024 * sizeof(float)),
	                                          YLDS ? WAVES * 2 * 128 * 20 * sizeof(float) : 0);
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_factor_product_x3<D, WAVES, DIAG, NBW, TR, IMG, YLDS, R32, ODD>), (int)lds_bytes, lds_done); e != hipSuccess) return e;
	if (t_ev_start != nullptr && t_ev_stop != nullptr) {
		// (the caller wants this launch timed: its own start / stop timestamps, no event records around it)
		const hipEvent_t e0 = t_ev_start, e1 = t_ev_stop;
		t_ev_start = t_ev_stop = nullptr;
		hipExtLaunchKernelGGL((k_factor_product_x3<D, WAVES, DIAG, NBW, TR, IMG, YLDS, R32, ODD>), grid, block, (std::uint32_t)lds_bytes, stream, e0, e1, 0u,
		                      A, tile_stride, reinterpret_cast<const bf16x8*>(F), RP / 32, slabs, slab_stride, RP, p.steps_total, p.xtiles, p.splits, with_reduce ? *rg : none, stamps);
		return hipGetLastError();
	}
	hipLaunchKernelGGL((k_factor_product_x3<D, WAVES, DIAG, NBW, TR, IMG, YLDS, R32, ODD>), grid, block, lds_bytes, stream,
	                   A, tile_stride, reinterpret_cast<const bf16x8*>(F), RP / 32, slabs, slab_stride, RP, p.steps_total, p.xtiles, p.splits, with_reduce ? *rg : none, stamps);
	return hipGetLastError();
}

// Pieces of an odd number of K-steps (k_factor_product_x3, ODD) when their longest is shorter than the longest piece of whole turns of the two-deep ring
static bool x3_odd_pieces(int steps_total, int pieces) {
	if (X3_DEAL_TURNS != 0 || X3_RING_X != 2 || pieces <= 0 || steps_total < 3 * pieces) return false;
	const int units = (steps_total + 1) / 2, pairs = (steps_total - pieces + 1) / 2;
	return 2 * ((pairs + pieces - 1) / pieces) + 1 < 2 * ((units + pieces - 1) / pieces);
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
	const bool odd = x3_odd_pieces(p.steps_total, 4 * p.splits);      // (the 64- and 32-column forms; the 128-column ones keep whole turns: no register left for the step ahead of the loop)
	t_ev_start = ev_start; t_ev_stop = ev_stop;
	struct Clear { ~Clear() { t_ev_start = t_ev_stop = nullptr; } } clear_on_exit;      // (a path that did not consume them -- an error return, grid.y > 1 -- leaves nothing behind)
#ifdef NMFAMD_DIAG_BUILD
	if (image_tile == 16 && RP == 64 && stamps != nullptr) {
		// stamped diagnostic builds of the PRODUCTION forms on the one resident image (tools/stamp_x3.py): NMFAMD_X3_VARIANT = 10..13 the x-tiled form,
		// 30..33 the y-tiled form (rows 32 b + r per lane); last digit 0 = no ring refill, 1 = refill A only, 2 = refill F only, 3 = the production loop
		static const int yv = [] { const char* e = tuning_env("NMFAMD_X3_VARIANT"); return e ? std::atoi(e) : 0; }();
		if (!y_tiled) switch (yv % 10) {
			case 0: return (odd ? launch_fp_x3<X3_RING_X, 4, 1, 2, false, 16, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps) : launch_fp_x3<X3_RING_X, 4, 1, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps));
			case 1: return (odd ? launch_fp_x3<X3_RING_X, 4, 2, 2, false, 16, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps) : launch_fp_x3<X3_RING_X, 4, 2, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps));
			case 2: return (odd ? launch_fp_x3<X3_RING_X, 4, 3, 2, false, 16, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps) : launch_fp_x3<X3_RING_X, 4, 3, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps));
			default: return (odd ? launch_fp_x3<X3_RING_X, 4, 4, 2, false, 16, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps) : launch_fp_x3<X3_RING_X, 4, 4, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps));
		}
		switch (yv % 10) {
			case 0: return (odd ? launch_fp_x3<X3_RING_X, 4, 1, 2, true, 16, false, true, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps) : launch_fp_x3<X3_RING_X, 4, 1, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps));
			case 1: return (odd ? launch_fp_x3<X3_RING_X, 4, 2, 2, true, 16, false, true, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps) : launch_fp_x3<X3_RING_X, 4, 2, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps));
			case 2: return (odd ? launch_fp_x3<X3_RING_X, 4, 3, 2, true, 16, false, true, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps) : launch_fp_x3<X3_RING_X, 4, 3, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps));
			default: return (odd ? launch_fp_x3<X3_RING_X, 4, 4, 2, true, 16, false, true, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps) : launch_fp_x3<X3_RING_X, 4, 4, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg, stamps));
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
			if (!y_tiled) return (odd ? launch_fp_x3<X3_RING_X, 4, 0, 1, false, 16, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3<X3_RING_X, 4, 0, 1, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg));
			return (odd ? launch_fp_x3<X3_RING_X, 4, 0, 1, true, 16, false, true, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3<X3_RING_X, 4, 0, 1, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg));
		}
#endif
		if (!y_tiled) return (odd ? launch_fp_x3<X3_RING_X, 4, 0, 2, false, 16, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3<X3_RING_X, 4, 0, 2, false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg));
		if (ylds) return launch_fp_x3<X3_RING_Y, 4, 0, 2, true, 16, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
		if (ydirect) return (odd ? launch_fp_x3<X3_RING_X, 4, 0, 2, true, 16, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3<X3_RING_X, 4, 0, 2, true, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg));
		return (odd ? launch_fp_x3<X3_RING_X, 4, 0, 2, true, 16, false, true, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3<X3_RING_X, 4, 0, 2, true, 16, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg));
	}
	if (y_tiled) {
		if (RP % 128 == 0) return launch_fp_x3<2, 4, 0, 4, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
#ifdef NMFAMD_DIAG_BUILD
		if (RP == 64 && p.col_split == 2) return (odd ? launch_fp_x3<X3_RING_X, 4, 0, 1, true, 128, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3<X3_RING_X, 4, 0, 1, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg));
#endif
		return (odd ? launch_fp_x3<X3_RING_X, 4, 0, 2, true, 128, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3<X3_RING_X, 4, 0, 2, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg));
	}
	static const int variant = [] { const char* e = tuning_env("NMFAMD_X3_VARIANT"); return e ? std::atoi(e) : 0; }();   // A/B switch for measurements
	// short reduction ranges (a column shard's V H^T): 128 x 32 per workgroup, two workgroups per x-tile (FactorProductPlan::col_split)
	if (RP == 64 && p.col_split == 2) return (odd ? launch_fp_x3<X3_RING_X, 4, 0, 1, false, 128, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3<X3_RING_X, 4, 0, 1>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg));
	// wide panels: 128 columns per pass over A (256 accumulator registers, ring depth 2) -- half the passes, MFMA-bound
	if (RP % 128 == 0 && variant != 20) return launch_fp_x3<2, 4, 0, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
#ifdef NMFAMD_DIAG_BUILD
	switch (variant) {
	case 1: return launch_fp_x3<2, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
	case 2: return launch_fp_x3<4, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg);
	default: break;
	}
#endif
	return (odd ? launch_fp_x3<X3_RING_X, 4, 0, 2, false, 128, false, false, true>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg) : launch_fp_x3<X3_RING_X, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, stream, rg));
}

} // namespace nmfamd
