// kernels_f64.hip -- the factor product OUT(c, x) = sum_y F(c, y) A(x, y) in double precision on the fp64 MFMA pipe.
//
// The reference instantiates every algorithm for double as well (include/nmfgpu.h:298-299; the R binding works in
// double); its products are cublasDgemm / Dsyrk / Dsymm (source/common/Matrix.h:314-442).  Same decomposition and the
// same x-tiled image of A as the fp32 kernel (kernels.hip, k_factor_product_f32), with
//   * v_mfma_f64_16x16x4_f64 (64 cycles, 2048 FLOP: the fp64 matrix peak of gfx950 equals its vector peak, 78.6 TFLOP/s,
//     but one MFMA replaces 16 VALU FMAs per lane and needs no LDS / cross-lane traffic for the operands),
//   * workgroup = 8 waves = one 128-row x-tile times one slice of the reduction range; the waves form 2 row halves of
//     64 rows times 4 pieces of the slice; a wave keeps a 64 x 64 block of the output = 4 x 4 MFMA tiles = 128 VGPRs,
//   * K-step = 4 y.  Operand lane maps (A: lane l holds row l & 15, k = l >> 4; B: column l & 15, k = l >> 4); the rows /
//     columns of the four tiles are interleaved (row 4 i + b, column 4 j + nb), so a lane's four A values and its four
//     F values are 32 contiguous bytes each,
//   * the four pieces of a row half are summed through LDS in piece order; one fp64 slab per slice.
// At 8 bytes per element the product is as close to the HBM roof (400 MB per pass at config 2: 67 us at 6 TB/s) as to
// the MFMA roof (6.4 GFLOP: 81 us at peak).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "kernels.h"
#include "tuning.h"

namespace nmfamd {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int F64_TH = 128;

// ------------------------------------------------------------------------------------------
// Gram matrix of a factor panel as PASSENGER workgroups of the fp64 product launch (round 6)
// ------------------------------------------------------------------------------------------
// The generic iteration spent two launches per factor on its Gram matrix (k_gram_f64 + k_gram_reduce_sym_f64: 7.4 + 5.0 us at the reference example's shape, both at
// the launch floor) in FRONT of the product against V, which does not need the result -- its consumer is the update kernel behind that launch.  So the matrix rides
// in the product launch, as it does at padded rank 64 in fp32 (gram_image.h) and at rank 256 in bf16 (tri_gram_tile.h):
//   * one passenger workgroup (8 waves) per 64 x 64 super-block (I <= J) and K slice: the two wave halves take the halves of the slice, a wave a 32 x 32 quarter as
//     2 x 2 MFMA tiles (k_gram_f64<2, .>'s loop); the upper half adds its accumulators to the lower one's through LDS; one partial block per workgroup;
//   * level 1: the workgroup releases its partial block and counts itself in on the super-block's counter; the LAST of the slices adds the partial blocks in slice
//     order (whoever is last, the order is fixed) and writes the block and its mirror image;
//   * W side: RP / 64 more passengers turn the W update's per-workgroup sums of squares into the pending column scale d(c) = 1 / sqrt(sum)
//     (kernel::normalizeColumns, KernelNormalizeColumns.cu:37-58, as a factor).  The matrix stays RAW (W as it lies in its panel: unnormalised, unsmoothed): the
//     H update applies D and nsNMF's S around its r x r product, S D G D S h (PanelFusedF64).  A first version formed S D G D S here, by the last finisher of all
//     super-blocks: one workgroup walking the 192 x 192 matrix twice made the launch 76 us at the reference example's shape (the product alone: 17).
// Nobody waits for anybody: no co-residency assumption.  The counters are left at zero.
// LDS-only barrier: this wave's LDS traffic has completed (lgkmcnt(0)) before it arrives; global stores stay in flight -- __syncthreads() also waits for them
// (vmcnt(0)), which made every round of the product's in-workgroup reduction wait for the previous round's slab stores (2.5 us per round: 10 us of a 19 us launch
// at the reference example's shape, tools/stamp_f64.py)
__device__ __forceinline__ void lds_barrier64() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// measurement builds: stamp k of workgroup `wg` (100 MHz wall clock), first lane of the workgroup only
__device__ __forceinline__ void stamp64(unsigned long long* stamps, int wg, int k) {
	if (stamps != nullptr && threadIdx.x == 0 && wg < 4096) stamps[(long)wg * 8 + k] = wall_clock64();
}

__device__ __forceinline__ bool ride64_arrive(unsigned* counter, unsigned target) {
	__shared__ unsigned s_last;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		s_last = old == target - 1u ? 1u : 0u;
		if (s_last) {
			__hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the next launch finds zero)
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the barrier below holds the other waves until the invalidate has completed)
		}
	}
	__syncthreads();
	return s_last != 0u;
}

// passenger `pid` of a launch of 512-thread workgroups; lds: at least 32 KiB (and 2 * RP + 8 doubles)
__device__ __forceinline__ void gram_ride_f64(const GramRideF64& g, int RP, int pid, double* lds) {
	typedef double f64x2l __attribute__((ext_vector_type(2)));
	typedef double f64x4l __attribute__((ext_vector_type(4)));
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int nb = RP / 64, nsuper = nb * (nb + 1) / 2;
	const int nscale = g.sumsq_part != nullptr ? nb : 0;
	if (g.stop == 1) return;
	const int swg = 2048 + pid;
	stamp64(g.stamps, swg, 0);
	if (pid >= nsuper * g.slices) {
		// ---- scale passenger: 64 columns of the pending column scale from the update kernel's partial sums of squares (8 groups of parts, added in group order)
		const int cb = pid - nsuper * g.slices;
		if (cb >= nscale) return;
		const int c = 64 * cb + lane;
		const int p0 = (int)(((long)g.sumsq_parts * wave) / 8), p1 = (int)(((long)g.sumsq_parts * (wave + 1)) / 8);
		double s = 0.0;
		for (int p = p0; p < p1; p += 8) {
			double v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = g.sumsq_part[(long)(p + u < p1 ? p + u : p0) * RP + c];
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (p + u < p1) s += v[u];
		}
		lds[wave * 64 + lane] = s;
		__syncthreads();
		if (wave == 0) {
			double t = lds[lane];
#pragma unroll
			for (int w = 1; w < 8; ++w) t += lds[w * 64 + lane];
			g.scale_out[c] = t > 0.0 ? 1.0 / sqrt(t) : 1.0;
		}
		return;
	}
	// which (super-block, K slice) this workgroup takes: from the engine's table (XCD-aware: the workgroups an XCD receives take the same K slices, so that every
	// XCD's L2 pulls its slices' rows of the panel once instead of the whole panel -- gram_ride_f64_items) or in pid order
	const int item = g.items != nullptr ? g.items[pid] : pid;
	const int sb = item / g.slices, slice = item - sb * g.slices;
	int I = 0, rem = sb;
	while (rem >= nb - I) { rem -= nb - I; ++I; }
	const int J = I + rem;
	const int l15 = lane & 15, kq = lane >> 4;
	const int q = wave & 3, kh = wave >> 2;
	const int ca = 64 * I + 32 * (q >> 1), cb = 64 * J + 32 * (q & 1);
	const int steps_total = (g.len + 3) / 4;
	const int pieces = 2 * g.slices, piece = 2 * slice + kh;
	const int s0 = (int)(((long)steps_total * piece) / pieces), s1 = (int)(((long)steps_total * (piece + 1)) / pieces);
	const int steps = s1 - s0;
	f64x4l acc[2][2];
#pragma unroll
	for (int a = 0; a < 2; ++a)
#pragma unroll
		for (int b = 0; b < 2; ++b)
#pragma unroll
			for (int gg = 0; gg < 4; ++gg) acc[a][b][gg] = 0.0;
	if (steps > 0) {
#ifndef RIDE64_RING
#define RIDE64_RING 8
#endif
		constexpr int DG = RIDE64_RING;      // (K-steps in flight; sixteen measured no faster)
		const double* pa = g.P + ((long)4 * s0 + kq) * RP + ca + 2 * l15;
		const double* pb = g.P + ((long)4 * s0 + kq) * RP + cb + 2 * l15;
		const long step = 4 * (long)RP;
		const int last = steps - 1;
		f64x2l va[DG], vb[DG];
#pragma unroll
		for (int d = 0; d < DG; ++d) {
			const int t = d < last ? d : last;
			va[d] = *reinterpret_cast<const f64x2l*>(pa + t * step);
			vb[d] = *reinterpret_cast<const f64x2l*>(pb + t * step);
		}
		__builtin_amdgcn_sched_barrier(0);
		int t = 0;
		for (; t + DG <= steps; t += DG) {
#pragma unroll
			for (int d = 0; d < DG; ++d) {
#pragma unroll
				for (int a = 0; a < 2; ++a)
#pragma unroll
					for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[d][a], vb[d][b], acc[a][b], 0, 0, 0);
				int tn = t + DG + d;
				tn = tn < last ? tn : last;
				va[d] = *reinterpret_cast<const f64x2l*>(pa + tn * step);
				vb[d] = *reinterpret_cast<const f64x2l*>(pb + tn * step);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		const int remn = steps - t;
#pragma unroll
		for (int d = 0; d < DG; ++d) {
			if (d < remn) {
#pragma unroll
				for (int a = 0; a < 2; ++a)
#pragma unroll
					for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[d][a], vb[d][b], acc[a][b], 0, 0, 0);
			}
		}
	}
	// the upper K half adds to the lower one through LDS ([quarter][16][64] doubles), which stores the workgroup's partial block:
	// element ((a * 2 + b) * 4 + g) * 64 + lane of quarter q  (C/D map: register g of lane (l15, kq) is tile row kq + 4 g, tile column l15)
	if (kh == 1) {
#pragma unroll
		for (int a = 0; a < 2; ++a)
#pragma unroll
			for (int b = 0; b < 2; ++b)
#pragma unroll
				for (int gg = 0; gg < 4; ++gg) lds[(q * 16 + (a * 2 + b) * 4 + gg) * 64 + lane] = acc[a][b][gg];
	}
	__syncthreads();
	double* out = g.partial + ((long)slice * nsuper + sb) * 4096;
	if (kh == 0) {
#pragma unroll
		for (int a = 0; a < 2; ++a)
#pragma unroll
			for (int b = 0; b < 2; ++b)
#pragma unroll
				for (int gg = 0; gg < 4; ++gg) {
					const int e = (q * 16 + (a * 2 + b) * 4 + gg) * 64 + lane;
					out[e] = acc[a][b][gg] + lds[e];
				}
	}
	stamp64(g.stamps, swg, 1);
	if (g.stop == 2) return;
	if (!ride64_arrive(g.counters + sb, (unsigned)g.slices)) { stamp64(g.stamps, swg, 2); return; }
	stamp64(g.stamps, swg, 2);
	if (g.stop == 3) return;
	// ---- level 1: this workgroup was the last of the block's slices -- add the partial blocks in slice order
	{
		const long pstride = (long)nsuper * 4096;
		const double* pp = g.partial + (long)sb * 4096 + 2 * tid;      // elements 2 tid, 2 tid + 1 of each 1 024-element quarter-half: pairs (e, e + 1) are lanes (l, l + 1)
		f64x2l sum[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) sum[k] = f64x2l{0.0, 0.0};
		// (eight slices = 32 loads of 16 bytes per thread requested at once: 128 registers, what the product's own path leaves without spilling)
		for (int u0 = 0; u0 < g.slices; u0 += 8) {
			f64x2l v[8][4];
#pragma unroll
			for (int u = 0; u < 8; ++u)
#pragma unroll
				for (int k = 0; k < 4; ++k) v[u][k] = *reinterpret_cast<const f64x2l*>(pp + (long)(u0 + u < g.slices ? u0 + u : 0) * pstride + 1024 * k);
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (u0 + u < g.slices) {
#pragma unroll
					for (int k = 0; k < 4; ++k) sum[k] += v[u][k];
				}
		}
		double* G = g.G;
#pragma unroll
		for (int k = 0; k < 4; ++k)
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const int e = 1024 * k + 2 * tid + h;
				const int el = e & 63, eg = (e >> 6) & 3, eab = (e >> 8) & 3, eq = e >> 10;
				const int rr = 64 * I + 32 * (eq >> 1) + 2 * ((el >> 4) + 4 * eg) + (eab >> 1);
				const int cc = 64 * J + 32 * (eq & 1) + 2 * (el & 15) + (eab & 1);
				G[(long)rr * RP + cc] = sum[k][h];
				if (I < J) G[(long)cc * RP + rr] = sum[k][h];
			}
	}
	stamp64(g.stamps, swg, 3);
}

// RH = row halves per workgroup: 2 (the 128-row x-tile: two halves x four pieces of the slice) or -- round 6 -- 1: a workgroup takes ONE 64-row half of an x-tile
// and its eight waves eight pieces of the slice.  Twice the workgroups with half the MFMAs each: for grids that leave more than half of the chip idle (the
// reference example's shape: 96 workgroups of 8 waves = two waves on every SIMD of 96 CUs, 15.6 us of matrix-pipe time per launch for 0.4 GFLOP, tools/stamp_f64.py).
template <int D, int NC, int RH>      // NC = 16-column tiles per wave: 4 (64 panel columns) or 2 (ranks <= 32: the first 32 columns only)
__global__ __launch_bounds__(512, 2) void k_factor_product_f64(
	const double* __restrict__ A, long tile_stride,
	const double* __restrict__ F, int RP,
	double* __restrict__ slabs, long slab_stride,
	int steps_total, int splits, int xtiles, int xhalves, GramRideF64 ride) {
	extern __shared__ __attribute__((aligned(16))) double lds64[];
	if ((int)blockIdx.x >= xtiles) {
		// passenger workgroups behind the x-tiles of every (slice, chunk) row of the grid
		const int extra = (int)gridDim.x - xtiles;
		const int pid = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * extra + ((int)blockIdx.x - xtiles);
		const int nbk = RP / 64;
		if (ride.P != nullptr && pid < (nbk * (nbk + 1) / 2) * ride.slices + nbk) gram_ride_f64(ride, RP, pid, lds64);
		return;
	}
	const int xt = RH == 2 ? (int)blockIdx.x : (int)blockIdx.x >> 1, sp = blockIdx.y;      // (xtiles counts the product workgroups along x: half tiles when RH = 1)
	const int pwg = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * xtiles + (int)blockIdx.x;
	stamp64(ride.stamps, pwg, 0);
	const int coff = 64 * blockIdx.z;             // 64-column chunk of the panel (grid.z = RP / 64)
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int l15 = lane & 15, kq = lane >> 4;
	const int rh = RH == 2 ? wave & 1 : (int)blockIdx.x & 1;      // row half,
	const int kp = RH == 2 ? wave >> 1 : wave;                    // piece of the slice
	constexpr int KP = 8 / RH;

	const int np = splits * KP, pidx = sp * KP + kp;
	const int s0 = (int)(((long)steps_total * pidx) / np);
	const int s1 = (int)(((long)steps_total * (pidx + 1)) / np);
	// (a 64-row half behind the last valid row -- the reference example's H side: 165 rows in two 128-row tiles -- is all padding: no K-steps, its slab rows are zeros)
	const int steps = 2 * xt + rh < xhalves ? s1 - s0 : 0;

	typedef double fvec __attribute__((ext_vector_type(NC)));
	f64x4 acc[4][NC];
#pragma unroll
	for (int b = 0; b < 4; ++b)
#pragma unroll
		for (int nb = 0; nb < NC; ++nb)
#pragma unroll
			for (int g = 0; g < 4; ++g) acc[b][nb][g] = 0.0;

	if (steps > 0) {
		// lane (i, k): rows 64 rh + 4 i .. + 3 of the tile at y = 4 t + k; columns coff + NC j .. + NC - 1 of the panel
		const double* ap = A + (long)xt * tile_stride + (long)(4 * s0 + kq) * F64_TH + 64 * rh + 4 * l15;
		const double* fp = F + (long)(4 * s0 + kq) * RP + coff + NC * l15;
		const long astep = 4 * F64_TH, fstep = 4 * (long)RP;
		const int last = steps - 1;
		f64x4 va[D];
		fvec fb[D];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int st = d < last ? d : last;
			va[d] = *reinterpret_cast<const f64x4*>(ap + st * astep);
			fb[d] = *reinterpret_cast<const fvec*>(fp + st * fstep);
		}
		__builtin_amdgcn_sched_barrier(0);
		int t = 0;
		for (; t + D <= steps; t += D) {
#pragma unroll
			for (int d = 0; d < D; ++d) {
#pragma unroll
				for (int b = 0; b < 4; ++b)
#pragma unroll
					for (int nb = 0; nb < NC; ++nb)
						acc[b][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[d][b], fb[d][nb], acc[b][nb], 0, 0, 0);
				int st = t + D + d;
				st = st < last ? st : last;
				va[d] = *reinterpret_cast<const f64x4*>(ap + st * astep);
				fb[d] = *reinterpret_cast<const fvec*>(fp + st * fstep);
				__builtin_amdgcn_sched_barrier(0);      // refill right behind the MFMAs that consumed the slot
			}
		}
		const int rem = steps - t;
#pragma unroll
		for (int d = 0; d < D; ++d) {
			if (d < rem) {
#pragma unroll
				for (int b = 0; b < 4; ++b)
#pragma unroll
					for (int nb = 0; nb < NC; ++nb)
						acc[b][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[d][b], fb[d][nb], acc[b][nb], 0, 0, 0);
			}
		}
	}

	stamp64(ride.stamps, pwg, 1);
	// ---- sum the four pieces of each row half through LDS, eight tiles per round ------------------------------
	// A round carries BPR = 8 / NC row blocks x NC column tiles of every wave.
	// LDS image of a round: [wave 8][tile 8][pair 2][lane 64] f64x2   (128 KiB)
	// C/D map of the 16x16 fp64 MFMA (measured: tests/test_gpu_parity.py identity-layout check): register g of lane l is
	// row (l >> 4) + 4 g, column l & 15.
	constexpr int BPR = 8 / NC, ROUNDS = 4 / BPR;
	f64x2* l2 = reinterpret_cast<f64x2*>(lds64);
	double* slab = slabs + (long)sp * slab_stride;
#pragma unroll
	for (int rd = 0; rd < ROUNDS; ++rd) {
		if (rd > 0) lds_barrier64();
#pragma unroll
		for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
			for (int pr = 0; pr < 2; ++pr) {
				const int b = BPR * rd + t8 / NC, nb = t8 % NC;
				f64x2 v;
				v[0] = acc[b][nb][2 * pr]; v[1] = acc[b][nb][2 * pr + 1];
				l2[(((wave * 8) + t8) * 2 + pr) * 64 + lane] = v;
			}
		lds_barrier64();
		stamp64(ride.stamps, pwg, 3 + rd);
		if (RH == 1) {
			// 16 (tile, register pair) slices per round, eight pieces each; a wave takes two neighbouring column tiles of one (row block, pair):
			//   NC = 4: row block w >> 2 of the round, pair (w >> 1) & 1, tiles 2 (w & 1) and 2 (w & 1) + 1;   NC = 2: row block w >> 1, pair w & 1, both tiles
			const int obl = NC == 4 ? wave >> 2 : wave >> 1;
			const int opr = NC == 4 ? (wave >> 1) & 1 : wave & 1;
			const int nb0 = NC == 4 ? 2 * (wave & 1) : 0;
			f64x2 sum[2];
#pragma unroll
			for (int q = 0; q < 2; ++q) {
				const int t8 = obl * NC + nb0 + q;
				f64x2 s = l2[((0 * 8 + t8) * 2 + opr) * 64 + lane];
#pragma unroll
				for (int p = 1; p < 8; ++p) s += l2[((p * 8 + t8) * 2 + opr) * 64 + lane];
				sum[q] = s;
			}
			const int b = BPR * rd + obl;
#pragma unroll
			for (int gg = 0; gg < 2; ++gg) {
				const int i = kq + 4 * (2 * opr + gg);
				const int x = xt * F64_TH + 64 * rh + 4 * i + b;
				f64x2 o;
				o[0] = sum[0][gg]; o[1] = sum[1][gg];
				*reinterpret_cast<f64x2*>(slab + (long)x * RP + coff + NC * l15 + nb0) = o;
			}
		} else {
			// 32 (row half, tile, register pair) slices per round, four per wave: all NC column tiles of
			//   NC = 4: (row half w & 1, row block (w >> 1) & 1 of the round, pair w >> 2)
			//   NC = 2: (row half w & 1, pair (w >> 1) & 1, row blocks 2 (w >> 2) and 2 (w >> 2) + 1 of the round)
			const int orh = wave & 1;
			const int opr = NC == 4 ? wave >> 2 : (wave >> 1) & 1;
#pragma unroll
			for (int k = 0; k < 4 / NC; ++k) {
				const int obl = NC == 4 ? (wave >> 1) & 1 : 2 * (wave >> 2) + k;      // row block inside the round
				f64x2 sum[NC];
#pragma unroll
				for (int nb = 0; nb < NC; ++nb) {
					const int t8 = obl * NC + nb;
					f64x2 s = l2[((((0 * 2 + orh) * 8) + t8) * 2 + opr) * 64 + lane];
#pragma unroll
					for (int p = 1; p < 4; ++p) s += l2[((((p * 2 + orh) * 8) + t8) * 2 + opr) * 64 + lane];
					sum[nb] = s;
				}
				const int b = BPR * rd + obl;
#pragma unroll
				for (int gg = 0; gg < 2; ++gg) {
					const int i = kq + 4 * (2 * opr + gg);                  // MFMA row of this value
					const int x = xt * F64_TH + 64 * orh + 4 * i + b;
					fvec o;
#pragma unroll
					for (int nb = 0; nb < NC; ++nb) o[nb] = sum[nb][gg];
					*reinterpret_cast<fvec*>(slab + (long)x * RP + coff + NC * l15) = o;
				}
			}
		}
	}
	stamp64(ride.stamps, pwg, 2);
}

// Reduction length in K-steps of four y; as many slices as fill the CUs, at least 16 K-steps per wave piece.
FactorProductPlan plan_factor_product_f64(int X, int Y, int RP, int num_cus) {
	FactorProductPlan p;
	p.th = F64_TH;
	p.xtiles = (X + F64_TH - 1) / F64_TH;
	p.xhalves = (X + 63) / 64;
	p.steps_total = (Y + 3) / 4;
	int min_steps = 16;
	if (const char* e = tuning_env("NMFAMD_F64_MIN_STEPS")) { if (std::atoi(e) > 0) min_steps = std::atoi(e); }
	int max_splits = p.steps_total / (min_steps * 4);
	if (max_splits < 1) max_splits = 1;
	p.splits = std::max(1, std::min(num_cus / std::max(1, p.xtiles), max_splits));
	p.nb = 4;
	p.chunks = RP / 64;
	// Half-tile workgroups (RH = 1: a 64-row half of an x-tile, eight pieces of the K slice per workgroup) where whole tiles would leave more than 60 % of the chip idle:
	// twice the workgroups along x and, with eight pieces instead of four, HALF the K slices for the same K-steps per wave -- half the slabs the update kernel
	// behind the launch adds.  Reference example (H side: 2 x-tiles x 3 chunks, 1 024 K-steps): 16 slices x 6 = 96 workgroups of 16 K-steps per wave (two waves
	// per SIMD: 15.6 us of matrix pipe) -> 8 slices x 12 = 96 workgroups, the same pipe time, 8 slabs; W side (32 x-tiles, 42 K-steps): 96 -> 192 workgroups
	// of 5 K-steps per wave.  76.0 -> 70.2 us per iteration (profiles/r06_f64_example.md).  The Gram passengers of the fused iteration take the CUs left over.
	p.half_tiles = (10 * p.xtiles * p.splits * p.chunks <= 4 * num_cus && p.steps_total / 8 >= 4) ? 1 : 0;
	if (const char* e = tuning_env("NMFAMD_F64_HALF_TILES")) p.half_tiles = std::atoi(e) != 0 ? 1 : 0;
	if (p.half_tiles) {
		// (more, shorter slices -- thirteen at the reference example's H side, 156 + 99 workgroups -- measured slower: not every workgroup of a grid that size is
		//  resident at once, the late ones end the launch)
		max_splits = std::max(1, p.steps_total / (min_steps * 8));
		p.splits = std::max(1, std::min(num_cus / std::max(1, p.xhalves * p.chunks), max_splits));
	}
	return p;
}

// A: x-tiled image (launch_tile<double>, tile height 128, the reduction length padded to a multiple of 4 with zeros);
// F: panel [y][RP]; slabs: plan.splits partial results, panel layout [x][RP].
// (the RP / 64 scale passengers always have their places in the grid: one layout per engine, whether a scale is pending or not)
int gram_ride_f64_workgroups(int RP, int slices) {
	const int nb = RP / 64;
	return (nb * (nb + 1) / 2) * slices + nb;
}

static void ride64_grid(const FactorProductPlan& p, int RP, int slices, int* xblocks, int* extra) {
	*xblocks = p.half_tiles ? 2 * p.xtiles : p.xtiles;
	const int pass = gram_ride_f64_workgroups(RP, slices), rows = p.splits * p.chunks;
	*extra = (pass + rows - 1) / rows;
}

// items[pid] = super-block * slices + slice for the passengers pid < super-blocks * slices of the launch launch_factor_product_f64(p, ..., ride) makes.
// Model: the hardware deals the workgroups of a grid to the eight XCDs in linear order (x fastest), workgroup L to XCD L mod 8 (what the KL gather and the
// split-operand product's placement rely on too).  SUPER-BLOCK b belongs to XCD b mod 8: an XCD's passengers take the slices of its super-blocks first, what is
// left over goes to the XCDs with passengers to spare -- the last arriver of a super-block then finds most of the partial blocks it adds in its own XCD's L2
// (plain-stored slabs are read at 104 - 122 GB/s per workgroup from the same XCD against 62 - 70 across XCDs: the guide's split-K figures; the reduction of
// sixteen 32 KB partial blocks is what ends the H-side launch at the reference example's shape).  A first table kept a K SLICE on one XCD (every L2 pulling only
// its slices' rows of the panel): no change -- the passengers' loop is bound by the fp64 matrix pipe.  Only the speed depends on the model being right, never
// the result: every item is taken exactly once.
void gram_ride_f64_items(const FactorProductPlan& p, int RP, int slices, std::vector<int>& items) {
	const int nbk = RP / 64, nsuper = nbk * (nbk + 1) / 2, count = nsuper * slices;
	int xblocks, extra;
	ride64_grid(p, RP, slices, &xblocks, &extra);
	const int GX = xblocks + extra, GY = p.splits;
	items.assign((size_t)count, -1);
	std::vector<std::vector<int>> wgs(8), work(8);
	for (int pid = 0; pid < count; ++pid) {
		const int row = pid / extra, x = xblocks + pid % extra;      // row = z * GY + y
		const long L = (long)x + (long)GX * ((row % GY) + (long)GY * (row / GY));
		wgs[(int)(L % 8)].push_back(pid);
	}
	for (int sb = 0; sb < nsuper; ++sb)
		for (int sl = 0; sl < slices; ++sl) work[sb % 8].push_back(sb * slices + sl);
	std::vector<int> spare_wgs, spare_work;
	for (int x = 0; x < 8; ++x) {
		const size_t k = std::min(wgs[x].size(), work[x].size());
		for (size_t i = 0; i < k; ++i) items[(size_t)wgs[x][i]] = work[x][i];
		for (size_t i = k; i < wgs[x].size(); ++i) spare_wgs.push_back(wgs[x][i]);
		for (size_t i = k; i < work[x].size(); ++i) spare_work.push_back(work[x][i]);
	}
	for (size_t i = 0; i < spare_wgs.size() && i < spare_work.size(); ++i) items[(size_t)spare_wgs[i]] = spare_work[i];
}

hipError_t launch_factor_product_f64(const FactorProductPlan& p, const double* A, long tile_stride, const double* F, int RP,
                                     double* slabs, long slab_stride, hipStream_t stream, const GramRideF64* ride_in) {
	constexpr int D = 6;
	if (p.th != F64_TH || RP % 64 != 0) return hipErrorInvalidValue;
	const size_t lds_bytes = 8 * 8 * 2 * 64 * sizeof(f64x2);
	GramRideF64 ride = {};
	int extra = 0;
	if (ride_in != nullptr && ride_in->P != nullptr) {
		if (ride_in->slices < 1 || RP > 512) return hipErrorInvalidValue;
		ride = *ride_in;
	}
	int xblocks = p.half_tiles ? 2 * p.xtiles : p.xtiles;
	if (ride.P != nullptr) ride64_grid(p, RP, ride.slices, &xblocks, &extra);
	dim3 grid(xblocks + extra, p.splits, p.chunks), block(512);
#define NMFAMD_F64_PRODUCT(NCV, RHV)                                                                                                                       \
	do {                                                                                                                                                  \
		static std::atomic<unsigned long long> done{0ull};                                                                                                \
		if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_factor_product_f64<D, NCV, RHV>), (int)lds_bytes, done); e != hipSuccess) return e; \
		hipLaunchKernelGGL((k_factor_product_f64<D, NCV, RHV>), grid, block, lds_bytes, stream, A, tile_stride, F, RP, slabs, slab_stride, p.steps_total, p.splits, xblocks, p.xhalves > 0 ? p.xhalves : 2 * p.xtiles, ride); \
		return hipGetLastError();                                                                                                                         \
	} while (0)
	// ranks <= 32 (nb == 2): the first 32 panel columns only (the rest of every slab stays at its initial zeros)
	if (p.nb == 2 && RP == 64) { if (p.half_tiles) NMFAMD_F64_PRODUCT(2, 1); else NMFAMD_F64_PRODUCT(2, 2); }
	if (p.half_tiles) NMFAMD_F64_PRODUCT(4, 1);
	NMFAMD_F64_PRODUCT(4, 2);
#undef NMFAMD_F64_PRODUCT
}


// ------------------------------------------------------------------------------------------
// panel update at padded rank 64 in double precision: slab reduction + r x r product on the fp64 MFMA pipe +
// element-wise update + error / norm partial sums (semantics of k_panel_update, kernels.hip, MODE_MU / MODE_LS;
// reference: symm/gemm + kernel::multiplyDivide, AlgorithmMultiplicativeFrobenius.h:181-191,235-244)
// ------------------------------------------------------------------------------------------
// Workgroup = 4 waves = 32 panel rows y, staged in LDS as [32][68] (coalesced global traffic both ways).
// D(c, y) = sum_k Q(k, c) vec(y, k): wave w owns the 16 columns c = 16 w + i and both 16-row y tiles;
//   A operand: lane (i = l & 15, kq = l >> 4) holds Q(4 t + kq, 16 w + i)  -- 128 contiguous bytes per kq, from L2;
//   B operand: lane (j = l & 15, kq) holds vec(16 yt + j, 4 t + kq)        -- LDS;
//   C/D: register g of lane (j, kq) is column c = 16 w + kq + 4 g of row y = 16 yt + j.
// Extras of the fused double-precision iteration (round 6; PanelFusedF64, kernels.h; template flag HS = the H update with W = Wt D S), applied to the [YB][LD]
// LDS images of a workgroup's panel rows by all 256 threads (TPR = 256 / YB threads per row, columns sub, sub + TPR, ...):
//   fused64_h_prepare:  s_dv(c) = d(c) on the first r_eff entries, zero behind them; s_rs(y) = sum of the old row's first r entries (smoothing only);
//                       num(y, c) <- d(c) num(y, c), then nsNMF's S on the first r entries of the row (k_smooth_panel's formula) -- (Wt D S)^T V = S D (Wt^T V)
//   the r x r product:  B operand u = D S old, formed as the value leaves LDS: u(k) = s_dv(k) ((diag - off) old(k) + off s_rs);  t = Q u on the MFMA pipe;
//                       den = S D t: v(c) = s_dv(c) t(c), den(c) = off (sigma - v(c)) + diag v(c), sigma = the row's sum of v over all column tiles (through LDS)
//                       -- S D (Wt^T Wt) D S old = (W S)^T (W S) old, AlgorithmNonSmoothNMF.h:175-177, without a pass over the r x r matrix
//   fused64_smooth_out: out(y, c) = S new(y, :) -- the smoothed panel the next product and its Gram passengers read (AlgorithmNonSmoothNMF.h:194)
// s_dv, filled at the kernel's start next to the slab loads (a first version read the scale from global memory inside fused64_h_prepare's loop: RP / TPR dependent
// loads per thread, 3 us of the H update at the reference example's shape)
__device__ __forceinline__ void fused64_fill_scale(double* s_dv, int RP, const PanelFusedF64& fx) {
	const int r_eff = fx.smooth ? fx.r : RP;
	for (int c = threadIdx.x; c < RP; c += 256) s_dv[c] = c < r_eff ? (fx.scale != nullptr ? fx.scale[c] : 1.0) : 0.0;
}
template <int YB>
__device__ __forceinline__ void fused64_h_prepare(double* s_num, const double* s_old, double* s_dv, double* s_rs, int LD, int RP, const PanelFusedF64& fx) {
	constexpr int TPR = 256 / YB;
	const int r_eff = fx.smooth ? fx.r : RP;
	const int y = threadIdx.x / TPR, sub = threadIdx.x % TPR;
	double* row = s_num + y * LD;
	double sum = 0.0, osum = 0.0;
	for (int c = sub; c < RP; c += TPR) {
		const double x = row[c] * s_dv[c];      // (zero from r_eff on)
		row[c] = x;
		sum += x;
		if (c < r_eff) osum += s_old[y * LD + c];
	}
	if (fx.smooth) {
#pragma unroll
		for (int w = 1; w < TPR; w <<= 1) { sum += __shfl_xor(sum, w); osum += __shfl_xor(osum, w); }
		for (int c = sub; c < fx.r; c += TPR) { const double x = row[c]; row[c] = fx.off * (sum - x) + fx.diag * x; }
		if (sub == 0) s_rs[y] = osum;
	} else if (sub == 0) s_rs[y] = 0.0;
}
// the trace workgroups behind a W update's panel workgroups (PanelFusedF64::trace_*): one wave per term, lanes stride the inner index, butterfly sum --
// k_trace_small's order (kernels.hip), hence its bits
__device__ __forceinline__ void fused64_trace(const PanelFusedF64& fx, int RP, int block) {
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int d = block * 4 + wave;
	if (d >= fx.trace_r) return;
	double s = 0.0;
	if (fx.trace_scale != nullptr) {
		const double fd = fx.trace_scale[d];
		for (int i = lane; i < fx.trace_r; i += 64) s += fx.trace_a[(long)i * RP + d] * (fx.trace_b[(long)d * RP + i] * (fd * fx.trace_scale[i]));
	} else {
		for (int i = lane; i < fx.trace_r; i += 64) s += fx.trace_a[(long)i * RP + d] * fx.trace_b[(long)d * RP + i];
	}
	for (int w = 32; w > 0; w >>= 1) s += __shfl_xor(s, w);
	if (lane == 0) fx.trace_out[d] = s;
}

template <int YB>
__device__ __forceinline__ void fused64_smooth_out(const double* s_new, int LD, int RP, long base, const PanelFusedF64& fx) {
	constexpr int TPR = 256 / YB;
	const int y = threadIdx.x / TPR, sub = threadIdx.x % TPR;
	const double* row = s_new + y * LD;
	double sum = 0.0;
	for (int c = sub; c < fx.r; c += TPR) sum += row[c];
#pragma unroll
	for (int w = 1; w < TPR; w <<= 1) sum += __shfl_xor(sum, w);
	double* o = fx.smooth_out + base + (long)y * RP;
	for (int c = sub; c < RP; c += TPR) { const double x = row[c]; o[c] = c < fx.r ? fx.off * (sum - x) + fx.diag * x : 0.0; }
}

template <int MODE, bool HS>
__global__ __launch_bounds__(256, 2) void k_panel_update64_f64(
	double* __restrict__ P, const double* __restrict__ slabs, int S, long slab_stride,
	const double* __restrict__ Q, double eps, double* __restrict__ ps, int len_valid,
	double* __restrict__ sumsq_part, double* __restrict__ num_out, PanelFusedF64 fx) {
	constexpr int YB = 32, LD = 68;
	if (fx.trace_out != nullptr && (int)blockIdx.x >= (int)gridDim.x - (fx.trace_r + 3) / 4) { fused64_trace(fx, 64, (int)blockIdx.x - ((int)gridDim.x - (fx.trace_r + 3) / 4)); return; }
	__shared__ __attribute__((aligned(16))) double s_num[YB * LD];
	__shared__ __attribute__((aligned(16))) double s_old[YB * LD];      // old values, then the new ones
	__shared__ double s_ps[4][YB];
	__shared__ double s_sig[HS ? 4 : 1][YB];
	__shared__ double s_dv[HS ? 64 : 1];
	__shared__ double s_rs[HS ? YB : 1];
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int l15 = lane & 15, kq = lane >> 4;
	const long base = (long)blockIdx.x * YB * 64;

	// A operands of the whole product, requested next to the panel loads: one L2 latency
	double qa[16];
#pragma unroll
	for (int t = 0; t < 16; ++t) qa[t] = Q[(long)(4 * t + kq) * 64 + 16 * wave + l15];

	if (HS) fused64_fill_scale(s_dv, 64, fx);
	// numerator = sum of the split-K slabs (slab order), old panel values: four 16-byte pieces per thread and array
	{
		f64x2 num[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) num[i] = *reinterpret_cast<const f64x2*>(slabs + base + 2l * (tid + 256 * i));
		if (MODE == PANEL_MU) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int e = tid + 256 * i, y = e >> 5, c2 = e & 31;
				f64x2 o = *reinterpret_cast<const f64x2*>(P + base + 2l * e);
				if (fx.old_scale != nullptr) o *= *reinterpret_cast<const f64x2*>(fx.old_scale + 2 * c2);      // (the panel's own pending column scale)
				*reinterpret_cast<f64x2*>(s_old + y * LD + 2 * c2) = o;
			}
		}
		// (four slabs requested at a time, added in slab order: one slab per round trip made the H update of a short, wide problem -- 16 slabs at the reference
		//  example's shape -- a chain of 15 dependent loads)
		for (int k = 1; k < S; k += 4) {
			f64x2 t[4][4];
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int i = 0; i < 4; ++i) t[u][i] = *reinterpret_cast<const f64x2*>(slabs + (long)(k + u < S ? k + u : 0) * slab_stride + base + 2l * (tid + 256 * i));
#pragma unroll
			for (int u = 0; u < 4; ++u)
				if (k + u < S) {
#pragma unroll
					for (int i = 0; i < 4; ++i) num[i] += t[u][i];
				}
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int e = tid + 256 * i, y = e >> 5, c2 = e & 31;
			*reinterpret_cast<f64x2*>(s_num + y * LD + 2 * c2) = num[i];
			if (num_out) *reinterpret_cast<f64x2*>(num_out + base + 2l * e) = num[i];
		}
	}
	__syncthreads();
	if (HS) {
		fused64_h_prepare<YB>(s_num, s_old, s_dv, s_rs, LD, 64, fx);
		__syncthreads();
	}

	const double* vec = (MODE == PANEL_MU ? s_old : s_num) + l15 * LD + kq;
	f64x4 acc[2];
#pragma unroll
	for (int yt = 0; yt < 2; ++yt)
#pragma unroll
		for (int g = 0; g < 4; ++g) acc[yt][g] = 0.0;
	const double h_a = HS ? fx.diag - fx.off : 1.0, h_b0 = HS ? fx.off * s_rs[l15] : 0.0, h_b1 = HS ? fx.off * s_rs[16 + l15] : 0.0;
#pragma unroll
	for (int t = 0; t < 16; ++t) {
		double b0 = vec[4 * t], b1 = vec[16 * LD + 4 * t];
		if (HS) { const double dk = s_dv[4 * t + kq]; b0 = dk * (h_a * b0 + h_b0); b1 = dk * (h_a * b1 + h_b1); }
		acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t], b0, acc[0], 0, 0, 0);
		acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t], b1, acc[1], 0, 0, 0);
	}
	if (HS) {
		// den = S D t
		double sig[2] = {0.0, 0.0};
#pragma unroll
		for (int yt = 0; yt < 2; ++yt)
#pragma unroll
			for (int g = 0; g < 4; ++g) { acc[yt][g] *= s_dv[16 * wave + kq + 4 * g]; sig[yt] += acc[yt][g]; }
		if (fx.smooth) {
#pragma unroll
			for (int yt = 0; yt < 2; ++yt) {
				double v = sig[yt];
				v += __shfl_xor(v, 16);
				v += __shfl_xor(v, 32);
				if (kq == 0) s_sig[wave][16 * yt + l15] = v;
			}
			__syncthreads();
#pragma unroll
			for (int yt = 0; yt < 2; ++yt) {
				const int y = 16 * yt + l15;
				const double sigma = ((s_sig[0][y] + s_sig[1][y]) + s_sig[2][y]) + s_sig[3][y];
#pragma unroll
				for (int g = 0; g < 4; ++g) acc[yt][g] = fx.off * (sigma - acc[yt][g]) + fx.diag * acc[yt][g];
			}
		}
	}

	double nv[2][4];
	double psum[2] = {0.0, 0.0};
#pragma unroll
	for (int yt = 0; yt < 2; ++yt)
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const int off = (16 * yt + l15) * LD + 16 * wave + kq + 4 * g;
			const double num = s_num[off];
			double o;
			if (MODE == PANEL_MU) o = s_old[off] * num / (acc[yt][g] + eps);
			else o = acc[yt][g] > 0.0 ? acc[yt][g] : 0.0;
			psum[yt] += o * num;
			nv[yt][g] = o;
		}
	__syncthreads();      // every wave has finished reading the old values as B operands
#pragma unroll
	for (int yt = 0; yt < 2; ++yt)
#pragma unroll
		for (int g = 0; g < 4; ++g) s_old[(16 * yt + l15) * LD + 16 * wave + kq + 4 * g] = nv[yt][g];
#pragma unroll
	for (int yt = 0; yt < 2; ++yt) {
		double v = psum[yt];
		v += __shfl_xor(v, 16);
		v += __shfl_xor(v, 32);
		if (kq == 0) s_ps[wave][16 * yt + l15] = v;
	}
	__syncthreads();

#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int e = tid + 256 * i, y = e >> 5, c2 = e & 31;
		*reinterpret_cast<f64x2*>(P + base + 2l * e) = *reinterpret_cast<const f64x2*>(s_old + y * LD + 2 * c2);
	}
	if (HS && fx.smooth_out != nullptr) fused64_smooth_out<YB>(s_old, LD, 64, base, fx);
	if (ps != nullptr && tid < YB) {
		const int y = blockIdx.x * YB + tid;
		if (y < len_valid) ps[y] = ((s_ps[0][tid] + s_ps[1][tid]) + s_ps[2][tid]) + s_ps[3][tid];
	}
	if (sumsq_part != nullptr && tid < 64) {
		double s = 0.0;
#pragma unroll 8
		for (int y = 0; y < YB; ++y) { const double v = s_old[y * LD + tid]; s += v * v; }
		sumsq_part[(long)blockIdx.x * 64 + tid] = s;
	}
}

// The same at padded ranks 128 ... 512 (the reference's example program runs r = 158 in double): 16 panel rows per
// workgroup as [16][RP + 4] LDS images, wave w owns the 16-column tiles ct = w, w + 4, ... (NCT of them); the A operand
// (Q, RP x RP, L2-resident) comes through a register ring, the B operand is one LDS read per K-step shared by the
// wave's NCT tiles.
template <int MODE, int NCT, bool HS>
__global__ __launch_bounds__(256, 2) void k_panel_update_wide_f64(
	double* __restrict__ P, const double* __restrict__ slabs, int S, long slab_stride,
	const double* __restrict__ Q, int RP, double eps, double* __restrict__ ps, int len_valid,
	double* __restrict__ sumsq_part, double* __restrict__ num_out, PanelFusedF64 fx) {
	extern __shared__ __attribute__((aligned(16))) double ldsw[];
	constexpr int YB = 16;
	if (fx.trace_out != nullptr && (int)blockIdx.x >= (int)gridDim.x - (fx.trace_r + 3) / 4) { fused64_trace(fx, RP, (int)blockIdx.x - ((int)gridDim.x - (fx.trace_r + 3) / 4)); return; }
	const int LD = RP + 4;
	double* s_num = ldsw;                  // [16][LD]
	double* s_old = ldsw + YB * LD;        // [16][LD]
	double* s_ps = s_old + YB * LD;        // [4][16]
	double* s_sig = s_ps + 64;             // HS: [4][16]
	double* s_rs = s_sig + 64;             // HS: [16]
	double* s_dv = s_rs + 16;              // HS: [RP]
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int l15 = lane & 15, kq = lane >> 4;
	const long base = (long)blockIdx.x * YB * RP;
	const int h2 = RP / 2;                 // 16-byte pieces per panel row
	constexpr int NE = YB * 32 * NCT / 256;   // pieces per thread: 16 rows * (64 NCT / 2) / 256 threads = 2 NCT
	stamp64(fx.stamps, blockIdx.x, 0);

	if (HS) fused64_fill_scale(s_dv, RP, fx);
	{
		f64x2 num[NE];
#pragma unroll
		for (int i = 0; i < NE; ++i) num[i] = *reinterpret_cast<const f64x2*>(slabs + base + 2l * (tid + 256 * i));
		if (MODE == PANEL_MU) {
#pragma unroll
			for (int i = 0; i < NE; ++i) {
				const int e = tid + 256 * i, y = e / h2, c2 = e - y * h2;
				f64x2 o = *reinterpret_cast<const f64x2*>(P + base + 2l * e);
				if (fx.old_scale != nullptr) o *= *reinterpret_cast<const f64x2*>(fx.old_scale + 2 * c2);      // (the panel's own pending column scale)
				*reinterpret_cast<f64x2*>(s_old + y * LD + 2 * c2) = o;
			}
		}
		// (SB slabs requested at a time, added in slab order -- see k_panel_update64_f64)
#ifndef WIDE64_SB
#define WIDE64_SB (NE <= 4 ? 8 : (NE <= 8 ? 4 : 2))      /* at most 32 sixteen-byte loads (128 registers) in flight; 48 at the narrow panels measured the same */
#endif
		constexpr int SB = WIDE64_SB;
		for (int k = 1; k < S; k += SB) {
			f64x2 t[SB][NE];
#pragma unroll
			for (int u = 0; u < SB; ++u)
#pragma unroll
				for (int i = 0; i < NE; ++i) t[u][i] = *reinterpret_cast<const f64x2*>(slabs + (long)(k + u < S ? k + u : 0) * slab_stride + base + 2l * (tid + 256 * i));
#pragma unroll
			for (int u = 0; u < SB; ++u)
				if (k + u < S) {
#pragma unroll
					for (int i = 0; i < NE; ++i) num[i] += t[u][i];
				}
		}
#pragma unroll
		for (int i = 0; i < NE; ++i) {
			const int e = tid + 256 * i, y = e / h2, c2 = e - y * h2;
			*reinterpret_cast<f64x2*>(s_num + y * LD + 2 * c2) = num[i];
			if (num_out) *reinterpret_cast<f64x2*>(num_out + base + 2l * e) = num[i];
		}
	}
	__syncthreads();
	stamp64(fx.stamps, blockIdx.x, 1);
	if (HS) {
		fused64_h_prepare<YB>(s_num, s_old, s_dv, s_rs, LD, RP, fx);
		__syncthreads();
	}
	stamp64(fx.stamps, blockIdx.x, 2);

	const double* vec = (MODE == PANEL_MU ? s_old : s_num) + l15 * LD + kq;
	f64x4 acc[NCT];
#pragma unroll
	for (int i = 0; i < NCT; ++i)
#pragma unroll
		for (int g = 0; g < 4; ++g) acc[i][g] = 0.0;
	const double* qp = Q + (long)kq * RP + 16 * wave + l15;      // + 4 t RP per K-step, + 64 i per tile
	const int steps = RP / 4;                                     // multiple of 16 (RP of 64): whole turns of the ring of eight
#ifndef WIDE64_RING
#define WIDE64_RING ((HS && NCT >= 7) ? 4 : 8)
#endif
	constexpr int D = WIDE64_RING;                                // (the H-side form at 448 / 512 columns: a shorter ring instead of spills; sixteen K-steps in flight at narrow panels measured +1 us)
	const double h_a = HS ? fx.diag - fx.off : 1.0, h_b = HS ? fx.off * s_rs[l15] : 0.0;
	const double* dvp = s_dv + kq;
	auto bval = [&](int t) { double h = vec[4 * t]; if (HS) h = dvp[4 * t] * (h_a * h + h_b); return h; };
	double a[D][NCT], b[D];
#pragma unroll
	for (int d = 0; d < D; ++d) {
		b[d] = bval(d);
#pragma unroll
		for (int i = 0; i < NCT; ++i) a[d][i] = qp[(long)(4 * d) * RP + 64 * i];
	}
	__builtin_amdgcn_sched_barrier(0);
	for (int t = 0; t < steps; t += D) {
#pragma unroll
		for (int d = 0; d < D; ++d) {
#pragma unroll
			for (int i = 0; i < NCT; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d][i], b[d], acc[i], 0, 0, 0);
			int tn = t + D + d;
			tn = tn < steps ? tn : steps - 1;
			b[d] = bval(tn);
#pragma unroll
			for (int i = 0; i < NCT; ++i) a[d][i] = qp[(long)(4 * tn) * RP + 64 * i];
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	stamp64(fx.stamps, blockIdx.x, 3);
	if (HS) {
		// den = S D t
		double sig = 0.0;
#pragma unroll
		for (int i = 0; i < NCT; ++i)
#pragma unroll
			for (int g = 0; g < 4; ++g) { acc[i][g] *= s_dv[16 * (wave + 4 * i) + kq + 4 * g]; sig += acc[i][g]; }
		if (fx.smooth) {
			sig += __shfl_xor(sig, 16);
			sig += __shfl_xor(sig, 32);
			if (kq == 0) s_sig[wave * YB + l15] = sig;
			__syncthreads();
			const double sigma = ((s_sig[l15] + s_sig[YB + l15]) + s_sig[2 * YB + l15]) + s_sig[3 * YB + l15];
#pragma unroll
			for (int i = 0; i < NCT; ++i)
#pragma unroll
				for (int g = 0; g < 4; ++g) acc[i][g] = fx.off * (sigma - acc[i][g]) + fx.diag * acc[i][g];
		}
	}

	double nv[NCT][4];
	double psum = 0.0;
#pragma unroll
	for (int i = 0; i < NCT; ++i)
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const int off = l15 * LD + 16 * (wave + 4 * i) + kq + 4 * g;
			const double num = s_num[off];
			double o;
			if (MODE == PANEL_MU) o = s_old[off] * num / (acc[i][g] + eps);
			else o = acc[i][g] > 0.0 ? acc[i][g] : 0.0;
			psum += o * num;
			nv[i][g] = o;
		}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < NCT; ++i)
#pragma unroll
		for (int g = 0; g < 4; ++g) s_old[l15 * LD + 16 * (wave + 4 * i) + kq + 4 * g] = nv[i][g];
	psum += __shfl_xor(psum, 16);
	psum += __shfl_xor(psum, 32);
	if (kq == 0) s_ps[wave * YB + l15] = psum;
	__syncthreads();
	stamp64(fx.stamps, blockIdx.x, 4);

#pragma unroll
	for (int i = 0; i < NE; ++i) {
		const int e = tid + 256 * i, y = e / h2, c2 = e - y * h2;
		*reinterpret_cast<f64x2*>(P + base + 2l * e) = *reinterpret_cast<const f64x2*>(s_old + y * LD + 2 * c2);
	}
	if (HS && fx.smooth_out != nullptr) fused64_smooth_out<YB>(s_old, LD, RP, base, fx);
	if (ps != nullptr && tid < YB) {
		const int y = blockIdx.x * YB + tid;
		if (y < len_valid) ps[y] = ((s_ps[tid] + s_ps[YB + tid]) + s_ps[2 * YB + tid]) + s_ps[3 * YB + tid];
	}
	if (sumsq_part != nullptr) {
		for (int c = tid; c < RP; c += 256) {
			double s = 0.0;
#pragma unroll
			for (int y = 0; y < YB; ++y) { const double v = s_old[y * LD + c]; s += v * v; }
			sumsq_part[(long)blockIdx.x * RP + c] = s;
		}
	}
	stamp64(fx.stamps, blockIdx.x, 5);
}

bool panel_update_wide_f64_available(int RP) { return RP >= 128 && RP % 64 == 0 && RP <= 512; }

template <int MODE, int NCT, bool HS>
static hipError_t launch_wide_f64(double* P, const double* slabs, int S, long slab_stride, const double* Q, int RP, int len_pad,
                                  double eps, double* ps, int len_valid, double* sumsq_part, double* num_out, hipStream_t stream, const PanelFusedF64& fx) {
	const size_t lds_bytes = sizeof(double) * (2 * 16 * (size_t)(RP + 4) + 64 + (HS ? 64 + 16 + (size_t)RP : 0));
	const size_t max_bytes = sizeof(double) * (2 * 16 * (size_t)(64 * NCT + 4) + 64 + (HS ? 64 + 16 + 64 * (size_t)NCT : 0));
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_panel_update_wide_f64<MODE, NCT, HS>), (int)max_bytes, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL((k_panel_update_wide_f64<MODE, NCT, HS>), dim3(len_pad / 16 + panel_fused_f64_extra_workgroups(&fx)), dim3(256), lds_bytes, stream,
	                   P, slabs, S, slab_stride, Q, RP, eps, ps, len_valid, sumsq_part, num_out, fx);
	return hipGetLastError();
}

// 16 panel rows per workgroup: len_pad / 16 norm partials (panel_update_parts, kernels.hip, knows)
hipError_t launch_panel_update_wide_f64(int mode, double* P, const double* slabs, int S, long slab_stride, const double* Q, int RP, int len_pad,
                                        double eps, double* ps, int len_valid, double* sumsq_part, double* num_out, hipStream_t stream, const PanelFusedF64* fused) {
	if (!panel_update_wide_f64_available(RP) || (mode != PANEL_MU && mode != PANEL_LS) || len_pad % 16 != 0) return hipErrorInvalidValue;
	PanelFusedF64 fx = {};
	if (fused != nullptr) fx = *fused;
	if (fx.h_side && mode != PANEL_MU) return hipErrorInvalidValue;
	if (!fx.smooth) { fx.off = 0.0; fx.diag = 1.0; }
#define NMFAMD_WIDE64(NCT)                                                                                                                       \
	return fx.h_side ? launch_wide_f64<PANEL_MU, NCT, true>(P, slabs, S, slab_stride, Q, RP, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream, fx) \
	     : mode == PANEL_MU ? launch_wide_f64<PANEL_MU, NCT, false>(P, slabs, S, slab_stride, Q, RP, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream, fx) \
	                        : launch_wide_f64<PANEL_LS, NCT, false>(P, slabs, S, slab_stride, Q, RP, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream, fx)
	switch (RP / 64) {      // 16-column tiles per wave
	case 2: NMFAMD_WIDE64(2);
	case 3: NMFAMD_WIDE64(3);
	case 4: NMFAMD_WIDE64(4);
	case 5: NMFAMD_WIDE64(5);
	case 6: NMFAMD_WIDE64(6);
	case 7: NMFAMD_WIDE64(7);
	default: NMFAMD_WIDE64(8);
	}
#undef NMFAMD_WIDE64
}

hipError_t launch_panel_update64_f64(int mode, double* P, const double* slabs, int S, long slab_stride, const double* Q, int len_pad,
                                     double eps, double* ps, int len_valid, double* sumsq_part, double* num_out, hipStream_t stream, const PanelFusedF64* fused) {
	if ((mode != PANEL_MU && mode != PANEL_LS) || len_pad % 32 != 0) return hipErrorInvalidValue;
	PanelFusedF64 fx = {};
	if (fused != nullptr) fx = *fused;
	if (fx.h_side && mode != PANEL_MU) return hipErrorInvalidValue;
	if (!fx.smooth) { fx.off = 0.0; fx.diag = 1.0; }
	dim3 grid(len_pad / 32 + panel_fused_f64_extra_workgroups(&fx)), block(256);
	if (fx.h_side) hipLaunchKernelGGL((k_panel_update64_f64<PANEL_MU, true>), grid, block, 0, stream, P, slabs, S, slab_stride, Q, eps, ps, len_valid, sumsq_part, num_out, fx);
	else if (mode == PANEL_MU) hipLaunchKernelGGL((k_panel_update64_f64<PANEL_MU, false>), grid, block, 0, stream, P, slabs, S, slab_stride, Q, eps, ps, len_valid, sumsq_part, num_out, fx);
	else hipLaunchKernelGGL((k_panel_update64_f64<PANEL_LS, false>), grid, block, 0, stream, P, slabs, S, slab_stride, Q, eps, ps, len_valid, sumsq_part, num_out, fx);
	return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// Gram matrix G = P P^T of a panel in double precision on the fp64 MFMA pipe (any padded rank: 64, k * 128)
// (reference: Dsyrk / Dgemm for W^T W and H H^T, AlgorithmMultiplicativeFrobenius.h:168-178,208-209)
// ------------------------------------------------------------------------------------------
// Workgroup = 4 waves = one 128 x 128 super-block (I <= J; 64 x 64 when RP = 64) over one slice of y; a wave owns a
// 64 x 64 (32 x 32) quarter as 4 x 4 (2 x 2) tiles of 16 x 16.  K-step = 4 panel rows; rows / columns of the tiles are
// interleaved (c = base + NT i + a), so a lane's operands are NT contiguous doubles.  Mirrored writes for I < J,
// slices summed in order by k_reduce_partials.
template <int NT, int D>      // NT = tiles per wave and direction: 4 (RP >= 128) or 2 (RP = 64)
__global__ __launch_bounds__(256, 2) void k_gram_f64(const double* __restrict__ P, int RP, int len, int parts, double* __restrict__ partial) {
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int l15 = lane & 15, kq = lane >> 4;
	constexpr int Q = 16 * NT;          // quarter edge: 64 or 32
	const int nb = RP / (2 * Q);
	int I = 0, rem = blockIdx.y;
	while (rem >= nb - I) { rem -= nb - I; ++I; }
	const int J = I + rem;
	const int ca = 2 * Q * I + Q * (wave >> 1), cb = 2 * Q * J + Q * (wave & 1);
	const int steps_total = (len + 3) / 4;
	const int s0 = (int)(((long)steps_total * blockIdx.x) / parts);
	const int s1 = (int)(((long)steps_total * (blockIdx.x + 1)) / parts);
	const int steps = s1 - s0;
	typedef double vecn __attribute__((ext_vector_type(NT)));

	f64x4 acc[NT][NT];
#pragma unroll
	for (int a = 0; a < NT; ++a)
#pragma unroll
		for (int b = 0; b < NT; ++b)
#pragma unroll
			for (int g = 0; g < 4; ++g) acc[a][b][g] = 0.0;

	if (steps > 0) {
		const double* pa = P + ((long)4 * s0 + kq) * RP + ca + NT * l15;
		const double* pb = P + ((long)4 * s0 + kq) * RP + cb + NT * l15;
		const long step = 4 * (long)RP;
		const int last = steps - 1;
		vecn va[D], vb[D];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int t = d < last ? d : last;
			va[d] = *reinterpret_cast<const vecn*>(pa + t * step);
			vb[d] = *reinterpret_cast<const vecn*>(pb + t * step);
		}
		__builtin_amdgcn_sched_barrier(0);
		int t = 0;
		for (; t + D <= steps; t += D) {
#pragma unroll
			for (int d = 0; d < D; ++d) {
#pragma unroll
				for (int a = 0; a < NT; ++a)
#pragma unroll
					for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[d][a], vb[d][b], acc[a][b], 0, 0, 0);
				int tn = t + D + d;
				tn = tn < last ? tn : last;
				va[d] = *reinterpret_cast<const vecn*>(pa + tn * step);
				vb[d] = *reinterpret_cast<const vecn*>(pb + tn * step);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		const int remn = steps - t;
#pragma unroll
		for (int d = 0; d < D; ++d) {
			if (d < remn) {
#pragma unroll
				for (int a = 0; a < NT; ++a)
#pragma unroll
					for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[d][a], vb[d][b], acc[a][b], 0, 0, 0);
			}
		}
	}
	// C/D map: register g of lane (j = l & 15, kq) is tile row i = kq + 4 g, tile column j
	double* out = partial + (long)blockIdx.x * RP * RP;
#pragma unroll
	for (int a = 0; a < NT; ++a)
#pragma unroll
		for (int b = 0; b < NT; ++b)
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				const int r = ca + NT * (kq + 4 * g) + a;
				const int c = cb + NT * l15 + b;
				// (off-diagonal super-blocks: the mirrored half is written ONCE, by the reduction -- k_gram_reduce_sym_f64; round 4 wrote it here, per slice, as
				//  8-byte stores 2 KB apart: a third of the kernel's time at the reference example's shape)
				out[(long)r * RP + c] = acc[a][b][g];
			}
}

// G = sum of the slices' partial matrices in k_reduce_partials' order (four groups of consecutive slices, eight loads in flight, groups added 0..3).  The slices hold
// the super-blocks (I, J), I <= J, of 128 x 128: an element of a block above the diagonal is also written to its mirrored place, an element below is left to its mirror.
// shift: log2 of the super-block edge the Gram kernel ran with (7: 128 x 128, padded ranks of 128; 6: 64 x 64, the other multiples of 64)
__global__ __launch_bounds__(256) void k_gram_reduce_sym_f64(const double* __restrict__ partial, int parts, int RP, double* __restrict__ G, int shift) {
	__shared__ double red[4][64];
	const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
	const long e = (long)blockIdx.x * 64 + tx;                 // (RP is a multiple of 64: a workgroup's 64 elements share a row and a super-block column)
	const int r = (int)(e / RP), c = (int)(e % RP);
	const int I = r >> shift, J = c >> shift;
	if (I > J) return;                                          // (whole workgroups: uniform)
	const long stride = (long)RP * RP;
	const int p0 = (parts * g) / 4, p1 = (parts * (g + 1)) / 4;
	double s = 0;
	for (int p = p0; p < p1; p += 8) {
		double v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) v[u] = partial[(long)(p + u < p1 ? p + u : p0) * stride + e];
#pragma unroll
		for (int u = 0; u < 8; ++u)
			if (p + u < p1) s += v[u];
	}
	red[g][tx] = s;
	__syncthreads();
	if (g == 0) {
		const double v = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
		G[e] = v;
		if (I < J) G[(long)c * RP + r] = v;
	}
}

// len: valid panel rows (rows behind them up to the padded length are zero); partial: parts * RP * RP elements of scratch
hipError_t launch_gram_f64(const double* P, int RP, int len, int parts, double* partial, double* G, hipStream_t stream) {
	if (RP % 64 != 0) return hipErrorInvalidValue;
	// super-blocks of 128 x 128 where the padded rank is a multiple of 128, of 64 x 64 elsewhere (64, 192, 320 ...: round 5, fp64 panels are padded to 64)
	const bool wide = RP % 128 == 0;
	const int nb = wide ? RP / 128 : RP / 64, nsuper = nb * (nb + 1) / 2;
	// at least 16 K-steps (64 panel rows) per slice -- except for short panels (the H side of the reference example: 165 columns at r = 158 were TWO slices, six
	// workgroups with 21 dependent K-steps each: 19.8 us for 21 MFLOP): there 4 K-steps per slice, so that the launch is one short round (round 5)
	parts = std::max(1, std::min(std::min(parts, std::max(16, 512 / nsuper)), std::max(1, len >= 1024 ? len / 64 : len / 16)));
	if (!wide) hipLaunchKernelGGL((k_gram_f64<2, 8>), dim3(parts, nsuper), dim3(256), 0, stream, P, RP, len, parts, partial);
	else hipLaunchKernelGGL((k_gram_f64<4, 6>), dim3(parts, nsuper), dim3(256), 0, stream, P, RP, len, parts, partial);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	if (RP == 64) return launch_reduce_partials<double>(partial, parts, (long)RP * RP, G, (long)RP * RP, stream);      // (one super-block: nothing to mirror)
	hipLaunchKernelGGL(k_gram_reduce_sym_f64, dim3((unsigned)((long)RP * RP / 64)), dim3(256), 0, stream, partial, parts, RP, G, wide ? 7 : 6);
	return hipGetLastError();
}

} // namespace nmfamd
