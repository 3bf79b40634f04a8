// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the NMF hot path and their launchers.
//
// Data layout in HBM:
//   * V   : column-major m x n, leading dimension padded to 128 rows, zero padding.
//   * Vt  : the transpose of V, column-major n x m, same padding.  V never changes during a
//           factorisation, so both orientations are kept resident; every product against V
//           then streams a matrix whose OUTPUT index is the contiguous one.
//   * factor panels (Wt = W^T, H, and every r x len intermediate such as W^T V or (V H^T)^T):
//           "panel layout" -- element (c, y) at P[y * RP + c], RP = padded rank, len padded to
//           128, zero padding.
//
// The two big products of every algorithm (reference: cublas gemm-TN `W^T V` and gemm-NT
// `V H^T`, source/common/Matrix.h:361-376 called from
// source/nmf/AlgorithmMultiplicativeFrobenius.h:187-188,240-241) are one kernel here:
//       OUT(c, x) = sum_y F(c, y) * A(x, y)          "factor product"
//   W^T V     : A = Vt (x = column of V, y = row of V), F = Wt   -> OUT in H's layout
//   (V H^T)^T : A = V  (x = row of V,    y = column of V), F = H -> OUT in Wt's layout
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>
#include <type_traits>

#include "kernels.h"
#include "tuning.h"
#include "inverse_gj64.h"

namespace nmfamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// factor product, fp32 MFMA
// ------------------------------------------------------------------------------------------
//
// One workgroup = 8 waves (two per SIMD, one workgroup per CU) = one 128-row x-tile times one
// slice of the y (reduction) range; the slice is cut again into 8 contiguous pieces, one per
// wave.  A wave keeps the whole 128 x (32*NB) output tile in 4*NB MFMA accumulators and
// streams its y piece two columns per v_mfma_f32_32x32x2_f32:
//
//   A operand (32 x 2): lane l holds A(x0 + 4*(l&31) + b, y + (l>>5)) for b = 0..3 -- ONE 16-byte
//       load per lane; component b feeds M-block b, i.e. MFMA row i of block b is matrix row
//       x0 + 4*i + b (the row order inside a tile is free).  A is stored x-TILED in HBM
//       (A(x, y) at A[(x/128)*tile_stride + y*128 + x%128]): the two columns of a K-step are 1 KiB
//       contiguous and a wave's whole y piece is one sequential stream -- with plain column-major
//       storage the same loads are 512-byte fragments 4*lda bytes apart and the main loop loses ~8 %
//       to DRAM row misses (in-kernel stamps: 1127 vs 1031 cycles per step pair).
//   B operand (2 x 32): lane l holds F(coff + NB*(l&31) + nb, y + (l>>5)) for nb = 0..NB-1 -- one
//       4*NB-byte load; MFMA column j of N-block nb is factor row coff + NB*j + nb.
//
// No LDS and no barrier in the main loop: operands go global -> VGPR through a D-deep register
// ring, the other wave on the SIMD covers what the ring does not.  The eight per-wave partial
// tiles are then summed through LDS in a fixed order (wave 0..7) and written as one fp32 slab
// per workgroup slice; slabs are summed in slice order by the consumer, so the result is
// deterministic (no atomics).
//
// Numerics: every accumulator is a k-ordered fmaf chain (the MFMA is exact fp32), see
// oracle_emulate_factor_product_f32 for the bit-exact restatement used by the tests.

constexpr int FP_WAVES = 8;
constexpr int FP_XT = 128;

// F operand of one K-step: NB column blocks of 32 per wave tile -- NB = 2 (64 panel columns, interleaved c = 2 j + nb, one
// 8-byte load) or NB = 1 (ranks <= 32: only the first 32 panel columns exist, c = j, one 4-byte load, half the MFMAs)
template <int NB> struct FVec;
template <> struct FVec<1> { typedef float type; };
template <> struct FVec<2> { typedef f32x2 type; };

template <int NB> __device__ inline float fcomp(const typename FVec<NB>::type& v, int i);
template <> __device__ inline float fcomp<1>(const float& v, int) { return v; }
template <> __device__ inline float fcomp<2>(const f32x2& v, int i) { return v[i]; }

// Extra workgroups of the factor-product launch (blockIdx.y == splits) reduce the partial Gram
// matrices the previous update kernel left behind -- work that has no dependence on the product
// itself and would otherwise cost its own launch on the critical path.  See kernels_mu64.hip.
//   Gu = sum_p partial[p]  (fixed order: two halves of p, then added)
//   normalize: scale(c) = Gu(c,c) > 0 ? 1/sqrt(Gu(c,c)) : 1, else scale = 1      (column norms of W)
//   G(a,b) = Gu(a,b) * scale(a) * scale(b)
// Every reduce workgroup derives the 64 scales itself, so no workgroup waits for another.
__device__ inline void gram_reduce_block(const GramReduceArgs& rg, int blk, float* lds) {
	const int tid = threadIdx.x;
	float* s_scale = lds;        // [64]
	float* s_tmp = lds + 64;     // [2][256]
	const int parts = rg.parts;
	if (rg.normalize) {
		if (tid < 128) {
			const int c = tid & 63, g = tid >> 6;
			const int p0 = (parts * g) / 2, p1 = (parts * (g + 1)) / 2;
			float sum = 0.f;
			int p = p0;
			for (; p + 8 <= p1; p += 8) {
				float v[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) v[u] = rg.partials[(long)(p + u) * 4096 + c * 65];
#pragma unroll
				for (int u = 0; u < 8; ++u) sum += v[u];
			}
			for (; p < p1; ++p) sum += rg.partials[(long)p * 4096 + c * 65];
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

template <typename T>
__global__ __launch_bounds__(512) void k_inverse_gj64(const T* __restrict__ A, int RP, int r, T* __restrict__ Ainv, T offdiag, T diag) {
	inverse_gj64_body<T>(A, RP, r, Ainv, offdiag, diag);
}

template <int XB, int D, bool STAMP, int DIAG = 0, int NB = 2>
__global__ __launch_bounds__(512, 2) void k_factor_product_f32(
	const float* __restrict__ A, long tile_stride,
	const float* __restrict__ F, int RP,
	float* __restrict__ slabs, long slab_stride,
	int steps_total, int splits, GramReduceArgs rg, unsigned long long* __restrict__ stamps) {
	const int coff = 64 * blockIdx.z;   // 64-column chunk of the panel (grid.z = RP / 64)
	// XB = 32-row M-blocks per wave tile: 4 (128-row x-tiles) or 5 (160-row x-tiles; the fifth block
	// is fed by an extra 4-byte load).  The plan picks the height that fills the 256 CUs best.
	// STAMP: diagnostic build only (nmfamd_tune_factor_product): per-wave shader-clock and 100 MHz
	// real-time stamps at kernel entry, first MFMA, end of the main loop and end of the epilogue,
	// written to a buffer of their own; the production instantiation has STAMP = false.
	constexpr int TH = 32 * XB;
	unsigned long long t_entry = 0, r_entry = 0, t_loop0 = 0, t_loop1 = 0;
	if (STAMP) { t_entry = __builtin_amdgcn_s_memtime(); r_entry = __builtin_amdgcn_s_memrealtime(); }
	typedef typename FVec<NB>::type fvec;
	extern __shared__ __attribute__((aligned(16))) float lds[];

	if (blockIdx.y == (unsigned)splits) {   // the passenger row of the grid (only launched when there is a passenger)
		// the 64 x 64 inverse of the least-squares algorithms rides as the FIRST block of the row: the workgroup
		// distributor hands blocks to the shader engines in turn and does not look for a free CU elsewhere, so the
		// block right behind the product's last one is the one that lands on a CU the product left idle
		if (rg.inv_a != nullptr) {
			if (blockIdx.x == 0) inverse_gj64_body<float>(rg.inv_a, 64, rg.inv_r, rg.inv_out, rg.inv_offdiag, rg.inv_diag);
		} else if (blockIdx.x < GRAM_REDUCE_BLOCKS) gram_reduce_block(rg, blockIdx.x, lds);
		return;
	}

	const int xt = blockIdx.x;
	const int sp = blockIdx.y;
	// wave-uniform values are forced into SGPRs so that the address arithmetic of the main loop
	// runs on the scalar unit
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int half = lane >> 5;
	const int l31 = lane & 31;

	// this wave's piece of the reduction range, in K-steps of two y each
	const int nw = splits * FP_WAVES;
	const int widx = sp * FP_WAVES + wave;
	const int s0 = (int)(((long)steps_total * widx) / nw);
	const int s1 = (int)(((long)steps_total * (widx + 1)) / nw);
	const int steps = s1 - s0;

	f32x16 acc[XB][NB];
#pragma unroll
	for (int b = 0; b < XB; ++b)
#pragma unroll
		for (int nb = 0; nb < NB; ++nb)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[b][nb][g] = 0.f;

	if (steps > 0) {
		// A is x-tiled: the TH rows of a tile are contiguous per column and the columns of a tile follow
		// each other, so a wave's whole y piece is ONE sequential stream (8 * TH bytes per K-step)
		const float* abase = A + (long)xt * tile_stride + (long)(2 * s0) * TH;   // uniform
		const float* fbase = F + (long)(2 * s0) * RP + coff;                      // uniform
		const unsigned aoff = (unsigned)(half * TH + 4 * l31);                    // per lane, elements
		const unsigned eoff = (unsigned)(half * TH + 128 + l31);                  // fifth block (XB == 5)
		const unsigned foff = (unsigned)(half * RP + NB * l31);
		const long astep = 2 * TH;
		const long fstep = 2 * (long)RP;
		const int last = steps - 1;

		f32x4 va[D];
		float ve[D];
		fvec fb[D];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int st = d < last ? d : last;
			va[d] = *reinterpret_cast<const f32x4*>(abase + st * astep + aoff);
			if (XB == 5) ve[d] = abase[st * astep + eoff]; else ve[d] = 0.f;
			fb[d] = *reinterpret_cast<const fvec*>(fbase + st * fstep + foff);
		}
		__builtin_amdgcn_sched_barrier(0);
		if (STAMP) { t_loop0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
		int t = 0;
		for (; t + D <= steps; t += D) {
#pragma unroll
			for (int d = 0; d < D; ++d) {
#pragma unroll
				for (int b = 0; b < XB; ++b)
#pragma unroll
					for (int nb = 0; nb < NB; ++nb)
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(b < 4 ? va[d][b & 3] : ve[d], fcomp<NB>(fb[d], nb), acc[b][nb], 0, 0, 0);
				int st = t + D + d;
				st = st < last ? st : last;
				// DIAG (stamped diagnostic builds only): 1 = no refill at all (pure MFMA issue rate),
				// 2 = refill A only, 3 = refill F only
				if (DIAG == 0 || DIAG == 2) {
					va[d] = *reinterpret_cast<const f32x4*>(abase + st * astep + aoff);
					if (XB == 5) ve[d] = abase[st * astep + eoff];
				}
				if (DIAG == 0 || DIAG == 3) fb[d] = *reinterpret_cast<const fvec*>(fbase + st * fstep + foff);
				// keep the operand that was NOT reloaded opaque to the optimiser (never the reloaded one:
				// an asm use would wait for the load right behind its issue)
				if (DIAG == 1 || DIAG == 3) { asm volatile("" : "+v"(va[d]), "+v"(ve[d])); }
				if (DIAG == 1 || DIAG == 2) { asm volatile("" : "+v"(fb[d])); }
				// pin the order: the refill of ring slot d is issued right behind the MFMAs that
				// consumed it, D-1 steps before its data is needed (hipcc otherwise sinks all loads
				// to the end of the unrolled body and the first step waits a full memory latency)
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		const int rem = steps - t;
#pragma unroll
		for (int d = 0; d < D; ++d) {
			if (d < rem) {
#pragma unroll
				for (int b = 0; b < XB; ++b)
#pragma unroll
					for (int nb = 0; nb < NB; ++nb)
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(b < 4 ? va[d][b & 3] : ve[d], fcomp<NB>(fb[d], nb), acc[b][nb], 0, 0, 0);
			}
		}
	}

	if (STAMP) { __builtin_amdgcn_sched_barrier(0); t_loop1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }

	// ---- sum the eight per-wave tiles through LDS, two M-blocks (four accumulator tiles) per round ----
	// LDS image of a round: [src wave 8][tile 4][q 4][lane 64] float4  (128 KiB)
	// C/D map of the 32x32 MFMA: register g of lane l is row (g&3) + 8*(g>>2) + 4*(l>>5), column l&31.
	// A round carries four accumulator tiles per wave: 2 M-blocks x 2 N-blocks (NB = 2) or 4 M-blocks (NB = 1).
	constexpr int MPR = 4 / NB;                        // M-blocks per round
	constexpr int ROUNDS = (XB + MPR - 1) / MPR;
	f32x4* l4 = reinterpret_cast<f32x4*>(lds);
	float* slab = slabs + (long)sp * slab_stride;
#pragma unroll
	for (int rd = 0; rd < ROUNDS; ++rd) {
		if (rd > 0) __syncthreads();
#pragma unroll
		for (int tl = 0; tl < 4; ++tl) {
			const int b = MPR * rd + tl / NB, nb = tl % NB;
			if (b < XB) {
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					f32x4 v;
					v[0] = acc[b < XB ? b : 0][nb][4 * q + 0]; v[1] = acc[b < XB ? b : 0][nb][4 * q + 1];
					v[2] = acc[b < XB ? b : 0][nb][4 * q + 2]; v[3] = acc[b < XB ? b : 0][nb][4 * q + 3];
					l4[((wave * 4 + tl) * 4 + q) * 64 + lane] = v;
				}
			}
		}
		__syncthreads();
		{
			// 16 (tile, register quad) slices per round, two per wave: wave w owns quad q = w & 3 of the tiles
			// NB * (w >> 2) + k, k = 0 .. NB-1 of M-block (w >> 2)   [NB = 2]
			// (w >> 2) and (w >> 2) + 2, i.e. M-blocks (w >> 2) and (w >> 2) + 2   [NB = 1]
			const int q = wave & 3;
			f32x4 sum[2];
			int bk[2];
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				const int tl = NB == 2 ? 2 * (wave >> 2) + k : (wave >> 2) + 2 * k;
				bk[k] = MPR * rd + tl / NB;
				f32x4 s = l4[((0 * 4 + tl) * 4 + q) * 64 + lane];
#pragma unroll
				for (int src = 1; src < FP_WAVES; ++src) s += l4[((src * 4 + tl) * 4 + q) * 64 + lane];
				sum[k] = s;
			}
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) {
				const int i = gi + 8 * q + 4 * half;   // MFMA row of this value
				if (NB == 2) {
					// both N-blocks of one M-block: columns 2 j and 2 j + 1 are one 8-byte store
					const int b = bk[0];
					if (b < XB) {
						const int x = xt * TH + (b < 4 ? 4 * i + b : 128 + i);
						f32x2 o; o[0] = sum[0][gi]; o[1] = sum[1][gi];
						*reinterpret_cast<f32x2*>(slab + (long)x * RP + coff + 2 * l31) = o;
					}
				} else {
#pragma unroll
					for (int k = 0; k < 2; ++k) {
						const int b = bk[k];
						if (b < XB) {
							const int x = xt * TH + (b < 4 ? 4 * i + b : 128 + i);
							slab[(long)x * RP + coff + l31] = sum[k][gi];
						}
					}
				}
			}
		}
	}
	if (STAMP) {
		__builtin_amdgcn_s_waitcnt(0);   // stores done
		const unsigned long long t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
		if (lane == 0) {
			unsigned long long* o = stamps + 8 * ((long)(blockIdx.y * gridDim.x + blockIdx.x) * FP_WAVES + wave);
			o[0] = t_entry; o[1] = t_loop0; o[2] = t_loop1; o[3] = t_end; o[4] = r_entry; o[5] = r_end; o[6] = steps;
			unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
			o[7] = xcc;
		}
	}
}

// X = valid output length.  x-tile height: 128 rows unless NMFAMD_FP_TILE=160 asks for the 160-row form.
// Measured at 10 000 x 5 000 (round 1): 160-row tiles fill 252 / 256 of the CUs instead of 237 / 240 and
// the product alone runs ~3-5 % faster, but the grid then leaves no CU for the passenger Gram
// reduction (it serialises behind the product: 72 vs 64 us in the iteration) and the extra split-K
// slab slows the update kernels; net 6 % slower per iteration, so 128 stays the default.
FactorProductPlan plan_factor_product(int X, int Y, int RP, int num_cus) {
	FactorProductPlan best;
	double best_score = -1.0;
	const int steps_total = (Y + 1) / 2;
	// keep at least 16 K-steps per wave, otherwise the prologue dominates
	int max_splits = steps_total / (16 * FP_WAVES);
	if (max_splits < 1) max_splits = 1;
	const char* force = getenv("NMFAMD_FP_TILE");
	const int forced = force ? atoi(force) : 128;
	for (int th = 128; th <= 160; th += 32) {
		if (forced != th) continue;
		FactorProductPlan p;
		p.th = th;
		p.xtiles = (X + th - 1) / th;
		p.steps_total = steps_total;
		int splits = num_cus / p.xtiles;
		if (splits < 1) splits = 1;
		if (splits > max_splits) splits = max_splits;
		p.splits = splits;
		p.nb = 2;                    // RP is a multiple of 64 (engine: padded_rank); the NB = 4 form needs 256 accumulator VGPRs
		p.chunks = RP / 64;
		const int wgs = p.xtiles * p.splits;
		const double fill = wgs >= num_cus ? 1.0 / ((wgs + num_cus - 1) / num_cus) * ((double)wgs / num_cus) : (double)wgs / num_cus;
		const double score = fill * ((double)X / ((double)p.xtiles * th)) * (th == 160 ? 0.985 : 1.0);   // 160: one more epilogue round
		if (score > best_score + 1e-9) { best_score = score; best = p; }
	}
	return best;
}

template <int XB, int D, bool STAMP, int DIAG = 0, int NB = 2>
static hipError_t launch_fp_d(const FactorProductPlan& p, const float* A, long tile_stride, const float* F, int RP,
                              float* slabs, long slab_stride, const GramReduceArgs* rg, unsigned long long* stamps, hipStream_t stream) {
	GramReduceArgs none = {nullptr, 0, nullptr, nullptr, 0};
	const bool wanted = rg != nullptr && (rg->partials != nullptr || rg->inv_a != nullptr);
	if (wanted && rg->partials != nullptr && rg->inv_a != nullptr) return hipErrorInvalidValue;      // one kind of passenger per launch
	const bool with_reduce = wanted && RP == 64 && p.xtiles >= GRAM_REDUCE_BLOCKS;
	if (wanted && !with_reduce) return hipErrorInvalidValue;
	dim3 grid(p.xtiles, p.splits + (with_reduce ? 1 : 0), p.chunks), block(512);      // passengers only ever with one chunk (RP == 64)
	const size_t lds_bytes = 8 * 4 * 4 * 64 * sizeof(f32x4);
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_factor_product_f32<XB, D, STAMP, DIAG, NB>), (int)lds_bytes, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL((k_factor_product_f32<XB, D, STAMP, DIAG, NB>), grid, block, lds_bytes, stream,
	                   A, tile_stride, F, RP, slabs, slab_stride, p.steps_total, p.splits, with_reduce ? *rg : none, stamps);
	return hipGetLastError();
}

// Register-ring depth of the main loop: 8 unless NMFAMD_FP_DEPTH selects another instantiation (tuning).
static int fp_depth() {
	static int d = -1;
	if (d < 0) {
		const char* e = tuning_env("NMFAMD_FP_DEPTH");
		d = e ? atoi(e) : 8;
		if (d != 4 && d != 6 && d != 8) d = 8;
	}
	return d;
}

template <int XB>
static hipError_t launch_fp(const FactorProductPlan& p, const float* A, long tile_stride, const float* F, int RP,
                            float* slabs, long slab_stride, const GramReduceArgs* rg, hipStream_t stream) {
	// ranks <= 32 (plan.nb == 1, padded rank 64): only the first 32 panel columns are computed and written; the other
	// 32 columns of every slab stay at the zeros they were allocated with
	if (p.nb == 1 && RP == 64 && XB == 4) return launch_fp_d<4, 8, false, 0, 1>(p, A, tile_stride, F, RP, slabs, slab_stride, rg, nullptr, stream);
#ifdef NMFAMD_DIAG_BUILD
	switch (fp_depth()) {
	case 4: return launch_fp_d<XB, 4, false>(p, A, tile_stride, F, RP, slabs, slab_stride, rg, nullptr, stream);
	case 6: return launch_fp_d<XB, 6, false>(p, A, tile_stride, F, RP, slabs, slab_stride, rg, nullptr, stream);
	default: break;
	}
#endif
	return launch_fp_d<XB, 8, false>(p, A, tile_stride, F, RP, slabs, slab_stride, rg, nullptr, stream);
}

hipError_t launch_factor_product_f32_stamped(const FactorProductPlan& p, const float* A, long tile_stride, const float* F, int RP,
                                             float* slabs, long slab_stride, unsigned long long* stamps, hipStream_t stream) {
#ifdef NMFAMD_DIAG_BUILD
	static int diag = -1;
	if (diag < 0) { const char* e = tuning_env("NMFAMD_FP_DIAG"); diag = e ? atoi(e) : 0; }
	if (p.th == 160) return launch_fp_d<5, 8, true, 0>(p, A, tile_stride, F, RP, slabs, slab_stride, nullptr, stamps, stream);
	switch (diag) {
	case 1: return launch_fp_d<4, 8, true, 1>(p, A, tile_stride, F, RP, slabs, slab_stride, nullptr, stamps, stream);
	case 2: return launch_fp_d<4, 8, true, 2>(p, A, tile_stride, F, RP, slabs, slab_stride, nullptr, stamps, stream);
	case 3: return launch_fp_d<4, 8, true, 3>(p, A, tile_stride, F, RP, slabs, slab_stride, nullptr, stamps, stream);
	default: return launch_fp_d<4, 8, true, 0>(p, A, tile_stride, F, RP, slabs, slab_stride, nullptr, stamps, stream);
	}
#else
	(void)p; (void)A; (void)tile_stride; (void)F; (void)RP; (void)slabs; (void)slab_stride; (void)stamps; (void)stream;
	return hipErrorNotSupported;      // stamped kernels exist in the diagnostic build only (tuning.h)
#endif
}

hipError_t launch_factor_product_f32(const FactorProductPlan& p, const float* A, long tile_stride, const float* F, int RP,
                                     float* slabs, long slab_stride, hipStream_t stream, const GramReduceArgs* rg) {
	if (p.th == 160) return launch_fp<5>(p, A, tile_stride, F, RP, slabs, slab_stride, rg, stream);
	return launch_fp<4>(p, A, tile_stride, F, RP, slabs, slab_stride, rg, stream);
}

// ------------------------------------------------------------------------------------------
// factor product, generic VALU form (fp64, and the fp32 cross-check in the tests)
// ------------------------------------------------------------------------------------------
// One workgroup = 64 x by 32 c outputs, the whole y range; A and F tiles staged through LDS.
// Thread (tx 0..15, tc 0..15) owns x = 4*tx..4*tx+3, c = 2*tc, 2*tc+1.
template <typename T>
__global__ __launch_bounds__(256) void k_factor_product_valu(
	const T* __restrict__ A, long lda, const T* __restrict__ F, int RP,
	T* __restrict__ out, int Y) {
	constexpr int YT = 16;
	__shared__ T sa[YT][64 + 1];
	__shared__ T sf[YT][32 + 1];
	const int x0 = blockIdx.x * 64, c0 = blockIdx.y * 32;
	const int tid = threadIdx.x, tx = tid & 15, tc = tid >> 4;
	T acc[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
	for (int y0 = 0; y0 < Y; y0 += YT) {
		for (int e = tid; e < YT * 64; e += 256) {
			int yy = e / 64, xx = e % 64;
			sa[yy][xx] = (y0 + yy < Y) ? A[(long)(y0 + yy) * lda + x0 + xx] : T(0);
		}
		for (int e = tid; e < YT * 32; e += 256) {
			int yy = e / 32, cc = e % 32;
			sf[yy][cc] = (y0 + yy < Y) ? F[(long)(y0 + yy) * RP + c0 + cc] : T(0);
		}
		__syncthreads();
#pragma unroll
		for (int yy = 0; yy < YT; ++yy) {
			T f0 = sf[yy][2 * tc], f1 = sf[yy][2 * tc + 1];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				T a = sa[yy][4 * tx + i];
				acc[i][0] += a * f0;
				acc[i][1] += a * f1;
			}
		}
		__syncthreads();
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		out[(long)(x0 + 4 * tx + i) * RP + c0 + 2 * tc] = acc[i][0];
		out[(long)(x0 + 4 * tx + i) * RP + c0 + 2 * tc + 1] = acc[i][1];
	}
}

template <typename T>
hipError_t launch_factor_product_valu(const T* A, long lda, int Xpad, int Y, const T* F, int RP, T* out, hipStream_t stream) {
	dim3 grid(Xpad / 64, RP / 32), block(256);
	hipLaunchKernelGGL((k_factor_product_valu<T>), grid, block, 0, stream, A, lda, F, RP, out, Y);
	return hipGetLastError();
}
template hipError_t launch_factor_product_valu<float>(const float*, long, int, int, const float*, int, float*, hipStream_t);
template hipError_t launch_factor_product_valu<double>(const double*, long, int, int, const double*, int, double*, hipStream_t);

// ------------------------------------------------------------------------------------------
// slab reduction: out = slab_0 + slab_1 + ... in slice order
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_reduce_slabs(const T* __restrict__ slabs, int S, long slab_stride, T* __restrict__ out, long count) {
	long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= count) return;
	T s = slabs[e];
	for (int k = 1; k < S; ++k) s += slabs[(long)k * slab_stride + e];
	out[e] = s;
}

// The same sum with four consecutive elements per thread (one 16- / 32-byte access per slab) and up to eight slabs requested together, added in slab order
// (round 4: the element-per-thread form above with one dependent load per slab is the 5 us launch between the product and the exchange of a column shard).
template <typename T>
__global__ __launch_bounds__(256) void k_reduce_slabs_v4(const T* __restrict__ slabs, int S, long slab_stride, T* __restrict__ out, long count4) {
	typedef T T4 __attribute__((ext_vector_type(4)));
	const long q = (long)blockIdx.x * 256 + threadIdx.x;
	if (q >= count4) return;
	const long e = 4 * q;
	T4 s = *reinterpret_cast<const T4*>(slabs + e);
	for (int k0 = 1; k0 < S; k0 += 7) {
		T4 t[7];
#pragma unroll
		for (int u = 0; u < 7; ++u) t[u] = *reinterpret_cast<const T4*>(slabs + (long)(k0 + u < S ? k0 + u : 0) * slab_stride + e);      // (clamped duplicates: this thread's own lines again)
#pragma unroll
		for (int u = 0; u < 7; ++u)
			if (k0 + u < S) s += t[u];
	}
	*reinterpret_cast<T4*>(out + e) = s;
}

template <typename T>
hipError_t launch_reduce_slabs(const T* slabs, int S, long slab_stride, T* out, long count, hipStream_t stream) {
	const int bs = 256;
	const bool aligned = count % 4 == 0 && slab_stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & (4 * sizeof(T) - 1)) == 0;
	if (aligned) {
		hipLaunchKernelGGL((k_reduce_slabs_v4<T>), dim3((unsigned)((count / 4 + bs - 1) / bs)), dim3(bs), 0, stream, slabs, S, slab_stride, out, count / 4);
		return hipGetLastError();
	}
	hipLaunchKernelGGL((k_reduce_slabs<T>), dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, stream, slabs, S, slab_stride, out, count);
	return hipGetLastError();
}
template hipError_t launch_reduce_slabs<float>(const float*, int, long, float*, long, hipStream_t);
template hipError_t launch_reduce_slabs<double>(const double*, int, long, double*, long, hipStream_t);

// ------------------------------------------------------------------------------------------
// Gram matrix of a factor panel: G(a, b) = sum_y P(a, y) P(b, y)
// (reference: syrk / gemm for W^T W and H H^T, AlgorithmMultiplicativeFrobenius.h:168-178,208-209,231-232)
// ------------------------------------------------------------------------------------------
// Stage 1: workgroup p sums its slice of y into partial[p] (RP x RP, column-major, both
// triangles, bitwise symmetric).  Stage 2 (k_reduce_slabs) adds the partials in order.
template <typename T>
__global__ __launch_bounds__(256) void k_gram_partial(const T* __restrict__ P, int RP, int len, int parts, T* __restrict__ partial) {
	constexpr int YT = 16;
	__shared__ T sa[YT][64 + 4];
	__shared__ T sb[YT][64 + 4];
	const int part = blockIdx.x;
	const int a0 = blockIdx.y * 64, b0 = blockIdx.z * 64;
	const int y_begin = (int)(((long)len * part) / parts);
	const int y_end = (int)(((long)len * (part + 1)) / parts);
	const int tid = threadIdx.x, ta = tid & 15, tb = tid >> 4;
	T acc[4][4];
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = 0;
	for (int y0 = y_begin; y0 < y_end; y0 += YT) {
		for (int e = tid; e < YT * 64; e += 256) {
			int yy = e >> 6, cc = e & 63;
			bool ok = y0 + yy < y_end;
			sa[yy][cc] = ok ? P[(long)(y0 + yy) * RP + a0 + cc] : T(0);
			sb[yy][cc] = ok ? P[(long)(y0 + yy) * RP + b0 + cc] : T(0);
		}
		__syncthreads();
#pragma unroll
		for (int yy = 0; yy < YT; ++yy) {
			T av[4], bv[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) { av[i] = sa[yy][4 * ta + i]; bv[i] = sb[yy][4 * tb + i]; }
#pragma unroll
			for (int i = 0; i < 4; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
		}
		__syncthreads();
	}
	T* out = partial + (long)part * RP * RP;
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) out[(long)(b0 + 4 * tb + j) * RP + a0 + 4 * ta + i] = acc[i][j];
}

template <typename T>
hipError_t launch_gram(const T* P, int RP, int len, int parts, T* partial, T* G, hipStream_t stream) {
	if constexpr (std::is_same<T, float>::value) {
		if (RP == 64) return launch_gram64_f32(P, len, parts, partial, G, stream);
		if (gram_wide_available(RP) && std::getenv("NMFAMD_FORCE_VALU") == nullptr) return launch_gram_wide_f32(P, RP, len, parts, partial, G, stream);
	}
	if constexpr (std::is_same<T, double>::value) {
		if (RP % 64 == 0 && std::getenv("NMFAMD_FORCE_VALU") == nullptr) return launch_gram_f64(P, RP, len, parts, partial, G, stream);
	}
	int blocks = RP / 64; // RP is a multiple of 64
	parts = std::max(1, std::min(parts, std::max(1, len / 64)));      // short panels: fewer, longer slices (less partial traffic)
	hipLaunchKernelGGL((k_gram_partial<T>), dim3(parts, blocks, blocks), dim3(256), 0, stream, P, RP, len, parts, partial);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	return launch_reduce_partials<T>(partial, parts, (long)RP * RP, G, (long)RP * RP, stream);
}
template hipError_t launch_gram<float>(const float*, int, int, int, float*, float*, hipStream_t);
template hipError_t launch_gram<double>(const double*, int, int, int, double*, double*, hipStream_t);

// ------------------------------------------------------------------------------------------
// panel update: the element-wise step of every algorithm, fused with the slab reduction, the
// small r x r product and the partial sums that follow it in the reference.
// ------------------------------------------------------------------------------------------
//   num(c, y)  = sum_s slab_s(c, y)
//   MODE_MU:  den = Q(c, :) . P(:, y);   P(c, y) <- P(c, y) * num / (den + eps)
//             (reference: symm/gemm `RR * H` / `W * RR` + kernel::multiplyDivide,
//              AlgorithmMultiplicativeFrobenius.h:181-191,235-244, KernelMultiplyDivide.cu:39-42)
//   MODE_LS:  P(c, y) <- max(0, Q(c, :) . num(:, y))        Q = inverse of the normal matrix
//             (reference: ormqr + trsm + setNegativeToZero, AlgorithmAlternatingLeastSquares.h:163-172)
//   MODE_SET: P(c, y) <- num(c, y)                          plain reduction into a panel
// Optional outputs:
//   ps(y)            = sum_c P_new(c, y) * num(c, y)   -- diag(H^T (W^T V)), the per-column terms of
//                      tr(H^T W^T V) (kernel::traceMultiplication, KernelTraceMultiplication.cu:43-80)
//   sumsq_part(wg,c) = sum_{y in workgroup} P_new(c, y)^2  -- first half of kernel::normalizeColumns
enum { MODE_MU = 0, MODE_LS = 1, MODE_SET = 2 };

template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_panel_update(
	T* __restrict__ P, const T* __restrict__ slabs, int S, long slab_stride,
	const T* __restrict__ Q, int RP, T eps,
	T* __restrict__ ps, int len_valid, T* __restrict__ sumsq_part, T* __restrict__ num_out, int YB) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	T* s_num = reinterpret_cast<T*>(smem_raw);       // [YB][RP]
	T* s_old = s_num + YB * RP;                      // [YB][RP]  (old panel values, or new * num products)
	const int y0 = blockIdx.x * YB;
	const int tid = threadIdx.x;
	const long base = (long)y0 * RP;

	for (int e = tid; e < YB * RP; e += 256) {
		T s = slabs[base + e];
		for (int k = 1; k < S; ++k) s += slabs[(long)k * slab_stride + base + e];
		s_num[e] = s;
		if (num_out) num_out[base + e] = s;
		if (MODE == MODE_MU) s_old[e] = P[base + e];
	}
	__syncthreads();

	const int c = tid % RP;
	const int ystride = 256 / RP > 0 ? 256 / RP : 1;
	// RP <= 256: one thread per (c, y); RP > 256 is handled by the c-loop below
	for (int cc = c; cc < RP; cc += 256) {
		for (int yy = tid / RP; yy < YB; yy += ystride) {
			T result;
			if (MODE == MODE_SET) {
				result = s_num[yy * RP + cc];
			} else {
				const T* vec = (MODE == MODE_MU) ? (s_old + yy * RP) : (s_num + yy * RP);
				T dot = 0;
				for (int k = 0; k < RP; ++k) dot += Q[(long)k * RP + cc] * vec[k];
				if (MODE == MODE_MU) {
					T value = s_old[yy * RP + cc];
					result = value * s_num[yy * RP + cc] / (dot + eps);
				} else {
					result = dot > T(0) ? dot : T(0);
				}
			}
			P[base + (long)yy * RP + cc] = result;
		}
	}
	if (ps == nullptr && sumsq_part == nullptr) return;
	__syncthreads();
	// s_old <- new values (every thread re-reads what the workgroup just wrote)
	for (int e = tid; e < YB * RP; e += 256) s_old[e] = P[base + e];
	__syncthreads();
	if (ps != nullptr && tid < YB) {
		T s = 0;
		for (int k = 0; k < RP; ++k) s += s_old[tid * RP + k] * s_num[tid * RP + k];
		if (y0 + tid < len_valid) ps[y0 + tid] = s;
	}
	if (sumsq_part != nullptr) {
		for (int cc = tid; cc < RP; cc += 256) {
			T s = 0;
			for (int yy = 0; yy < YB; ++yy) { T v = s_old[yy * RP + cc]; s += v * v; }
			sumsq_part[(long)blockIdx.x * RP + cc] = s;
		}
	}
}

// panel columns per workgroup: 32, fewer when two [rows][RP] images would not fit 64 KiB of LDS
int panel_update_rows(int RP, size_t elem) {
	int yb = 32;
	while (yb > 1 && 2 * (size_t)yb * RP * elem > 65536) yb >>= 1;
	return yb;
}

static bool use_wide_update(int RP) { return panel_update_wide_available(RP) && std::getenv("NMFAMD_FORCE_VALU") == nullptr; }

int panel_update_parts(int RP, size_t elem, int len_pad) {
	if (elem == 4 && RP == 64) return tuning_env("NMFAMD_UPDATE64_OLD") ? len_pad / 128 : len_pad / 64;   // k_panel_update64_lds_f32 (old: k_panel_update64_f32)
	if (elem == 4 && use_wide_update(RP)) return len_pad / 32;   // k_panel_update_wide_f32
	if (elem == 8 && panel_update_wide_f64_available(RP) && std::getenv("NMFAMD_FORCE_VALU") == nullptr) return len_pad / 16;   // k_panel_update_wide_f64
	return len_pad / panel_update_rows(RP, elem);
}

bool panel_update_delivers_gram(int RP, size_t elem) {
	return elem == 4 && RP == 64 && tuning_env("NMFAMD_UPDATE64_OLD") == nullptr && std::getenv("NMFAMD_FORCE_VALU") == nullptr;
}

template <typename T>
hipError_t launch_panel_update(int mode, T* P, const T* slabs, int S, long slab_stride, const T* Q, int RP, int len_pad,
                               T eps, T* ps, int len_valid, T* sumsq_part, T* num_out, hipStream_t stream, T* gram_partial, void* x3_out, int x3_ks, void* q_split,
                               const PanelTriExtras* tri) {
	if ((gram_partial != nullptr || x3_out != nullptr) && !(panel_update_delivers_gram(RP, sizeof(T)) && mode != MODE_SET)) return hipErrorInvalidValue;
	if (tri != nullptr && !(std::is_same<T, float>::value && RP == 256 && mode == MODE_MU && use_wide_update(RP))) return hipErrorInvalidValue;
	if constexpr (std::is_same<T, float>::value) {
		if (RP == 64 && mode != MODE_SET) {
			if (tuning_env("NMFAMD_UPDATE64_OLD")) return launch_panel_update64_f32(mode, P, slabs, S, slab_stride, Q, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream);
			return launch_panel_update64_lds_f32(mode, P, slabs, S, slab_stride, Q, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream, gram_partial, x3_out, x3_ks);
		}
		if (use_wide_update(RP) && mode != MODE_SET)
			return launch_panel_update_wide_f32(mode, P, slabs, S, slab_stride, Q, RP, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream, q_split, tri);
	}
	if constexpr (std::is_same<T, double>::value) {
		// same 32 rows per workgroup as the generic kernel at this rank: the norm-partial count does not change
		if (RP == 64 && mode != MODE_SET && std::getenv("NMFAMD_FORCE_VALU") == nullptr)
			return launch_panel_update64_f64(mode, P, slabs, S, slab_stride, Q, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream);
		if (panel_update_wide_f64_available(RP) && mode != MODE_SET && std::getenv("NMFAMD_FORCE_VALU") == nullptr)
			return launch_panel_update_wide_f64(mode, P, slabs, S, slab_stride, Q, RP, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream);
	}
	const int yb = panel_update_rows(RP, sizeof(T));
	dim3 grid(len_pad / yb), block(256);
	size_t smem = 2 * (size_t)yb * RP * sizeof(T);
	switch (mode) {
	case MODE_MU: hipLaunchKernelGGL((k_panel_update<T, MODE_MU>), grid, block, smem, stream, P, slabs, S, slab_stride, Q, RP, eps, ps, len_valid, sumsq_part, num_out, yb); break;
	case MODE_LS: hipLaunchKernelGGL((k_panel_update<T, MODE_LS>), grid, block, smem, stream, P, slabs, S, slab_stride, Q, RP, eps, ps, len_valid, sumsq_part, num_out, yb); break;
	default: hipLaunchKernelGGL((k_panel_update<T, MODE_SET>), grid, block, smem, stream, P, slabs, S, slab_stride, Q, RP, eps, ps, len_valid, sumsq_part, num_out, yb); break;
	}
	return hipGetLastError();
}
template hipError_t launch_panel_update<float>(int, float*, const float*, int, long, const float*, int, int, float, float*, int, float*, float*, hipStream_t, float*, void*, int, void*, const PanelTriExtras*);
template hipError_t launch_panel_update<double>(int, double*, const double*, int, long, const double*, int, int, double, double*, int, double*, double*, hipStream_t, double*, void*, int, void*, const PanelTriExtras*);

// ------------------------------------------------------------------------------------------
// column normalisation of W (rows of the Wt panel): second half of kernel::normalizeColumns
// (KernelNormalizeColumns.cu:37-58): sum > 0 ? x / sqrt(sum) : x
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_normalize_panel(T* __restrict__ P, int RP, const T* __restrict__ sumsq_part, int parts) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	T* s_norm = reinterpret_cast<T*>(smem_raw);
	for (int c = threadIdx.x; c < RP; c += 256) {
		T s = 0;
		for (int p = 0; p < parts; ++p) s += sumsq_part[(long)p * RP + c];
		s_norm[c] = s > T(0) ? (T)sqrt(s) : T(0);
	}
	__syncthreads();
	const long base = (long)blockIdx.x * 32 * RP;
	for (int e = threadIdx.x; e < 32 * RP; e += 256) {
		T nrm = s_norm[e % RP];
		if (nrm > T(0)) P[base + e] = P[base + e] / nrm;
	}
}

template <typename T>
hipError_t launch_normalize_panel(T* P, int RP, int len_pad, T* sumsq_part, int parts, hipStream_t stream) {
	return launch_normalize_panel_v2<T>(P, RP, len_pad, sumsq_part, parts, stream);
}
template hipError_t launch_normalize_panel<float>(float*, int, int, float*, int, hipStream_t);
template hipError_t launch_normalize_panel<double>(double*, int, int, double*, int, hipStream_t);

// ------------------------------------------------------------------------------------------
// nsNMF smoothing of a panel: out(:, y) = S P(:, y), S = (1-theta) I + (theta/r) 1 1^T
// (reference: gemm with the explicit S matrix, AlgorithmNonSmoothNMF.h:131-134,175,194;
//  here S is applied analytically: out(c) = offdiag * sum_c' P(c') + (diag - offdiag) * P(c))
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_smooth_panel(const T* __restrict__ P, T* __restrict__ out, int RP, int r, long len_pad, T offdiag, T diag) {
	// one wave per panel column y: lanes stride the factor rows (coalesced), butterfly sum
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const long y = (long)blockIdx.x * 4 + wave;
	if (y >= len_pad) return;
	const T* p = P + y * RP;
	T* o = out + y * RP;
	T sum = 0;
	for (int c = lane; c < r; c += 64) sum += p[c];
	for (int w = 32; w > 0; w >>= 1) sum += __shfl_xor(sum, w);
	for (int c = lane; c < RP; c += 64) o[c] = c < r ? offdiag * (sum - p[c]) + diag * p[c] : T(0);
}

template <typename T>
hipError_t launch_smooth_panel(const T* P, T* out, int RP, int r, long len_pad, T offdiag, T diag, hipStream_t stream) {
	hipLaunchKernelGGL((k_smooth_panel<T>), dim3((unsigned)((len_pad + 3) / 4)), dim3(256), 0, stream, P, out, RP, r, len_pad, offdiag, diag);
	return hipGetLastError();
}
template hipError_t launch_smooth_panel<float>(const float*, float*, int, int, long, float, float, hipStream_t);
template hipError_t launch_smooth_panel<double>(const double*, double*, int, int, long, double, double, hipStream_t);

// ------------------------------------------------------------------------------------------
// small r x r helpers (single workgroup)
// ------------------------------------------------------------------------------------------
// ps(d) = sum_i A(d, i) B(i, d): the r terms of tr(H H^T W^T W)
// (kernel::traceMultiplication<false>, KernelTraceMultiplication.cu:43-80 at AlgorithmMultiplicativeFrobenius.h:212)
// b_colsq (optional, r sums of squares): B stands for diag(f) B diag(f), f(c) = b_colsq[c] > 0 ? 1 / sqrt(b_colsq[c]) : 1 -- the Gram matrix of a panel whose
// column normalisation is still pending (rank-256 bf16 path, PanelTriExtras)
// b_scale (optional, r factors): the same with the factors given directly, f(c) = b_scale[c] (the fused double-precision iteration's pending column scale)
template <typename T>
__global__ __launch_bounds__(256) void k_trace_small(const T* __restrict__ A, const T* __restrict__ B, int RP, int r, T* __restrict__ ps, const T* __restrict__ b_colsq,
                                                     const T* __restrict__ b_scale) {
	// one wave per diagonal element, lanes stride the inner index, butterfly sum (fixed order)
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int d = blockIdx.x * 4 + wave;
	if (d >= r) return;
	T s = 0;
	if (b_scale != nullptr) {
		const T fd = b_scale[d];
		for (int i = lane; i < r; i += 64) s += A[(long)i * RP + d] * (B[(long)d * RP + i] * (fd * b_scale[i]));
	} else if (b_colsq != nullptr) {
		const T fd = b_colsq[d] > T(0) ? T(1) / (T)sqrt(b_colsq[d]) : T(1);
		for (int i = lane; i < r; i += 64) {
			const T fi = b_colsq[i] > T(0) ? T(1) / (T)sqrt(b_colsq[i]) : T(1);
			s += A[(long)i * RP + d] * ((B[(long)d * RP + i] * fd) * fi);
		}
	} else
	for (int i = lane; i < r; i += 64) s += A[(long)i * RP + d] * B[(long)d * RP + i];
	for (int w = 32; w > 0; w >>= 1) s += __shfl_xor(s, w);
	if (lane == 0) ps[d] = s;
}

template <typename T>
hipError_t launch_trace_small(const T* A, const T* B, int RP, int r, T* ps, hipStream_t stream, const T* b_colsq, const T* b_scale) {
	hipLaunchKernelGGL((k_trace_small<T>), dim3((r + 3) / 4), dim3(256), 0, stream, A, B, RP, r, ps, b_colsq, b_scale);
	return hipGetLastError();
}
template hipError_t launch_trace_small<float>(const float*, const float*, int, int, float*, hipStream_t, const float*, const float*);
template hipError_t launch_trace_small<double>(const double*, const double*, int, int, double*, hipStream_t, const double*, const double*);

// ps(d) = sum_y A(d, y) B(d, y) over two panels: the r terms of tr(W_old^T (V H^T)) used by the
// least-squares family (AlgorithmAlternatingLeastSquares.h:199-205) and GDCLS (:259-264).
// One workgroup per d-block would be overkill: r <= 256 rows, len <= ~1e5 -> one workgroup per row.
template <typename T>
__global__ __launch_bounds__(256) void k_row_dot(const T* __restrict__ A, const T* __restrict__ B, int RP, long len, T* __restrict__ ps) {
	__shared__ T red[256];
	const int d = blockIdx.x;
	T s = 0;
	for (long y = threadIdx.x; y < len; y += 256) s += A[y * RP + d] * B[y * RP + d];
	red[threadIdx.x] = s;
	__syncthreads();
	for (int w = 128; w > 0; w >>= 1) {
		if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
		__syncthreads();
	}
	if (threadIdx.x == 0) ps[d] = red[0];
}

// The same sums with coalesced reads (round 4: the kernel above walks a panel column with one 4-byte load per 256-byte row -- every workgroup touches every cache
// line of both panels: 25 us at config 5 on every error iteration).  Workgroup g takes a contiguous range of panel rows, a thread four consecutive d (16-byte
// loads) of every (1024 / RP)-th row of the range; per-workgroup partial vectors, summed in order by launch_reduce_partials.
template <typename T>
__global__ __launch_bounds__(256) void k_row_dot_part(const T* __restrict__ A, const T* __restrict__ B, int RP, long len, int groups, T* __restrict__ part) {
	typedef T T4 __attribute__((ext_vector_type(4)));
	__shared__ T s_p[256 * 4];
	const int per_row = RP / 4, c4 = threadIdx.x % per_row, yy = threadIdx.x / per_row, ystep = 256 / per_row;
	const long y0 = (len * blockIdx.x) / groups, y1 = (len * (blockIdx.x + 1)) / groups;
	T4 acc = {0, 0, 0, 0};
	for (long y = y0 + yy; y < y1; y += 4l * ystep) {
		T4 a[4], b[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const long yu = y + (long)u * ystep;
			const long e = (yu < y1 ? yu : y0) * RP + 4 * c4;
			a[u] = *reinterpret_cast<const T4*>(A + e); b[u] = *reinterpret_cast<const T4*>(B + e);
		}
#pragma unroll
		for (int u = 0; u < 4; ++u)
			if (y + (long)u * ystep < y1) acc += a[u] * b[u];
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) s_p[threadIdx.x * 4 + i] = acc[i];
	__syncthreads();
	if ((int)threadIdx.x < RP) {
		const int c = threadIdx.x;
		T v = 0;
		for (int g = 0; g < ystep; ++g) v += s_p[(g * per_row + c / 4) * 4 + (c & 3)];
		part[(long)blockIdx.x * RP + c] = v;
	}
}

// part (optional): ROW_DOT_GROUPS * RP elements of scratch -- with it the coalesced two-launch form runs (RP <= 256)
template <typename T>
hipError_t launch_row_dot(const T* A, const T* B, int RP, int r, long len, T* ps, hipStream_t stream, T* part) {
	if (part != nullptr && RP <= 256 && RP % 64 == 0 && len >= 4096) {
		hipLaunchKernelGGL((k_row_dot_part<T>), dim3(ROW_DOT_GROUPS), dim3(256), 0, stream, A, B, RP, len, ROW_DOT_GROUPS, part);
		if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
		return launch_reduce_partials<T>(part, ROW_DOT_GROUPS, RP, ps, r, stream);
	}
	hipLaunchKernelGGL((k_row_dot<T>), dim3(r), dim3(256), 0, stream, A, B, RP, len, ps);
	return hipGetLastError();
}
template hipError_t launch_row_dot<float>(const float*, const float*, int, int, long, float*, hipStream_t, float*);
template hipError_t launch_row_dot<double>(const double*, const double*, int, int, long, double*, hipStream_t, double*);

// A(i, j) = [reuse] A(i, j) + (i == j ? diag : offdiag) on the r x r block
// (kernel::fillMatrix / addConstantToMatrix, KernelFillMatrix.cu:29-45)
template <typename T>
__global__ void k_fill_small(T* __restrict__ A, int RP, int r, int reuse, T offdiag, T diag) {
	for (int e = threadIdx.x; e < r * r; e += blockDim.x) {
		int i = e % r, j = e / r;
		T old = reuse ? A[(long)j * RP + i] : T(0);
		A[(long)j * RP + i] = old + (i == j ? diag : offdiag);
	}
}

template <typename T>
hipError_t launch_fill_small(T* A, int RP, int r, int reuse, T offdiag, T diag, hipStream_t stream) {
	hipLaunchKernelGGL((k_fill_small<T>), dim3(1), dim3(256), 0, stream, A, RP, r, reuse, offdiag, diag);
	return hipGetLastError();
}
template hipError_t launch_fill_small<float>(float*, int, int, int, float, float, hipStream_t);
template hipError_t launch_fill_small<double>(double*, int, int, int, double, double, hipStream_t);

// dst[i] = src[i], i < count: the error terms of an iteration into their pinned host buffer (dst: device-visible host memory).  A kernel, not hipMemcpyAsync: the
// runtime's device-to-host copy costs the stream 12 us of idle time in front of its 1.4 us blit and 6 us behind it (rocprofv3 trace of config 2's error iterations).
template <typename T>
__global__ __launch_bounds__(256) void k_copy_small(T* __restrict__ dst, const T* __restrict__ src, long count) {
	for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) dst[i] = src[i];
}

template <typename T>
hipError_t launch_copy_small(T* dst, const T* src, long count, hipStream_t stream) {
	if (count <= 0) return hipSuccess;
	hipLaunchKernelGGL((k_copy_small<T>), dim3((unsigned)std::min<long>(64, (count + 255) / 256)), dim3(256), 0, stream, dst, src, count);
	return hipGetLastError();
}
// dst = [a (na elements) | b (nb elements)]
template <typename T>
__global__ __launch_bounds__(256) void k_copy_two(T* __restrict__ dst, const T* __restrict__ a, long na, const T* __restrict__ b, long nb) {
	for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (long)gridDim.x * 256) dst[i] = i < na ? a[i] : b[i - na];
}

template <typename T>
hipError_t launch_copy_two(T* dst, const T* a, long na, const T* b, long nb, hipStream_t stream) {
	if (na + nb <= 0) return hipSuccess;
	hipLaunchKernelGGL((k_copy_two<T>), dim3((unsigned)std::min<long>(64, (na + nb + 255) / 256)), dim3(256), 0, stream, dst, a, na, b, nb);
	return hipGetLastError();
}
template hipError_t launch_copy_two<float>(float*, const float*, long, const float*, long, hipStream_t);
template hipError_t launch_copy_two<double>(double*, const double*, long, const double*, long, hipStream_t);
template hipError_t launch_copy_small<float>(float*, const float*, long, hipStream_t);
template hipError_t launch_copy_small<double>(double*, const double*, long, hipStream_t);

// Inverse of the r x r normal matrix by Householder QR, one workgroup, arithmetic in double.
// (reference: cusolverDn geqrf, then ormqr + trsm per right-hand side, Matrix.h:565-618; here
//  the factorisation is turned into an explicit inverse once, X = R^-1 Q^T, so that applying it
//  to the r x n / m x r right-hand sides is a plain r x r product inside k_panel_update.)
// work: 2 * r * r doubles of global scratch (QR image, and Q^T accumulated from the identity).
template <typename T>
__global__ __launch_bounds__(256) void k_inverse_small(const T* __restrict__ A, int RP, int r, T* __restrict__ Ainv, double* __restrict__ work) {
	__shared__ double red[256];
	__shared__ double s_tau, s_beta;
	double* M = work;              // r x r column-major
	double* X = work + (long)r * r; // r x r column-major, starts as I, ends as R^-1 Q^T
	const int tid = threadIdx.x;
	for (int e = tid; e < r * r; e += 256) {
		int i = e % r, j = e / r;
		M[e] = (double)A[(long)j * RP + i];
		X[e] = i == j ? 1.0 : 0.0;
	}
	__syncthreads();
	for (int k = 0; k < r; ++k) {
		// norm of the sub-column
		double s = 0;
		for (int i = k + 1 + tid; i < r; i += 256) { double v = M[(long)k * r + i]; s += v * v; }
		red[tid] = s;
		__syncthreads();
		for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
		if (tid == 0) {
			double alpha = M[(long)k * r + k], xnorm2 = red[0];
			if (xnorm2 == 0.0) { s_tau = 0.0; s_beta = alpha; }
			else {
				double beta = sqrt(alpha * alpha + xnorm2);
				if (alpha >= 0) beta = -beta;
				s_tau = (beta - alpha) / beta;
				s_beta = beta;
				M[(long)k * r + k] = 1.0 / (alpha - beta); // scale, consumed below
			}
		}
		__syncthreads();
		const double tau = s_tau;
		if (tau != 0.0) {
			const double scale = M[(long)k * r + k];
			__syncthreads();
			for (int i = k + 1 + tid; i < r; i += 256) M[(long)k * r + i] *= scale;
			if (tid == 0) M[(long)k * r + k] = s_beta;
			__syncthreads();
			// apply H_k to the trailing columns of M and to all columns of X: one thread per column
			for (int j = tid; j < (r - k - 1) + r; j += 256) {
				double* col = j < r - k - 1 ? (M + (long)(k + 1 + j) * r) : (X + (long)(j - (r - k - 1)) * r);
				double w = col[k];
				for (int i = k + 1; i < r; ++i) w += M[(long)k * r + i] * col[i];
				w *= tau;
				col[k] -= w;
				for (int i = k + 1; i < r; ++i) col[i] -= w * M[(long)k * r + i];
			}
		}
		__syncthreads();
	}
	// back substitution R Z = X (X currently holds Q^T), one thread per column
	for (int j = tid; j < r; j += 256) {
		double* x = X + (long)j * r;
		for (int k = r - 1; k >= 0; --k) {
			double s = x[k];
			for (int p = k + 1; p < r; ++p) s -= M[(long)p * r + k] * x[p];
			x[k] = s / M[(long)k * r + k];
		}
	}
	__syncthreads();
	for (int e = tid; e < RP * RP; e += 256) {
		int i = e % RP, j = e / RP;
		Ainv[e] = (i < r && j < r) ? (T)X[(long)j * r + i] : T(0);
	}
}

template <typename T>
hipError_t launch_inverse_small(T* A, int RP, int r, T* Ainv, double* work, T offdiag, T diag, hipStream_t stream) {
	if (r <= 64) {
		hipLaunchKernelGGL((k_inverse_gj64<T>), dim3(1), dim3(512), 0, stream, A, RP, r, Ainv, offdiag, diag);
		return hipGetLastError();
	}
	if (offdiag != T(0) || diag != T(0)) {
		hipError_t e = launch_fill_small<T>(A, RP, r, 1, offdiag, diag, stream);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL((k_inverse_small<T>), dim3(1), dim3(256), 0, stream, A, RP, r, Ainv, work);
	return hipGetLastError();
}
template hipError_t launch_inverse_small<float>(float*, int, int, float*, double*, float, float, hipStream_t);
template hipError_t launch_inverse_small<double>(double*, int, int, double*, double*, double, double, hipStream_t);

// ------------------------------------------------------------------------------------------
// one-time data movement kernels (outside the iteration loop)
// ------------------------------------------------------------------------------------------
// dst(j, i) = src(i, j); src is rows x cols with leading dimension lds_, dst has leading dimension ldd.
template <typename T>
__global__ __launch_bounds__(256) void k_transpose(const T* __restrict__ src, long lds_, int rows, int cols, T* __restrict__ dst, long ldd, unsigned gx) {
	__shared__ T tile[32][33];
	// one-dimensional grid (grid.y stops at 65 535: a 2.1 M-row panel would not launch): block b = (b % gx, b / gx)
	const int i0 = (int)(blockIdx.x % gx) * 32, j0 = (int)(blockIdx.x / gx) * 32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	for (int jj = ty; jj < 32; jj += 8) {
		int i = i0 + tx, j = j0 + jj;
		tile[jj][tx] = (i < rows && j < cols) ? src[(long)j * lds_ + i] : T(0);
	}
	__syncthreads();
	for (int ii = ty; ii < 32; ii += 8) {
		int i = i0 + ii, j = j0 + tx;
		if (i < rows && j < cols) dst[(long)i * ldd + j] = tile[tx][ii];
	}
}

template <typename T>
hipError_t launch_transpose(const T* src, long lds_, int rows, int cols, T* dst, long ldd, hipStream_t stream) {
	const unsigned gx = (unsigned)((rows + 31) / 32), gy = (unsigned)((cols + 31) / 32);
	if ((unsigned long long)gx * gy > 0x7fffffffull) return hipErrorInvalidValue;
	hipLaunchKernelGGL((k_transpose<T>), dim3(gx * gy), dim3(256), 0, stream, src, lds_, rows, cols, dst, ldd, gx);
	return hipGetLastError();
}
template hipError_t launch_transpose<float>(const float*, long, int, int, float*, long, hipStream_t);
template hipError_t launch_transpose<double>(const double*, long, int, int, double*, long, hipStream_t);

// ps(j) = sum_i V(i, j)^2 -- the per-column terms of tr(V^T V)
// (kernel::traceMultiplication<true>(V, V), AlgorithmMultiplicativeFrobenius.h:119-121)
template <typename T>
__global__ __launch_bounds__(256) void k_column_sumsq(const T* __restrict__ V, long ldv, int rows, T* __restrict__ ps, int* __restrict__ range_flag) {
	__shared__ T red[256];
	const T* col = V + (long)blockIdx.x * ldv;
	T s = 0;
	bool odd = false;      // a value the exact three-way bf16 split of kernels_x3.hip does not cover (see launch_column_sumsq)
	for (int i = threadIdx.x; i < rows; i += 256) {
		const T v = col[i];
		s += v * v;
		if (sizeof(T) == 4) { const float a = fabsf((float)v); odd |= !(a <= 0x1p126f) || (a != 0.f && a < 0x1p-100f); }
	}
	if (range_flag != nullptr && __any(odd) && (threadIdx.x & 63) == 0) atomicOr(range_flag, 1);
	red[threadIdx.x] = s;
	__syncthreads();
	for (int w = 128; w > 0; w >>= 1) {
		if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
		__syncthreads();
	}
	if (threadIdx.x == 0) ps[blockIdx.x] = red[0];
}

template <typename T>
hipError_t launch_column_sumsq(const T* V, long ldv, int rows, int cols, T* ps, hipStream_t stream, int* range_flag) {
	hipLaunchKernelGGL((k_column_sumsq<T>), dim3(cols), dim3(256), 0, stream, V, ldv, rows, ps, range_flag);
	return hipGetLastError();
}
template hipError_t launch_column_sumsq<float>(const float*, long, int, int, float*, hipStream_t, int*);
template hipError_t launch_column_sumsq<double>(const double*, long, int, int, double*, hipStream_t, int*);

// Sparse -> dense on the device, honouring the index base
// (reference: cusparse csr2dense / csc2dense / coo2csr+csr2dense, Matrix.h:145-232).
// The destination has been zero-filled.  One thread per stored element.
template <typename T>
__global__ void k_densify(int format, const T* __restrict__ values, const int* __restrict__ ptr, const int* __restrict__ idx,
                          const int* __restrict__ idx2, long nnz, int outer, int base, T* __restrict__ V, long ldv, int rows, int cols) {
	long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= nnz) return;
	int i, j;
	if (format == 3) { // COO: idx = row indices, idx2 = column indices
		i = idx[p] - base; j = idx2[p] - base;
	} else {
		// find the row (CSR) / column (CSC) that owns element p: last o with ptr[o] - base <= p
		int lo = 0, hi = outer; // ptr has outer + 1 entries
		while (hi - lo > 1) { int mid = (lo + hi) >> 1; if ((long)(ptr[mid] - base) <= p) lo = mid; else hi = mid; }
		if (format == 1) { i = lo; j = idx[p] - base; } else { j = lo; i = idx[p] - base; }
	}
	// Duplicate coordinates ADD (as on the sparse-compute path, Engine::upload_triplets): a plain store would leave whichever duplicate's
	// thread came last -- a race.  The destination is zero-filled, so a unique coordinate is stored exactly (0 + v); two duplicates add
	// exactly in either order; with three or more the rounding can depend on the order of the additions.  (cusparse's csr2dense,
	// Matrix.h:161-168, documents nothing for duplicates.)
	if (i >= 0 && i < rows && j >= 0 && j < cols) atomicAdd(&V[(long)j * ldv + i], values[p]);
}

template <typename T>
hipError_t launch_densify(int format, const T* values, const int* ptr, const int* idx, const int* idx2, long nnz, int outer, int base,
                          T* V, long ldv, int rows, int cols, hipStream_t stream) {
	if (nnz == 0) return hipSuccess;
	hipLaunchKernelGGL((k_densify<T>), dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, stream, format, values, ptr, idx, idx2, nnz, outer, base, V, ldv, rows, cols);
	return hipGetLastError();
}
template hipError_t launch_densify<float>(int, const float*, const int*, const int*, const int*, long, int, int, float*, long, int, int, hipStream_t);
template hipError_t launch_densify<double>(int, const double*, const int*, const int*, const int*, long, int, int, double*, long, int, int, hipStream_t);

// Uniform (0, 1] fill for AllRandomValues (reference: curandGenerateUniform with the XORWOW
// default generator over ld * cols elements, source/init/RandomValueStrategy.cpp:29-49; cuRAND is
// closed source, so only the distribution -- uniform on (0,1], same seed for W and H -- is
// reproduced).  Counter-based generator: splitmix64 of (seed, element index).
template <typename T>
__global__ void k_fill_uniform(T* __restrict__ P, int RP, int r, long len, long len_pad, uint64_t seed, long y_first) {
	long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= len_pad * RP) return;
	long y = e / RP; int c = (int)(e % RP);
	T v = T(0);
	if (y < len && c < r) {
		uint64_t z = seed * 0x9E3779B97F4A7C15ull + (uint64_t)((y + y_first) * (long)r + c) + 0x632BE59BD9B4E019ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z = z ^ (z >> 31);
		if (sizeof(T) == 4) v = (T)(((z >> 40) + 1) * (1.0f / 16777216.0f));          // 24 bits -> (0, 1]
		else v = (T)(((z >> 11) + 1) * (1.0 / 9007199254740992.0));                    // 53 bits -> (0, 1]
	}
	P[e] = v;
}

template <typename T>
hipError_t launch_fill_uniform(T* P, int RP, int r, long len, long len_pad, uint64_t seed, hipStream_t stream, long y_first) {
	long count = len_pad * RP;
	hipLaunchKernelGGL((k_fill_uniform<T>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, P, RP, r, len, len_pad, seed, y_first);
	return hipGetLastError();
}
template hipError_t launch_fill_uniform<float>(float*, int, int, long, long, uint64_t, hipStream_t, long);
template hipError_t launch_fill_uniform<double>(double*, int, int, long, long, uint64_t, hipStream_t, long);

// ------------------------------------------------------------------------------------------
// x-tiled storage of the streamed matrix (see k_factor_product_f32): A(x, y) at
// A[(x / th) * tile_stride + y * th + x % th], th = tile height (128 or 160)
// ------------------------------------------------------------------------------------------
// dst(x, y) = src(x, y): column-major src (ld) -> tiled dst.  dst has been zero-filled.
template <typename T>
__global__ __launch_bounds__(256) void k_tile(const T* __restrict__ src, long ld, int X, int Y, T* __restrict__ dst, long tile_stride, int th, int untile) {
	const long x = (long)blockIdx.y * 256 + threadIdx.x;
	const long y = blockIdx.x;
	if (x >= X || y >= Y) return;
	const long t = (x / th) * tile_stride + y * th + (x % th);
	if (untile) dst[y * ld + x] = src[t];   // (roles swapped: src tiled, dst column-major)
	else dst[t] = src[y * ld + x];
}

// dst(j, i) = src(i, j): column-major src (rows = I, cols = J, ld) -> tiled dst of the transpose.
template <typename T>
__global__ __launch_bounds__(256) void k_tile_transposed(const T* __restrict__ src, long ld, int I, int J, T* __restrict__ dst, long tile_stride, int th, unsigned gx) {
	__shared__ T tile[32][33];
	const int i0 = (int)(blockIdx.x % gx) * 32, j0 = (int)(blockIdx.x / gx) * 32;      // one-dimensional grid, see k_transpose
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	for (int jj = ty; jj < 32; jj += 8) {
		const int i = i0 + tx, j = j0 + jj;
		tile[jj][tx] = (i < I && j < J) ? src[(long)j * ld + i] : T(0);
	}
	__syncthreads();
	for (int ii = ty; ii < 32; ii += 8) {
		const int i = i0 + ii, j = j0 + tx;
		if (i < I && j < J) dst[(long)(j / th) * tile_stride + (long)i * th + (j % th)] = tile[tx][ii];
	}
}

template <typename T>
hipError_t launch_tile(const T* src, long ld, int X, int Y, T* dst, long tile_stride, int th, bool untile, hipStream_t stream) {
	hipLaunchKernelGGL((k_tile<T>), dim3(Y, (X + 255) / 256), dim3(256), 0, stream, src, ld, X, Y, dst, tile_stride, th, untile ? 1 : 0);
	return hipGetLastError();
}
template hipError_t launch_tile<float>(const float*, long, int, int, float*, long, int, bool, hipStream_t);
template hipError_t launch_tile<double>(const double*, long, int, int, double*, long, int, bool, hipStream_t);

template <typename T>
hipError_t launch_tile_transposed(const T* src, long ld, int I, int J, T* dst, long tile_stride, int th, hipStream_t stream) {
	const unsigned gx = (unsigned)((I + 31) / 32), gy = (unsigned)((J + 31) / 32);
	if ((unsigned long long)gx * gy > 0x7fffffffull) return hipErrorInvalidValue;
	hipLaunchKernelGGL((k_tile_transposed<T>), dim3(gx * gy), dim3(256), 0, stream, src, ld, I, J, dst, tile_stride, th, gx);
	return hipGetLastError();
}
template hipError_t launch_tile_transposed<float>(const float*, long, int, int, float*, long, int, hipStream_t);
template hipError_t launch_tile_transposed<double>(const double*, long, int, int, double*, long, int, hipStream_t);

} // namespace nmfamd
