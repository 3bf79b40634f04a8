// kernels_wide.hip -- fp32 MFMA forms of the r x r x (m + n) sized work at padded ranks 128 ... 512.
//
// At rank 256 (BASELINE config 4) the products `RR * H` / `W * RR` and the Gram matrices are
// 2 * r^2 * (m + n) FLOP each -- as much as a product against V costs in bf16 -- so they cannot stay
// on the generic VALU kernels of kernels.hip.  Same arithmetic as k_panel_update / k_gram_partial
// (reference: symm/gemm + kernel::multiplyDivide, AlgorithmMultiplicativeFrobenius.h:181-191,235-244;
// syrk, :168-178,208-209) up to fp32 summation order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.h"
#include "tuning.h"
#include "split3.h"
#include "gram_wide.h"

namespace nmfamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WIDE_YB = 32;       // panel rows (y) per workgroup
constexpr int WIDE_MAX_RP = 512;

// ------------------------------------------------------------------------------------------
// panel update, wide panels: slab reduction + r x r product on the MFMA pipe + element-wise update
// + error / norm partial sums.  Semantics of k_panel_update (kernels.hip), MODE_MU and MODE_LS.
// ------------------------------------------------------------------------------------------
// Workgroup = 4 waves = 32 panel rows y.  The reduced numerator and the old panel values sit in LDS
// as [32][RP + 4] images (coalesced global traffic both ways).  D(c, y) = sum_k Q(k, c) vec(y, k):
//   wave w owns the column blocks cb = w, w + 4, ... (32 columns c each), all 32 y;
//   B operand: lane (y = l & 31, h = l >> 5) reads vec(y, 8u + 4h .. + 3) as one ds_read_b128 and feeds
//              the four values to four MFMAs -- the K order is k = 8u + 4h + gi;
//   A operand: lane (c = l & 31, h) holds Q(8u + 4h + gi, 32 cb + c): 128 B per half-wave from L2.
// The MFMA C/D map gives lane (y, h) the rows c = 32 cb + 8q + 4h + gi -- four consecutive c per q, so
// the element-wise step reads old / num and writes new as b128 LDS accesses.
// FX (round 6, PanelFusedF32, kernels.h; the fp32 counterpart of kernels_f64.hip's HS path): W is carried unnormalised with a pending column scale d and nsNMF's S
// is applied around the r x r product -- num <- S D (sum of the slabs), u = D S old as the B operand leaves LDS, den = S D (Q u) with the row's sum through LDS --
// and the update writes what its consumers read: the smoothed panel S H and the SPLIT IMAGE of the next product's operand (k_pack_panel_x3's layout), so that
// no pack, smoothing or normalisation launch is left in the iteration.  Any padded rank of the kernel (NCB = 1 .. 4); Q as its split image only.
template <int MODE, int NCB, bool FX = false>      // NCB = column blocks per wave = RP / 128
__global__ __launch_bounds__(256, 2) void k_panel_update_wide_f32(
	float* __restrict__ P, const float* __restrict__ slabs, int S, long slab_stride,
	const float* __restrict__ Q, int RP, float eps, float* __restrict__ ps, int len_valid,
	float* __restrict__ sumsq_part, float* __restrict__ num_out, const bf16x8* __restrict__ Qx3, const PanelTriExtras tri, const PanelFusedF32 fx) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	const int LD = RP + 4;
	float* s_num = lds;                       // [32][LD]
	float* s_old = lds + WIDE_YB * LD;        // [32][LD]   old values, then the new ones
	float* s_ps = s_old + WIDE_YB * LD;       // [4][32] error terms, [4][32] row sums of the new rows
	// FX: the RP factors (zero from r on when smoothing), [4][32] partial sums of D (Q u), the 32 row sums of the old rows, the 32 row sums of the new rows
	float* s_fd = s_ps + 256;                 // [RP]
	float* s_fsig = s_fd + RP;                // [4][32]
	float* s_frs = s_fsig + 128;              // [32]
	float* s_fns = s_frs + 32;                // [32]
	const bool fxh = FX && fx.h_side != 0;
	// (uniforms that meet vector values go through a VGPR: no packed fp32 instruction may read a scalar register, split3.h)
	const float f_off = FX ? in_vgpr(fx.off) : 0.f, f_diag = FX ? in_vgpr(fx.diag) : 1.f, f_ha = FX ? in_vgpr(fx.diag - fx.off) : 1.f;
	// PanelTriExtras::den_transform (rank 256; the launcher sizes the LDS for it): the product's operand u = D S old, the factors D, partial sums of D (G u)
	const bool dent = NCB == 2 && tri.den_transform;
	float* s_u = s_ps + 256;                  // [32][LD]
	float* s_dg = s_u + WIDE_YB * LD;         // [RP]
	float* s_tau = s_dg + RP;                 // [4][32]
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const long base = (long)blockIdx.x * WIDE_YB * RP;
	const int q4 = RP / 4;                    // float4 per panel row
	if (dent) s_dg[tid] = tri.den_colsq != nullptr ? tri_pending_scale(tri.den_colsq, tri.den_colsq_parts, 256, tid) : 1.0f;      // (256 threads, 256 columns)
	if (fxh) {
		const int r_eff = fx.smooth ? fx.r : RP;
		for (int c = tid; c < RP; c += 256) s_fd[c] = c < r_eff ? (fx.scale != nullptr ? fx.scale[c] : 1.0f) : 0.0f;
	}

	// 1. numerator = sum of the split-K slabs (slab order), old panel values.  All of a thread's loads of one
	//    slab are issued together (NE independent 16-byte loads): one memory latency per slab, not per element.
	constexpr int NE = WIDE_YB * 32 * NCB / 256;      // float4 per thread
	{
		f32x4 num[NE];
#pragma unroll
		for (int i = 0; i < NE; ++i) num[i] = *reinterpret_cast<const f32x4*>(slabs + base + 4l * (tid + 256 * i));
		if (MODE == PANEL_MU) {
			f32x4 old[NE];
#pragma unroll
			for (int i = 0; i < NE; ++i) old[i] = *reinterpret_cast<const f32x4*>(P + base + 4l * (tid + 256 * i));
			f32x4 od = {1.f, 1.f, 1.f, 1.f};
			if (NCB == 2 && tri.old_colsq != nullptr) {      // the panel's pending column scale (rank 256 only: this thread's columns are 4 (tid & 63) .. in every row)
#pragma unroll
				for (int j = 0; j < 4; ++j) od[j] = tri_pending_scale(tri.old_colsq, tri.old_colsq_parts, 256, 4 * (tid & 63) + j);
			}
#pragma unroll
			for (int i = 0; i < NE; ++i) {
				const int e = tid + 256 * i, y = e / q4, c4 = e - y * q4;
				if (NCB == 2 && tri.old_colsq != nullptr) old[i] *= od;
				if (FX && fx.old_scale != nullptr) old[i] *= *reinterpret_cast<const f32x4*>(fx.old_scale + 4 * c4);      // (the panel's own pending column scale)
				*reinterpret_cast<f32x4*>(s_old + y * LD + 4 * c4) = old[i];
			}
			if (dent) {
				// u(y, c) = d(c) (den_a old(y, c) + den_b sum_c' old(y, c')) for c < r: a panel row is the 64 float4 of one wave (as for the numerator below)
				const int c0 = 4 * (tid & 63);
				f32x4 dg = {1.f, 1.f, 1.f, 1.f};
				if (tri.den_colsq != nullptr) {
#pragma unroll
					for (int j = 0; j < 4; ++j) dg[j] = tri_pending_scale(tri.den_colsq, tri.den_colsq_parts, 256, c0 + j);
				}
				float rs[NE];
#pragma unroll
				for (int i = 0; i < NE; ++i) rs[i] = (old[i][0] + old[i][1]) + (old[i][2] + old[i][3]);      // (columns >= r hold zeros)
#pragma unroll
				for (int w = 32; w > 0; w >>= 1)
#pragma unroll
					for (int i = 0; i < NE; ++i) rs[i] += __shfl_xor(rs[i], w);
#pragma unroll
				for (int i = 0; i < NE; ++i) {
					const int y = (tid + 256 * i) / q4;
					f32x4 u;
#pragma unroll
					for (int j = 0; j < 4; ++j) u[j] = c0 + j < tri.r ? dg[j] * (tri.den_a * old[i][j] + tri.den_b * rs[i]) : 0.f;
					*reinterpret_cast<f32x4*>(s_u + y * LD + c0) = u;
				}
			}
		}
		// (three slabs in flight, added in slab order: config 4's H update sums nine of them -- nine latencies in a row otherwise)
		int k = 1;
		for (; k + 3 <= S; k += 3) {
			f32x4 t[3][NE];
#pragma unroll
			for (int j = 0; j < 3; ++j)
#pragma unroll
				for (int i = 0; i < NE; ++i) t[j][i] = *reinterpret_cast<const f32x4*>(slabs + (long)(k + j) * slab_stride + base + 4l * (tid + 256 * i));
#pragma unroll
			for (int j = 0; j < 3; ++j)
#pragma unroll
				for (int i = 0; i < NE; ++i) num[i] += t[j][i];
		}
		for (; k < S; ++k) {
			f32x4 t[NE];
#pragma unroll
			for (int i = 0; i < NE; ++i) t[i] = *reinterpret_cast<const f32x4*>(slabs + (long)k * slab_stride + base + 4l * (tid + 256 * i));
#pragma unroll
			for (int i = 0; i < NE; ++i) num[i] += t[i];
		}
		if (NCB == 2 && tri.num_transform) {
			// scale and nsNMF smoothing on the output side of the product (PanelTriExtras): RP = 256 (the launcher checks), so a panel row is the 64
			// float4 of one wave (e = tid + 256 i: row tid / 64 + 4 i, columns 4 (tid & 63) ..) and its sum is a butterfly over the wave
			const int c0 = 4 * (tid & 63);
			f32x4 d = {1.f, 1.f, 1.f, 1.f};
			if (tri.num_colsq != nullptr) {
#pragma unroll
				for (int j = 0; j < 4; ++j) d[j] = tri_pending_scale(tri.num_colsq, tri.num_colsq_parts, 256, c0 + j);
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) d[j] = c0 + j < tri.r ? d[j] : 0.f;
			float rs[NE];
#pragma unroll
			for (int i = 0; i < NE; ++i) { num[i] *= d; rs[i] = (num[i][0] + num[i][1]) + (num[i][2] + num[i][3]); }
#pragma unroll
			for (int w = 32; w > 0; w >>= 1)
#pragma unroll
				for (int i = 0; i < NE; ++i) rs[i] += __shfl_xor(rs[i], w);
#pragma unroll
			for (int i = 0; i < NE; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j) num[i][j] = c0 + j < tri.r ? tri.num_a * num[i][j] + tri.num_b * rs[i] : 0.f;
		}
#pragma unroll
		for (int i = 0; i < NE; ++i) {
			const int e = tid + 256 * i, y = e / q4, c4 = e - y * q4;
			*reinterpret_cast<f32x4*>(s_num + y * LD + 4 * c4) = num[i];
			if (num_out) *reinterpret_cast<f32x4*>(num_out + base + 4l * e) = num[i];
		}
	}
	__syncthreads();
	if (fxh) {
		// num(y, c) <- d(c) num(y, c), then S on the first r entries of the row (k_smooth_panel's formula); the old row's sum for the B operand.  Eight threads per row.
		const int y = tid >> 3, sub = tid & 7;
		float* row = s_num + y * LD;
		float sum = 0.f, osum = 0.f;
		for (int c = sub; c < RP; c += 8) {
			const float x = row[c] * s_fd[c];      // (zero from r_eff on)
			row[c] = x;
			sum += x;
			if (s_fd[c] != 0.f) osum += s_old[y * LD + c];
		}
		if (fx.smooth) {
#pragma unroll
			for (int w = 1; w < 8; w <<= 1) { sum += __shfl_xor(sum, w); osum += __shfl_xor(osum, w); }
			for (int c = sub; c < fx.r; c += 8) { const float x = row[c]; row[c] = f_off * (sum - x) + f_diag * x; }
			if (sub == 0) s_frs[y] = osum;
		} else if (sub == 0) s_frs[y] = 0.f;
		__syncthreads();
	}

	// 2. the r x r product
	const float* vec = (dent ? s_u : MODE == PANEL_MU ? s_old : s_num) + l31 * LD + 4 * half;
	f32x16 acc[NCB];
#pragma unroll
	for (int i = 0; i < NCB; ++i)
#pragma unroll
		for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
	if (Qx3 != nullptr) {
		// The same product at fp32 accuracy on the bf16 matrix pipe (kernels_x3.hip): Q comes pre-split into three bf16
		// planes in fragment order (k_pack_panel_x3 of Q: A(c, k) = Q(k, c)), the LDS operand is split in registers;
		// six 32x32x16 MFMAs per 16 k replace eight 32x32x2 fp32 ones at a quarter of their cycles each.
		const int NBT = RP / 32, ksteps = RP / 16;
		const bf16x8* qf = Qx3 + (long)wave * 192 + lane;        // + (ks * NBT + 4 i) * 192 + plane * 64
		const float* vb = (dent ? s_u : MODE == PANEL_MU ? s_old : s_num) + l31 * LD + 8 * half;
		constexpr int DX = NCB <= 2 ? 4 : 2;
		bf16x8 af[DX][NCB][3];
#pragma unroll
		for (int d = 0; d < DX; ++d)
#pragma unroll
			for (int i = 0; i < NCB; ++i)
#pragma unroll
				for (int pl = 0; pl < 3; ++pl) af[d][i][pl] = qf[((long)d * NBT + 4 * i) * 192 + pl * 64];
		__builtin_amdgcn_sched_barrier(0);
		for (int u = 0; u < ksteps; u += DX) {
#pragma unroll
			for (int d = 0; d < DX; ++d) {
				float v[8];
				const f32x4 b0 = *reinterpret_cast<const f32x4*>(vb + 16 * (u + d));
				const f32x4 b1 = *reinterpret_cast<const f32x4*>(vb + 16 * (u + d) + 4);
#pragma unroll
				for (int j = 0; j < 4; ++j) { v[j] = b0[j]; v[4 + j] = b1[j]; }
				if (fxh) {
					// u = D S old, formed as the value leaves LDS: u(k) = d(k) ((diag - off) old(k) + off rowsum), k = 16 (u + d) + 8 half + j
					const float h_a = f_ha, h_b = f_off * s_frs[l31];
					const f32x4 d0 = *reinterpret_cast<const f32x4*>(s_fd + 16 * (u + d) + 8 * half), d1 = *reinterpret_cast<const f32x4*>(s_fd + 16 * (u + d) + 8 * half + 4);
#pragma unroll
					for (int j = 0; j < 4; ++j) { v[j] = d0[j] * (h_a * v[j] + h_b); v[4 + j] = d1[j] * (h_a * v[4 + j] + h_b); }
				}
				bf16x8 hi, mid, lo;
				const bool b16 = NCB == 2 && tri.old_as_bf16;      // (PanelTriExtras: the panel's rows enter the product rounded to bf16; uniform)
				if (b16) {
#pragma unroll
					for (int j = 0; j < 8; ++j) hi[j] = (__bf16)v[j];
					mid = hi; lo = hi;
				} else split3(v, hi, mid, lo);
#pragma unroll
				for (int i = 0; i < NCB; ++i) {
					acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[d][i][2], hi, acc[i], 0, 0, 0);
					if (!b16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[d][i][0], lo, acc[i], 0, 0, 0);
					if (!b16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[d][i][1], mid, acc[i], 0, 0, 0);
					acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[d][i][1], hi, acc[i], 0, 0, 0);
					if (!b16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[d][i][0], mid, acc[i], 0, 0, 0);
					acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[d][i][0], hi, acc[i], 0, 0, 0);
				}
				int nu = u + DX + d;
				nu = nu < ksteps ? nu : ksteps - 1;       // tail: harmless re-load of the last K-step
#pragma unroll
				for (int i = 0; i < NCB; ++i)
#pragma unroll
					for (int pl = 0; pl < 3; ++pl) af[d][i][pl] = qf[((long)nu * NBT + 4 * i) * 192 + pl * 64];
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	} else {
	const float* qp = Q + (long)(4 * half) * RP + 32 * wave + l31;
	const int groups = RP / 8;                // multiple of 16
	// The A operands come from L2 (Q is RP x RP, shared by every workgroup): a D-deep register ring keeps
	// D groups of loads in flight so that the L2 latency is paid once, not once per group.
	constexpr int D = NCB <= 2 ? 8 : 4;
	float a[D][NCB][4];
	f32x4 b[D];
#pragma unroll
	for (int d = 0; d < D; ++d) {
		b[d] = *reinterpret_cast<const f32x4*>(vec + 8 * d);
#pragma unroll
		for (int i = 0; i < NCB; ++i)
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) a[d][i][gi] = qp[(long)(8 * d + gi) * RP + 128 * i];
	}
	__builtin_amdgcn_sched_barrier(0);
	for (int u = 0; u < groups; u += D) {
#pragma unroll
		for (int d = 0; d < D; ++d) {
#pragma unroll
			for (int gi = 0; gi < 4; ++gi)
#pragma unroll
				for (int i = 0; i < NCB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][i][gi], b[d][gi], acc[i], 0, 0, 0);
			int nu = u + D + d;
			nu = nu < groups ? nu : groups - 1;       // tail: harmless re-load of the last group
			b[d] = *reinterpret_cast<const f32x4*>(vec + 8 * nu);
#pragma unroll
			for (int i = 0; i < NCB; ++i)
#pragma unroll
				for (int gi = 0; gi < 4; ++gi) a[d][i][gi] = qp[(long)(8 * nu + gi) * RP + 128 * i];
			__builtin_amdgcn_sched_barrier(0);
		}
	}

	}

	// den = S D acc: acc(c, y) <- den_a d(c) acc(c, y) + den_b sum_c' d(c') acc(c', y); the sum runs over this lane's 16 NCB columns, its half-wave
	// partner's and the other three waves'
	if (dent) {
		float tl = 0.f;
#pragma unroll
		for (int i = 0; i < NCB; ++i)
#pragma unroll
			for (int g = 0; g < 16; ++g) {
				const int c = 32 * (wave + 4 * i) + 8 * (g >> 2) + 4 * half + (g & 3);
				acc[i][g] *= s_dg[c];
				tl += acc[i][g];
			}
		tl += __shfl_xor(tl, 32);
		if (half == 0) s_tau[wave * 32 + l31] = tl;
		__syncthreads();
		const float tau = ((s_tau[l31] + s_tau[32 + l31]) + s_tau[64 + l31]) + s_tau[96 + l31];
#pragma unroll
		for (int i = 0; i < NCB; ++i)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[i][g] = tri.den_a * acc[i][g] + tri.den_b * tau;
	}

	if (fxh) {
		// den = S D t: v(c) = d(c) t(c); den(c) = off (sigma - v(c)) + diag v(c), sigma = the row's sum of v over all column blocks and waves
		float tl = 0.f;
#pragma unroll
		for (int i = 0; i < NCB; ++i)
#pragma unroll
			for (int g = 0; g < 16; ++g) {
				const int c = 32 * (wave + 4 * i) + 8 * (g >> 2) + 4 * half + (g & 3);
				acc[i][g] *= s_fd[c];
				tl += acc[i][g];
			}
		if (fx.smooth) {
			tl += __shfl_xor(tl, 32);
			if (half == 0) s_fsig[wave * 32 + l31] = tl;
			__syncthreads();
			const float sigma = ((s_fsig[l31] + s_fsig[32 + l31]) + s_fsig[64 + l31]) + s_fsig[96 + l31];
#pragma unroll
			for (int i = 0; i < NCB; ++i)
#pragma unroll
				for (int g = 0; g < 16; ++g) acc[i][g] = f_off * (sigma - acc[i][g]) + f_diag * acc[i][g];
		}
	}

	// 3. element-wise step in the C/D layout; new values replace the old ones in LDS once every wave
	//    has finished reading them as B operands
	f32x4 nv[NCB][4];
	float psum = 0.f, rsum = 0.f;                 // rsum: this lane's share of the new row's sum (smoothed fragments, PanelTriExtras)
#pragma unroll
	for (int i = 0; i < NCB; ++i)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int c = 32 * (wave + 4 * i) + 8 * q + 4 * half;
			const f32x4 num = *reinterpret_cast<const f32x4*>(s_num + l31 * LD + c);
			f32x4 o;
			if (MODE == PANEL_MU) {
				const f32x4 old = *reinterpret_cast<const f32x4*>(s_old + l31 * LD + c);
#pragma unroll
				for (int gi = 0; gi < 4; ++gi) o[gi] = old[gi] * num[gi] / (acc[i][4 * q + gi] + eps);
			} else {
#pragma unroll
				for (int gi = 0; gi < 4; ++gi) { const float d = acc[i][4 * q + gi]; o[gi] = d > 0.f ? d : 0.f; }
			}
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) psum += o[gi] * num[gi];
			if (NCB == 2) rsum += (o[0] + o[1]) + (o[2] + o[3]);
			nv[i][q] = o;
		}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < NCB; ++i)
#pragma unroll
		for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(s_old + l31 * LD + 32 * (wave + 4 * i) + 8 * q + 4 * half) = nv[i][q];
	psum += __shfl_xor(psum, 32);
	if (half == 0) s_ps[wave * 32 + l31] = psum;
	if (NCB == 2) {
		rsum += __shfl_xor(rsum, 32);
		if (half == 0) s_ps[128 + wave * 32 + l31] = rsum;
	}
	__syncthreads();

	// 4. coalesced write-out, per-row error terms, per-column sums of squares
	for (int e = tid; e < WIDE_YB * q4; e += 256) {
		const int y = e / q4, c4 = e - y * q4;
		*reinterpret_cast<f32x4*>(P + base + 4l * e) = *reinterpret_cast<const f32x4*>(s_old + y * LD + 4 * c4);
	}
	if (ps != nullptr && tid < WIDE_YB) {
		const int y = blockIdx.x * WIDE_YB + tid;
		if (y < len_valid) ps[y] = ((s_ps[tid] + s_ps[32 + tid]) + s_ps[64 + tid]) + s_ps[96 + tid];
	}
	if (sumsq_part != nullptr) {
		for (int c = tid; c < RP; c += 256) {
			float s = 0.f;
#pragma unroll 8
			for (int y = 0; y < WIDE_YB; ++y) { const float v = s_old[y * LD + c]; s += v * v; }
			sumsq_part[(long)blockIdx.x * RP + c] = s;
		}
	}
	if (FX && (fx.smooth_out != nullptr || fx.x3_out != nullptr)) {
		// what the consumers of the new rows read: the smoothed panel S new (the operand of V (S H)^T and of its Gram matrix) and the split image of the next
		// product's operand -- of the smoothed rows where there is smoothing, of the rows as they lie otherwise (W: unnormalised, its scale stays pending)
		const float* src = s_old;
		if (fxh && fx.smooth) {
			const int y = tid >> 3, sub = tid & 7;
			const float* row = s_old + y * LD;
			float sum = 0.f;
			for (int c = sub; c < fx.r; c += 8) sum += row[c];
#pragma unroll
			for (int w = 1; w < 8; w <<= 1) sum += __shfl_xor(sum, w);
			float* sm = s_num + y * LD;                  // (the numerators have been consumed)
			for (int c = sub; c < RP; c += 8) { const float x = row[c]; sm[c] = c < fx.r ? f_off * (sum - x) + f_diag * x : 0.f; }
			__syncthreads();
			src = s_num;
			if (fx.smooth_out != nullptr) {
				for (int e = tid; e < WIDE_YB * q4; e += 256) {
					const int y2 = e / q4, c4 = e - y2 * q4;
					*reinterpret_cast<f32x4*>(fx.smooth_out + base + 4l * e) = *reinterpret_cast<const f32x4*>(s_num + y2 * LD + 4 * c4);
				}
			}
		}
		if (fx.x3_out != nullptr) {
			// slot (kk, nb, h, r) of the workgroup's two K-steps: the eight rows 16 kk + 8 h + j of column 32 nb + r, split exactly into three bf16 planes
			const int NBT = RP / 32;
			for (int t = tid; t < 128 * NBT; t += 256) {
				const int r = t & 31, h = (t >> 5) & 1, rest = t >> 6, nb = rest % NBT, kk = rest / NBT;
				const long ks = (long)blockIdx.x * 2 + kk;
				if (ks < fx.x3_ks) {
					float v[8];
#pragma unroll
					for (int j = 0; j < 8; ++j) v[j] = src[(16 * kk + 8 * h + j) * LD + 32 * nb + r];
					store_split3(reinterpret_cast<bf16x8*>(fx.x3_out), ks, NBT, nb, h, r, v);
				}
			}
		}
	}
	if (NCB == 2 && tri.frag_out != nullptr) {
		// bf16 fragments of the 32 new rows (two K-steps): fragment (kk, cb, h, c) = rows 16 kk + 8 h .. + 7 of column 32 cb + c, at
		// [(K-step * 8 + cb) * 64 + 32 h + c] -- the layout k_finish_panel_bf16 writes; optionally smoothed like there:
		// f = frag_b * rowsum + frag_a * x for c < r (frag_a = diag - offdiag, frag_b = offdiag of the analytic S).
		// Thread (kk, h, cg): the eight rows of its (kk, h) as 16-byte LDS reads of columns 4 cg .. 4 cg + 3 -> four fragments, 64 contiguous bytes
		// per thread and 4 KB per wave on the way out.
		const bool smooth = tri.frag_b != 0.0f || tri.frag_a != 1.0f;
		const int cg = tid & 63, kh = tid >> 6, kk = kh >> 1, h = kh & 1;
		const long ks = (long)blockIdx.x * 2 + kk;
		if (ks < tri.frag_KS) {
			f32x4 x[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const f32x4*>(s_old + (16 * kk + 8 * h + j) * LD + 4 * cg);
			if (smooth) {
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const int y = 16 * kk + 8 * h + j;
					const float rs = ((s_ps[128 + y] + s_ps[160 + y]) + s_ps[192 + y]) + s_ps[224 + y];      // (columns >= r hold zeros)
#pragma unroll
					for (int i = 0; i < 4; ++i) x[j][i] = 4 * cg + i < tri.r ? tri.frag_b * rs + tri.frag_a * x[j][i] : 0.f;
				}
			}
			bf16x8* dst = reinterpret_cast<bf16x8*>(tri.frag_out) + (ks * 8 + (cg >> 3)) * 64 + 32 * h + 4 * (cg & 7);
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				bf16x8 v;
#pragma unroll
				for (int j = 0; j < 8; ++j) v[j] = (__bf16)x[j][i];
				dst[i] = v;
			}
		}
	}
}

// ------------------------------------------------------------------------------------------
// Gram matrix G = P P^T of a wide panel on the MFMA pipe (reference: syrk / gemm for W^T W and H H^T,
// AlgorithmMultiplicativeFrobenius.h:168-178,208-209; AlgorithmNonSmoothNMF.h:176,196,201)
// ------------------------------------------------------------------------------------------
// Workgroup = 4 waves = one 128 x 128 super-block (I <= J) of G over one slice of y; wave (wi, wj) owns the
// 64 x 64 quarter as 2 x 2 MFMA tiles.  K-step = 2 panel rows y (lane half h takes y = 2t + h): an operand is
// one coalesced 128-B read of P(y, 32 block + (l & 31)) per half-wave, and A and B operands have the same lane
// map, so a diagonal quarter needs two loads per K-step instead of four.  D-deep register ring.
// Every workgroup writes its 128 x 128 partial (and its transpose for I < J); partials are summed in slice
// order by k_reduce_partials.
template <int D>
__global__ __launch_bounds__(256, 2) void k_gram_wide_f32(const float* __restrict__ P, int RP, int len, int parts, float* __restrict__ partial) {
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int half = lane >> 5, l31 = lane & 31;
	// blockIdx.y enumerates the super-blocks of the upper triangle row by row
	const int nb = RP / 128;
	int I = 0, rem = blockIdx.y;
	while (rem >= nb - I) { rem -= nb - I; ++I; }
	const int J = I + rem;
	const int wi = wave >> 1, wj = wave & 1;
	const int ca = 128 * I + 64 * wi, cb = 128 * J + 64 * wj;      // first row / column of the quarter
	const int steps_total = (len + 1) / 2;
	const int s0 = (int)(((long)steps_total * blockIdx.x) / parts);
	const int s1 = (int)(((long)steps_total * (blockIdx.x + 1)) / parts);
	const int steps = s1 - s0;

	f32x16 acc[2][2];
#pragma unroll
	for (int a = 0; a < 2; ++a)
#pragma unroll
		for (int b = 0; b < 2; ++b)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

	if (steps > 0) {
		// rows past len read as zero: clamp the row and scale by 0 (len may be odd; panel rows past len are zero anyway
		// up to the padded length, which is even)
		const float* pa = P + ((long)2 * s0 + half) * RP + ca + l31;
		const float* pb = P + ((long)2 * s0 + half) * RP + cb + l31;
		const int last = steps - 1;
		float va[D][2], vb[D][2];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int t = d < last ? d : last;
			va[d][0] = pa[(long)2 * t * RP]; va[d][1] = pa[(long)2 * t * RP + 32];
			vb[d][0] = pb[(long)2 * t * RP]; vb[d][1] = pb[(long)2 * t * RP + 32];
		}
		__builtin_amdgcn_sched_barrier(0);
		int t = 0;
		for (; t + D <= steps; t += D) {
#pragma unroll
			for (int d = 0; d < D; ++d) {
#pragma unroll
				for (int a = 0; a < 2; ++a)
#pragma unroll
					for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[d][a], vb[d][b], acc[a][b], 0, 0, 0);
				int tn = t + D + d;
				tn = tn < last ? tn : last;
				va[d][0] = pa[(long)2 * tn * RP]; va[d][1] = pa[(long)2 * tn * RP + 32];
				vb[d][0] = pb[(long)2 * tn * RP]; vb[d][1] = pb[(long)2 * tn * RP + 32];
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		const int remn = steps - t;
#pragma unroll
		for (int d = 0; d < D; ++d) {
			if (d < remn) {
#pragma unroll
				for (int a = 0; a < 2; ++a)
#pragma unroll
					for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[d][a], vb[d][b], acc[a][b], 0, 0, 0);
			}
		}
	}
	// C/D map: register g of lane l is row (g&3) + 8*(g>>2) + 4*(l>>5) (A index), column l&31 (B index)
	float* out = partial + (long)blockIdx.x * RP * RP;
#pragma unroll
	for (int a = 0; a < 2; ++a)
#pragma unroll
		for (int b = 0; b < 2; ++b)
#pragma unroll
			for (int g = 0; g < 16; ++g) {
				const int r = ca + 32 * a + (g & 3) + 8 * (g >> 2) + 4 * half;
				const int c = cb + 32 * b + l31;
				out[(long)r * RP + c] = acc[a][b][g];
				if (I != J) out[(long)c * RP + r] = acc[a][b][g];
			}
}

// (the slice's body: gram_wide.h -- shared with the passenger workgroups of the product launch)
template <int D, bool MIRROR = true>
__global__ __launch_bounds__(256, 2) void k_gram_wide_x3(const float* __restrict__ P, int RP, int len, int parts, float* __restrict__ partial) {
	gram_wide_slice<D, MIRROR>(P, RP, len, parts, partial, (int)blockIdx.x, (int)blockIdx.y);
}

bool gram_wide_available(int RP) { return RP >= 128 && RP % 128 == 0; }

// len: valid panel rows (an odd len reads one zero row of the padding); partial: parts * RP * RP elements of scratch
hipError_t launch_gram_wide_f32(const float* P, int RP, int len, int parts, float* partial, float* G, hipStream_t stream) {
	if (!gram_wide_available(RP)) return hipErrorInvalidValue;
	const int nb = RP / 128, nsuper = nb * (nb + 1) / 2;
	// two workgroups per CU are enough; fewer, longer slices keep the partial traffic down
	static const bool native = tuning_env("NMFAMD_WIDE_FP32_MFMA") != nullptr;       // A/B switch: fp32 MFMA instructions
	static const int wgs = [] { const char* e = tuning_env("NMFAMD_GRAM_WGS"); return e ? std::atoi(e) : 0; }();
	const int target = wgs > 0 ? wgs : 512;
	parts = std::max(1, std::min(std::min(parts, std::max(16, target / nsuper)), std::max(1, len / 64)));      // and at least 32 K-steps per slice
	if (native) hipLaunchKernelGGL((k_gram_wide_f32<8>), dim3(parts, nsuper), dim3(256), 0, stream, P, RP, len, parts, partial);
	else hipLaunchKernelGGL((k_gram_wide_x3<2>), dim3(parts, nsuper), dim3(256), 0, stream, P, RP, len, parts, partial);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	return launch_reduce_partials<float>(partial, parts, (long)RP * RP, G, (long)RP * RP, stream);
}

// The slices of k_gram_wide_x3 reduced, and in the same launch what the update kernel reads: the matrix's split image (A(c, k) = G(k, c): k_pack_panel_x3's layout
// of G as a panel -- launch_panel_update_wide_f32 with Q == nullptr) and, W side, the pending column scale d(c) = 1 / sqrt(sum of the W update's per-workgroup sums
// of squares) (kernel::normalizeColumns as a factor, KernelNormalizeColumns.cu:37-58).  The generic sequence ran k_reduce_partials, k_pack_panel_x3 (and, for W,
// k_compact_partials + k_normalize_panel_v2) for this.  One workgroup per 8 x 32 tile of G (a first version with 32 x 32 tiles -- 16 workgroups at rank 128, 0.5 MB
// of partials through each CU -- took 21.6 us): the four waves add a quarter of the slices each, eight loads in flight, quarters added in order; RP / 64 more
// workgroups for the scale.
__global__ __launch_bounds__(256) void k_gram_reduce_x3(const float* __restrict__ partial, int parts, int RP, float* __restrict__ G, bf16x8* __restrict__ qx3,
                                                        const float* __restrict__ sumsq_part, int sq_parts, float* __restrict__ scale_out) {
	__shared__ __attribute__((aligned(16))) float s_p[4 * 8 * 32];
	__shared__ float s_t[8 * 33];
	const int tid = threadIdx.x;
	const int nbt = RP / 32, tiles = (RP / 8) * nbt;
	const int g = tid >> 6, t64 = tid & 63;
	if ((int)blockIdx.x >= tiles) {
		if (sumsq_part == nullptr) return;
		const int c = 64 * ((int)blockIdx.x - tiles) + t64;
		const int p0 = (int)(((long)sq_parts * g) / 4), p1 = (int)(((long)sq_parts * (g + 1)) / 4);
		float s = 0.f;
		for (int p = p0; p < p1; p += 8) {
			float v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = sumsq_part[(long)(p + u < p1 ? p + u : p0) * RP + c];
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (p + u < p1) s += v[u];
		}
		s_p[tid] = s;
		__syncthreads();
		if (g == 0) {
			const float t = ((s_p[tid] + s_p[64 + tid]) + s_p[128 + tid]) + s_p[192 + tid];
			scale_out[c] = t > 0.f ? 1.0f / sqrtf(t) : 1.0f;
		}
		return;
	}
	// tile = eight rows k x 32 columns c of G (one (K-step, half) of one column block of the split image); the four waves take a quarter of the slices each
	const int ti = (int)blockIdx.x / nbt, tj = (int)blockIdx.x % nbt;
	// The slices hold the 32 x 32 blocks on and above the diagonal (k_gram_wide_x3<.., false>): a tile above the diagonal blocks also writes its mirror image --
	// G(c, k) and the image slots of those rows -- and the tiles below have nothing to do.
	if ((ti >> 2) > tj) return;
	const bool mirror = (ti >> 2) < tj;
	const int rr = t64 >> 3, c4 = 4 * (t64 & 7);
	const long e = (long)(8 * ti + rr) * RP + 32 * tj + c4, stride = (long)RP * RP;
	const int p0 = (int)(((long)parts * g) / 4), p1 = (int)(((long)parts * (g + 1)) / 4);
	f32x4 sum = {0.f, 0.f, 0.f, 0.f};
	for (int p = p0; p < p1; p += 8) {
		f32x4 v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(partial + (long)(p + u < p1 ? p + u : p0) * stride + e);
#pragma unroll
		for (int u = 0; u < 8; ++u)
			if (p + u < p1) sum += v[u];
	}
	*reinterpret_cast<f32x4*>(s_p + g * 256 + 4 * t64) = sum;
	__syncthreads();
	if (g == 0) {
		f32x4 o = *reinterpret_cast<const f32x4*>(s_p + 4 * t64);
#pragma unroll
		for (int k = 1; k < 4; ++k) o += *reinterpret_cast<const f32x4*>(s_p + k * 256 + 4 * t64);
		*reinterpret_cast<f32x4*>(G + e) = o;
#pragma unroll
		for (int k = 0; k < 4; ++k) s_t[rr * 33 + c4 + k] = o[k];
	}
	__syncthreads();
	if (qx3 != nullptr && tid < 32) {
		float v8[8];
#pragma unroll
		for (int kk = 0; kk < 8; ++kk) v8[kk] = s_t[kk * 33 + tid];            // rows k = 8 ti + kk of column c = 32 tj + tid
		store_split3(qx3, (8 * ti) >> 4, nbt, tj, ti & 1, tid, v8);
	}
	if (mirror) {
		// G(k' = 32 tj + j, c' = 8 ti + i) = tile(i, j): 32 rows of eight values; image slots: K-step 2 tj + jg / 2, half jg & 1 of column block ti / 4, lane 8 (ti % 4) + i
		if (tid < 64) {
			const int j = tid >> 1, i4 = 4 * (tid & 1);
			f32x4 o;
#pragma unroll
			for (int k = 0; k < 4; ++k) o[k] = s_t[(i4 + k) * 33 + j];
			*reinterpret_cast<f32x4*>(G + (long)(32 * tj + j) * RP + 8 * ti + i4) = o;
		} else if (qx3 != nullptr && tid < 96) {
			const int t = tid - 64, i = t & 7, jg = t >> 3;
			float v8[8];
#pragma unroll
			for (int kk = 0; kk < 8; ++kk) v8[kk] = s_t[i * 33 + 8 * jg + kk];
			store_split3(qx3, 2 * tj + (jg >> 1), nbt, ti >> 2, jg & 1, 8 * (ti & 3) + i, v8);
		}
	}
}

// G, its split image and (sumsq_part != nullptr) the pending column scale in TWO launches: slices, then k_gram_reduce_x3
int gram_wide_fused_parts(int RP, int len, int parts) {
	const int nb = RP / 128, nsuper = nb * (nb + 1) / 2;
	return std::max(1, std::min(std::min(parts, std::max(16, 512 / nsuper)), std::max(1, len / 64)));
}

hipError_t launch_gram_reduce_x3(const float* partial, int parts, int RP, float* G, void* qx3, const float* sumsq_part, int sq_parts, float* scale_out, hipStream_t stream) {
	if (!gram_wide_available(RP) || qx3 == nullptr || parts < 1) return hipErrorInvalidValue;
	const int nbt = RP / 32;
	hipLaunchKernelGGL(k_gram_reduce_x3, dim3((RP / 8) * nbt + (sumsq_part != nullptr ? RP / 64 : 0)), dim3(256), 0, stream, partial, parts, RP, G, reinterpret_cast<bf16x8*>(qx3),
	                   sumsq_part, sq_parts, scale_out);
	return hipGetLastError();
}

hipError_t launch_gram_wide_fused_f32(const float* P, int RP, int len, int parts, float* partial, float* G, void* qx3, const float* sumsq_part, int sq_parts,
                                      float* scale_out, hipStream_t stream) {
	if (!gram_wide_available(RP) || qx3 == nullptr) return hipErrorInvalidValue;
	const int nb = RP / 128, nsuper = nb * (nb + 1) / 2;
	parts = gram_wide_fused_parts(RP, len, parts);
	hipLaunchKernelGGL((k_gram_wide_x3<2, false>), dim3(parts, nsuper), dim3(256), 0, stream, P, RP, len, parts, partial);
	if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
	return launch_gram_reduce_x3(partial, parts, RP, G, qx3, sumsq_part, sq_parts, scale_out, stream);
}


// ------------------------------------------------------------------------------------------
// The same at padded rank 64 (nsNMF, GDCLS and the least-squares family at r <= 64; the multiplicative update has
// its own fused kernel, kernels_mu64.hip): 64 panel rows per workgroup staged in LDS, wave (cb, yh) owns the 32
// columns 32 cb + i of the 32 rows 32 yh + j; all eight K groups of the A operand are requested up front.
// Replaces k_panel_update64_f32 (kernels_fast.hip), whose row-per-lane global loads were texture-addresser bound.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_panel_update64_lds_f32(
	float* __restrict__ P, const float* __restrict__ slabs, int S, long slab_stride,
	const float* __restrict__ Q, float eps, float* __restrict__ ps, int len_valid,
	float* __restrict__ sumsq_part, float* __restrict__ num_out, float* __restrict__ gram_partial,
	bf16x8* __restrict__ x3_out, int x3_ks) {
	constexpr int YB = 64, LD = 68;
	__shared__ __attribute__((aligned(16))) float s_num[YB * LD];
	__shared__ __attribute__((aligned(16))) float s_old[YB * LD];
	__shared__ float s_ps[2][YB];
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const int cb = wave & 1, yh = wave >> 1;
	const long base = (long)blockIdx.x * YB * 64;

	float qa[8][4];
#pragma unroll
	for (int u = 0; u < 8; ++u)
#pragma unroll
		for (int gi = 0; gi < 4; ++gi) qa[u][gi] = Q[(long)(8 * u + 4 * half + gi) * 64 + 32 * cb + l31];

	{
		f32x4 num[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) num[i] = *reinterpret_cast<const f32x4*>(slabs + base + 4l * (tid + 256 * i));
		if (MODE == PANEL_MU) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int e = tid + 256 * i, y = e >> 4, c4 = e & 15;
				*reinterpret_cast<f32x4*>(s_old + y * LD + 4 * c4) = *reinterpret_cast<const f32x4*>(P + base + 4l * e);
			}
		}
		// the K slices in batches of up to five, requested together and added in slice order (round 4: one slice per turn of a loop was one dependent round
		// trip per slice -- five of them for the six slices of config 5's W^T V)
		for (int k0 = 1; k0 < S; k0 += 5) {
			f32x4 t[5][4];
#pragma unroll
			for (int u = 0; u < 5; ++u) {
				const int k = k0 + u < S ? k0 + u : 0;      // clamped duplicate, discarded below
#pragma unroll
				for (int i = 0; i < 4; ++i) t[u][i] = *reinterpret_cast<const f32x4*>(slabs + (long)k * slab_stride + base + 4l * (tid + 256 * i));
			}
#pragma unroll
			for (int u = 0; u < 5; ++u)
				if (k0 + u < S) {
#pragma unroll
					for (int i = 0; i < 4; ++i) num[i] += t[u][i];
				}
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int e = tid + 256 * i, y = e >> 4, c4 = e & 15;
			*reinterpret_cast<f32x4*>(s_num + y * LD + 4 * c4) = num[i];
			if (num_out) *reinterpret_cast<f32x4*>(num_out + base + 4l * e) = num[i];
		}
	}
	__syncthreads();

	const float* vec = (MODE == PANEL_MU ? s_old : s_num) + (32 * yh + l31) * LD + 4 * half;
	f32x16 acc;
#pragma unroll
	for (int g = 0; g < 16; ++g) acc[g] = 0.f;
#pragma unroll
	for (int u = 0; u < 8; ++u) {
		const f32x4 b = *reinterpret_cast<const f32x4*>(vec + 8 * u);
#pragma unroll
		for (int gi = 0; gi < 4; ++gi) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[u][gi], b[gi], acc, 0, 0, 0);
	}

	f32x4 nv[4];
	float psum = 0.f;
	const int yrow = 32 * yh + l31;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const int c = 32 * cb + 8 * q + 4 * half;
		const f32x4 num = *reinterpret_cast<const f32x4*>(s_num + yrow * LD + c);
		f32x4 o;
		if (MODE == PANEL_MU) {
			const f32x4 old = *reinterpret_cast<const f32x4*>(s_old + yrow * LD + c);
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) o[gi] = old[gi] * num[gi] / (acc[4 * q + gi] + eps);
		} else {
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) { const float d = acc[4 * q + gi]; o[gi] = d > 0.f ? d : 0.f; }
		}
#pragma unroll
		for (int gi = 0; gi < 4; ++gi) psum += o[gi] * num[gi];
		nv[q] = o;
	}
	__syncthreads();
#pragma unroll
	for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(s_old + yrow * LD + 32 * cb + 8 * q + 4 * half) = nv[q];
	psum += __shfl_xor(psum, 32);
	if (half == 0) s_ps[cb][yrow] = psum;
	__syncthreads();

#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int e = tid + 256 * i, y = e >> 4, c4 = e & 15;
		*reinterpret_cast<f32x4*>(P + base + 4l * e) = *reinterpret_cast<const f32x4*>(s_old + y * LD + 4 * c4);
	}
	if (ps != nullptr && tid < YB) {
		const int y = blockIdx.x * YB + tid;
		if (y < len_valid) ps[y] = s_ps[0][tid] + s_ps[1][tid];
	}
	if (sumsq_part != nullptr && tid < 64) {
		float s = 0.f;
#pragma unroll 8
		for (int y = 0; y < YB; ++y) { const float v = s_old[y * LD + tid]; s += v * v; }
		sumsq_part[(long)blockIdx.x * 64 + tid] = s;
	}
	if (x3_out != nullptr) {
		// the split (3 x bf16) image of the new rows for the next factor product (as in k_mu64_update): four K-steps of
		// 16 rows per tile, two (K-step, column block, half, lane) slots per thread
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const int slot = tid + 256 * i;
			const int r = slot & 31, h = (slot >> 5) & 1, nb = (slot >> 6) & 1, kk = slot >> 7;
			const long ks = 4l * blockIdx.x + kk;
			if (ks < x3_ks) {
				float v[8];
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const int yl = 16 * kk + 8 * h + j;
					v[j] = blockIdx.x * YB + yl < len_valid ? s_old[yl * LD + 32 * nb + r] : 0.f;
				}
				store_split3(x3_out, ks, 2, nb, h, r, v);
			}
		}
	}
	if (gram_partial != nullptr) {
		// partial Gram matrix of the 64 new rows (layout of k_mu64_update: one 64 x 64 matrix per workgroup), so that the
		// caller can reduce W^T W / H H^T with the 16-block reduction instead of a pass over the panel
		const int ab = wave >> 1, bb = wave & 1;
		f32x16 g;
#pragma unroll
		for (int i = 0; i < 16; ++i) g[i] = 0.f;
#pragma unroll 8
		for (int x = 0; x < YB; x += 2) {
			const float a = s_old[(x + half) * LD + ab * 32 + l31];
			const float b = s_old[(x + half) * LD + bb * 32 + l31];
			g = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, g, 0, 0, 0);
		}
		float* out = gram_partial + (long)blockIdx.x * 4096;
#pragma unroll
		for (int q = 0; q < 4; ++q)
#pragma unroll
			for (int gi = 0; gi < 4; ++gi) out[(long)(ab * 32 + gi + 8 * q + 4 * half) * 64 + bb * 32 + l31] = g[4 * q + gi];
	}
}

// 64 panel rows per workgroup: len_pad / 64 norm partials
hipError_t launch_panel_update64_lds_f32(int mode, float* P, const float* slabs, int S, long slab_stride, const float* Q, int len_pad,
                                         float eps, float* ps, int len_valid, float* sumsq_part, float* num_out, hipStream_t stream, float* gram_partial,
                                         void* x3_out, int x3_ks) {
	if ((mode != PANEL_MU && mode != PANEL_LS) || len_pad % 64 != 0) return hipErrorInvalidValue;
	dim3 grid(len_pad / 64), block(256);
	bf16x8* xo = reinterpret_cast<bf16x8*>(x3_out);
	if (mode == PANEL_MU) hipLaunchKernelGGL((k_panel_update64_lds_f32<PANEL_MU>), grid, block, 0, stream, P, slabs, S, slab_stride, Q, eps, ps, len_valid, sumsq_part, num_out, gram_partial, xo, x3_ks);
	else hipLaunchKernelGGL((k_panel_update64_lds_f32<PANEL_LS>), grid, block, 0, stream, P, slabs, S, slab_stride, Q, eps, ps, len_valid, sumsq_part, num_out, gram_partial, xo, x3_ks);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// MODE_MU on long panels (the W update of config 4: 50 000 x 256): 64 panel rows per workgroup.
// k_panel_update_wide_f32 streams the whole split image of Q (6 RP^2 bytes: 393 KB at rank 256) from L2 once per 32 rows --
// 614 MB of L2 traffic per W update, as much time as the MFMAs themselves.  Here a workgroup covers two row blocks per
// fragment of Q (half the L2 traffic, two independent accumulator chains per fragment), keeps ONE fp32 image in LDS (the old
// values: B operand, then the new values for the coalesced write-out) and takes the numerator straight from global memory in
// the C/D layout, requested before the MFMA loop.  Same arithmetic and summation order per element as k_panel_update_wide_f32
// with split operands; the norm partials stay one vector per 32 rows (panel_update_parts()).
// ------------------------------------------------------------------------------------------
template <int NCB>
__global__ __launch_bounds__(256, 2) void k_panel_update_wide64_mu(
	const float* P, float* Pout, const float* __restrict__ slabs, int S, long slab_stride, int RP, float eps, float* __restrict__ ps, int len_valid,
	float* __restrict__ sumsq_part, const bf16x8* __restrict__ Qx3) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	constexpr int YB = 64;
	const int LD = RP + 4;
	float* s_old = lds;                       // [64][LD]
	float* s_ps = lds + YB * LD;              // [4][64]
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const long base = (long)blockIdx.x * YB * RP;
	const int q4 = RP / 4;

	// The old panel values reach LDS one K-step ahead of the MFMAs that read them: the 64 x 16 chunk of K-step u is one 16-byte
	// load per thread (row tid / 4, piece tid % 4), DC chunks in flight.  A staging phase of the whole tile in front of the MFMA
	// loop made every workgroup of the launch wait for HBM at the same time and multiply at the same time.
	constexpr int DC = 4, DX = 2;
	const int ksteps = RP / 16, NBT = RP / 32;
	const float* gsrc = P + base + (long)(tid >> 2) * RP + 4 * (tid & 3);
	float* sdst = s_old + (tid >> 2) * LD + 4 * (tid & 3);
	f32x4 ch[DC];
#pragma unroll
	for (int d = 0; d < DC; ++d) ch[d] = *reinterpret_cast<const f32x4*>(gsrc + 16 * d);

	f32x16 acc[2][NCB];
#pragma unroll
	for (int rb = 0; rb < 2; ++rb)
#pragma unroll
		for (int i = 0; i < NCB; ++i)
#pragma unroll
			for (int g = 0; g < 16; ++g) acc[rb][i][g] = 0.f;
	const bf16x8* qf = Qx3 + (long)wave * 192 + lane;        // + (ks * NBT + 4 i) * 192 + plane * 64
	const float* vb = s_old + l31 * LD + 8 * half;
	bf16x8 af[DX][NCB][3];
#pragma unroll
	for (int d = 0; d < DX; ++d)
#pragma unroll
		for (int i = 0; i < NCB; ++i)
#pragma unroll
			for (int pl = 0; pl < 3; ++pl) af[d][i][pl] = qf[((long)d * NBT + 4 * i) * 192 + pl * 64];
	*reinterpret_cast<f32x4*>(sdst) = ch[0];
	ch[0] = *reinterpret_cast<const f32x4*>(gsrc + 16 * DC);

	// D(c, y) = sum_k Q(k, c) old(y, k), six bf16 MFMAs per 16 k on exactly split operands
	for (int u = 0; u < ksteps; u += DC) {
#pragma unroll
		for (int d = 0; d < DC; ++d) {
			const int ks = u + d;
			__syncthreads();                                  // chunk ks is in LDS
			if (ks + 1 < ksteps) {
				*reinterpret_cast<f32x4*>(sdst + 16 * (ks + 1)) = ch[(d + 1) % DC];
				int nc = ks + 1 + DC;
				nc = nc < ksteps ? nc : ksteps - 1;           // tail: harmless re-load
				ch[(d + 1) % DC] = *reinterpret_cast<const f32x4*>(gsrc + 16 * nc);
			}
			bf16x8 hi[2], mid[2], lo[2];
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
				float v[8];
				const f32x4 b0 = *reinterpret_cast<const f32x4*>(vb + 32 * rb * LD + 16 * ks);
				const f32x4 b1 = *reinterpret_cast<const f32x4*>(vb + 32 * rb * LD + 16 * ks + 4);
#pragma unroll
				for (int j = 0; j < 4; ++j) { v[j] = b0[j]; v[4 + j] = b1[j]; }
				split3(v, hi[rb], mid[rb], lo[rb]);
			}
			const int dq = d % DX;
#pragma unroll
			for (int i = 0; i < NCB; ++i)
#pragma unroll
				for (int rb = 0; rb < 2; ++rb) {
					acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[dq][i][2], hi[rb], acc[rb][i], 0, 0, 0);
					acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[dq][i][0], lo[rb], acc[rb][i], 0, 0, 0);
					acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[dq][i][1], mid[rb], acc[rb][i], 0, 0, 0);
					acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[dq][i][1], hi[rb], acc[rb][i], 0, 0, 0);
					acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[dq][i][0], mid[rb], acc[rb][i], 0, 0, 0);
					acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[dq][i][0], hi[rb], acc[rb][i], 0, 0, 0);
				}
			int nu = ks + DX;
			nu = nu < ksteps ? nu : ksteps - 1;       // tail: harmless re-load of the last K-step
#pragma unroll
			for (int i = 0; i < NCB; ++i)
#pragma unroll
				for (int pl = 0; pl < 3; ++pl) af[dq][i][pl] = qf[((long)nu * NBT + 4 * i) * 192 + pl * 64];
		}
	}

	// the numerator in the C/D layout: lane (y = l31 + 32 rb, half) owns c = 32 (wave + 4 i) + 8 q + 4 half + gi
	f32x4 num[2][NCB][4];
#pragma unroll
	for (int rb = 0; rb < 2; ++rb)
#pragma unroll
		for (int i = 0; i < NCB; ++i)
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const long off = base + (long)(32 * rb + l31) * RP + 32 * (wave + 4 * i) + 8 * q + 4 * half;
				f32x4 v = *reinterpret_cast<const f32x4*>(slabs + off);
				for (int k = 1; k < S; ++k) v += *reinterpret_cast<const f32x4*>(slabs + (long)k * slab_stride + off);
				num[rb][i][q] = v;
			}

	// element-wise step in the C/D layout
	f32x4 nv[2][NCB][4];
	float psum[2] = {0.f, 0.f};
#pragma unroll
	for (int rb = 0; rb < 2; ++rb)
#pragma unroll
		for (int i = 0; i < NCB; ++i)
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int c = 32 * (wave + 4 * i) + 8 * q + 4 * half;
				const f32x4 old = *reinterpret_cast<const f32x4*>(s_old + (32 * rb + l31) * LD + c);
				f32x4 o;
#pragma unroll
				for (int gi = 0; gi < 4; ++gi) o[gi] = old[gi] * num[rb][i][q][gi] / (acc[rb][i][4 * q + gi] + eps);
#pragma unroll
				for (int gi = 0; gi < 4; ++gi) psum[rb] += o[gi] * num[rb][i][q][gi];
				nv[rb][i][q] = o;
			}
	__syncthreads();
#pragma unroll
	for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
		for (int i = 0; i < NCB; ++i)
#pragma unroll
			for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(s_old + (32 * rb + l31) * LD + 32 * (wave + 4 * i) + 8 * q + 4 * half) = nv[rb][i][q];
		psum[rb] += __shfl_xor(psum[rb], 32);
		if (half == 0) s_ps[wave * 64 + 32 * rb + l31] = psum[rb];
	}
	__syncthreads();

	// coalesced write-out, per-row error terms, per-column sums of squares (one vector per 32 rows)
	for (int e = tid; e < YB * q4; e += 256) {
		const int y = e / q4, c4 = e - y * q4;
		*reinterpret_cast<f32x4*>(Pout + base + 4l * e) = *reinterpret_cast<const f32x4*>(s_old + y * LD + 4 * c4);
	}
	if (ps != nullptr && tid < YB) {
		const int y = blockIdx.x * YB + tid;
		if (y < len_valid) ps[y] = ((s_ps[tid] + s_ps[64 + tid]) + s_ps[128 + tid]) + s_ps[192 + tid];
	}
	if (sumsq_part != nullptr) {
		for (int e = tid; e < 2 * RP; e += 256) {
			const int rb = e / RP, c = e - rb * RP;
			float s = 0.f;
#pragma unroll 8
			for (int y = 0; y < 32; ++y) { const float v = s_old[(32 * rb + y) * LD + c]; s += v * v; }
			sumsq_part[((long)blockIdx.x * 2 + rb) * RP + c] = s;
		}
	}
}

template <int NCB>
static hipError_t launch_wide64(const float* P, float* Pout, const float* slabs, int S, long slab_stride, int RP, int len_pad, float eps, float* ps, int len_valid,
                                float* sumsq_part, hipStream_t stream, const void* qx3) {
	const size_t lds_bytes = sizeof(float) * (64 * (size_t)(RP + 4) + 256);
	const size_t max_bytes = sizeof(float) * (64 * (size_t)(128 * NCB + 4) + 256);
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_panel_update_wide64_mu<NCB>), (int)max_bytes, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL((k_panel_update_wide64_mu<NCB>), dim3(len_pad / 64), dim3(256), lds_bytes, stream,
	                   P, Pout, slabs, S, slab_stride, RP, eps, ps, len_valid, sumsq_part, reinterpret_cast<const bf16x8*>(qx3));
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// MODE_MU on long panels, second form: a wave owns 32 whole panel rows.
// The kernels above let the four waves of a workgroup share the panel rows (B operand, through LDS) and split the columns:
// every wave splits the same rows again, the panel tile is staged and written out through LDS, and with two such workgroups per
// CU the launch is a sum of serial chains (profiles/r02_c4_kernel_experiments.md: 65 us, 37 us with everything but the skeleton
// removed, against 25 us of HBM time and 20 us of MFMA time).  Here the waves of a workgroup share Q instead:
//   * the K-step's fragments of the split image of Q (6 RP bytes per k: 24 KB at rank 256) go through a two-slot LDS ring, loaded
//     once per workgroup and K-step, read by all four waves;
//   * a wave takes its own 32 rows straight from global memory (a lane: row l & 31, the 8 k of its half: two 16-byte loads per
//     K-step, four K-steps in flight), splits them ONCE, and multiplies them with all RP / 32 column blocks: one split per
//     48 MFMAs instead of one per 12;
//   * the MFMA computes D(y, c) = sum_k old(y, k) Q(k, c) with the ROWS as its M index: a lane ends up with 16 rows of ONE column
//     per block, so old / num / new values are 4-byte accesses that are contiguous across lanes (128 B per row and block), the
//     column sums of squares are in-lane sums plus one cross-half exchange, and nothing is staged through LDS.
// Same value per element as k_panel_update_wide64_mu (same six-term product, same K order).
// Measured at config 4 (profiles/r02_c4_kernel_experiments.md): 58 us against 65 - 68; the parts still add up instead of overlapping
// (MFMAs 16.5, epilogue 17, staging of Q 13, operand split 8.5, skeleton 16), and delaying the second workgroup of each CU by
// 6 .. 18 us made it slower, not faster.
// ------------------------------------------------------------------------------------------
// A_BF16 (PanelTriExtras::old_as_bf16, the rank-256 bf16 mode): the old rows enter the r x r product rounded to bf16 -- the very values the product against V
// multiplies with -- so the operand split and three of the six MFMAs per tile go; Q keeps its three planes.  The element-wise step still uses the fp32 rows.
template <int NC, bool HAS_PS, bool A_BF16>          // NC: column blocks of 32 (RP / 32); HAS_PS: per-row error terms wanted
__global__ __launch_bounds__(256, 2) void k_panel_update_rows_mu(
	const float* P, float* Pout, const float* __restrict__ slabs, int S, long slab_stride, float eps, float* __restrict__ ps, int len_valid,
	float* __restrict__ sumsq_part, const bf16x8* __restrict__ Qx3, const float* __restrict__ old_colsq, int old_colsq_parts, bf16x8* __restrict__ frag_out, long frag_KS) {
	constexpr int RP = 32 * NC, KSTEPS = RP / 16, FR = NC * 192;      // fragments (16 B) of Q per K-step
	constexpr int QL = FR / 256;                                      // Q fragments per thread and K-step (6 at rank 256, 3 at 128)
	static_assert(FR % 256 == 0, "whole fragments per thread");
	__shared__ __attribute__((aligned(16))) bf16x8 s_q[2][FR];
	__shared__ float s_sq[4][RP];
	__shared__ __attribute__((aligned(16))) float s_d[RP];           // the panel's pending column scale (PanelTriExtras::old_colsq): old values are old * d
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int half = lane >> 5, l31 = lane & 31;
	const bool scaled = old_colsq != nullptr;                         // (kernel argument: uniform)
	if (scaled) for (int c = tid; c < RP; c += 256) s_d[c] = tri_pending_scale(old_colsq, old_colsq_parts, RP, c);
	const long row0 = (long)blockIdx.x * 128 + 32 * wave;            // this wave's rows (wave-uniform)
	const float* prow = P + row0 * RP + (l31 * RP + 8 * half);        // + 16 u: the 8 k of this lane in K-step u

	constexpr int DA = 4;                                             // K-steps of the wave's own rows in flight
	f32x4 ra[DA][2];
#pragma unroll
	for (int d = 0; d < DA; ++d) { ra[d][0] = *reinterpret_cast<const f32x4*>(prow + 16 * d); ra[d][1] = *reinterpret_cast<const f32x4*>(prow + 16 * d + 4); }
	bf16x8 rq[QL];
#pragma unroll
	for (int i = 0; i < QL; ++i) rq[i] = Qx3[tid + 256 * i];
#pragma unroll
	for (int i = 0; i < QL; ++i) s_q[0][tid + 256 * i] = rq[i];
#pragma unroll
	for (int i = 0; i < QL; ++i) rq[i] = Qx3[(long)FR + tid + 256 * i];

	f32x16 acc[NC];
#pragma unroll
	for (int cb = 0; cb < NC; ++cb)
#pragma unroll
		for (int g = 0; g < 16; ++g) acc[cb][g] = 0.f;

	bf16x8 nh, nm, nl;
	if (scaled) __syncthreads();
	{
		float v[8];
#pragma unroll
		for (int j = 0; j < 4; ++j) { v[j] = ra[0][0][j]; v[4 + j] = ra[0][1][j]; }
		if (scaled) {
			const f32x4 d0 = *reinterpret_cast<const f32x4*>(&s_d[8 * half]), d1 = *reinterpret_cast<const f32x4*>(&s_d[8 * half + 4]);
#pragma unroll
			for (int j = 0; j < 4; ++j) { v[j] *= d0[j]; v[4 + j] *= d1[j]; }
		}
		if (A_BF16) {
#pragma unroll
			for (int j = 0; j < 8; ++j) nh[j] = (__bf16)v[j];
		} else split3(v, nh, nm, nl);
	}
	for (int u = 0; u < KSTEPS; u += DA) {
#pragma unroll
		for (int d = 0; d < DA; ++d) {
			const int ks = u + d;
			__syncthreads();                                  // Q of K-step ks is in slot ks & 1; slot (ks + 1) & 1 is free
			if (ks + 1 < KSTEPS) {
#pragma unroll
				for (int i = 0; i < QL; ++i) s_q[(d + 1) & 1][tid + 256 * i] = rq[i];
				int nq = ks + 2;
				nq = nq < KSTEPS ? nq : KSTEPS - 1;           // tail: harmless re-load
#pragma unroll
				for (int i = 0; i < QL; ++i) rq[i] = Qx3[(long)nq * FR + tid + 256 * i];
			}
			// this K-step's operand was split during the previous one; the next one's is split among this step's MFMAs
			const bf16x8 hi = nh, mid = nm, lo = nl;
			{
				float v[8];
#pragma unroll
				for (int j = 0; j < 4; ++j) { v[j] = ra[(d + 1) % DA][0][j]; v[4 + j] = ra[(d + 1) % DA][1][j]; }
				if (scaled) {
					const int kn = ks + 1 < KSTEPS ? ks + 1 : KSTEPS - 1;      // (the K-step these values belong to)
					const f32x4 d0 = *reinterpret_cast<const f32x4*>(&s_d[16 * kn + 8 * half]), d1 = *reinterpret_cast<const f32x4*>(&s_d[16 * kn + 8 * half + 4]);
#pragma unroll
					for (int j = 0; j < 4; ++j) { v[j] *= d0[j]; v[4 + j] *= d1[j]; }
				}
				if (A_BF16) {
#pragma unroll
					for (int j = 0; j < 8; ++j) nh[j] = (__bf16)v[j];
				} else split3(v, nh, nm, nl);
				int na = ks + DA;
				na = na < KSTEPS ? na : KSTEPS - 1;
				ra[d][0] = *reinterpret_cast<const f32x4*>(prow + 16 * na); ra[d][1] = *reinterpret_cast<const f32x4*>(prow + 16 * na + 4);
			}
			const bf16x8* q = &s_q[d & 1][lane];
			// the fragments of block cb + 1 are read while block cb multiplies (two sets of three live, not NC)
			bf16x8 qa[2][3];
			qa[0][0] = q[0]; qa[0][1] = q[64]; qa[0][2] = q[128];
#pragma unroll
			for (int cb = 0; cb < NC; ++cb) {
				if (cb + 1 < NC) { qa[(cb + 1) & 1][0] = q[(cb + 1) * 192]; qa[(cb + 1) & 1][1] = q[(cb + 1) * 192 + 64]; qa[(cb + 1) & 1][2] = q[(cb + 1) * 192 + 128]; }
				const bf16x8 q0 = qa[cb & 1][0], q1 = qa[cb & 1][1], q2 = qa[cb & 1][2];
				// smallest terms first; A = the panel rows (M = y), B = Q (N = c)
				acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, q2, acc[cb], 0, 0, 0);
				if (!A_BF16) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, q0, acc[cb], 0, 0, 0);
				if (!A_BF16) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mid, q1, acc[cb], 0, 0, 0);
				acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, q1, acc[cb], 0, 0, 0);
				if (!A_BF16) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mid, q0, acc[cb], 0, 0, 0);
				acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, q0, acc[cb], 0, 0, 0);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}

	// element-wise step: register g of lane l is row (g & 3) + 8 (g >> 2) + 4 (l >> 5) of the wave's 32, column 32 cb + (l & 31):
	// wave-uniform bases + one lane offset, 128 contiguous bytes per row and block
	const float* pw = P + row0 * RP;
	const float* nw = slabs + row0 * RP;
	float* ow = Pout + row0 * RP;
	const int lofs = 4 * half * RP + l31;
	float rowdot[HAS_PS ? 16 : 1];
#pragma unroll
	for (int g = 0; g < (HAS_PS ? 16 : 1); ++g) rowdot[g] = 0.f;
	// the values of block cb + 1 are requested before block cb is worked on: the numerator comes from HBM, and eight
	// load -> divide -> store chains in a row would each expose that latency
	float oldv[2][16], numv[2][16];
	auto request = [&](int cb, float (&o)[16], float (&nm)[16]) {
#pragma unroll
		for (int g = 0; g < 16; ++g) {
			const int idx = ((g & 3) + 8 * (g >> 2)) * RP + 32 * cb;
			o[g] = pw[idx + lofs];
			float nv = nw[idx + lofs];
			for (int k = 1; k < S; ++k) nv += nw[(long)k * slab_stride + idx + lofs];
			nm[g] = nv;
		}
	};
	request(0, oldv[0], numv[0]);
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int cb = 0; cb < NC; ++cb) {
		if (cb + 1 < NC) request(cb + 1, oldv[(cb + 1) & 1], numv[(cb + 1) & 1]);
		__builtin_amdgcn_sched_barrier(0);
		float sq = 0.f;
		const float dcol = scaled ? s_d[32 * cb + l31] : 1.0f;
		float nw16[16];
#pragma unroll
		for (int g = 0; g < 16; ++g) {
			const int idx = ((g & 3) + 8 * (g >> 2)) * RP + 32 * cb;
			const float o = (oldv[cb & 1][g] * dcol) * numv[cb & 1][g] / (acc[cb][g] + eps);
			ow[idx + lofs] = o;
			nw16[g] = o;
			if (HAS_PS) rowdot[g] += o * numv[cb & 1][g];
			sq += o * o;                                        // rows in register order
		}
		if (frag_out != nullptr) {
			// bf16 fragments of the new rows: a fragment is rows 8 h .. 8 h + 7 of one column within a 16-row K-step; register g holds row
			// (g & 3) + 8 (g >> 2) + 4 half.  v_permlane32_swap(x, y) exchanges x's upper half-wave with y's lower one: with x = register 8 kk + g4 and
			// y = register 8 kk + 4 + g4 the lower half-wave ends up with rows g4 and 4 + g4 of K-step kk, the upper one with rows 8 + g4 and 12 + g4.
#pragma unroll
			for (int kk = 0; kk < 2; ++kk) {
				bf16x8 f;
#pragma unroll
				for (int g4 = 0; g4 < 4; ++g4) {
					const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(nw16[8 * kk + g4]), __float_as_uint(nw16[8 * kk + 4 + g4]), false, false);
					f[g4] = (__bf16)__uint_as_float(sw[0]);
					f[4 + g4] = (__bf16)__uint_as_float(sw[1]);
				}
				const long ks = (row0 >> 4) + kk;
				if (ks < frag_KS) frag_out[(ks * NC + cb) * 64 + lane] = f;
			}
		}
		if (sumsq_part != nullptr) {
			sq += __shfl_xor(sq, 32);
			if (half == 0) s_sq[wave][32 * cb + l31] = sq;
		}
		__builtin_amdgcn_sched_barrier(0);
	}
	if (HAS_PS && ps != nullptr) {
		// per-row terms sum_c new(y, c) num(y, c): over the 32 lanes of a half
#pragma unroll
		for (int w = 16; w > 0; w >>= 1)
#pragma unroll
			for (int g = 0; g < 16; ++g) rowdot[g] += __shfl_xor(rowdot[g], w);
		if (l31 == 0) {
#pragma unroll
			for (int g = 0; g < 16; ++g) {
				const long y = row0 + (g & 3) + 8 * (g >> 2) + 4 * half;
				if (y < len_valid) ps[y] = rowdot[g];
			}
		}
	}
	if (sumsq_part != nullptr) {
		// one vector per 32 rows, as panel_update_parts() promises: 4 per workgroup
		__builtin_amdgcn_s_waitcnt(0xc07f);
		__builtin_amdgcn_wave_barrier();
		for (int c = lane; c < RP; c += 64) sumsq_part[((long)blockIdx.x * 4 + wave) * RP + c] = s_sq[wave][c];
	}
}

template <int NC>
static hipError_t launch_rows_mu(const float* P, float* Pout, const float* slabs, int S, long slab_stride, int len_pad, float eps, float* ps, int len_valid,
                                 float* sumsq_part, hipStream_t stream, const void* qx3, const PanelTriExtras* tri) {
	const int blocks = len_pad / 128;
	const float* old_colsq = tri ? tri->old_colsq : nullptr;
	const int old_parts = tri ? tri->old_colsq_parts : 0;
	bf16x8* frag = tri ? reinterpret_cast<bf16x8*>(tri->frag_out) : nullptr;
	const long frag_KS = tri ? tri->frag_KS : 0;
	if (ps != nullptr) hipLaunchKernelGGL((k_panel_update_rows_mu<NC, true, false>), dim3(blocks), dim3(256), 0, stream, P, Pout, slabs, S, slab_stride, eps, ps, len_valid,
	                                      sumsq_part, reinterpret_cast<const bf16x8*>(qx3), old_colsq, old_parts, frag, frag_KS);
	else if (tri != nullptr && tri->old_as_bf16) hipLaunchKernelGGL((k_panel_update_rows_mu<NC, false, true>), dim3(blocks), dim3(256), 0, stream, P, Pout, slabs, S, slab_stride, eps, ps, len_valid,
	                                                               sumsq_part, reinterpret_cast<const bf16x8*>(qx3), old_colsq, old_parts, frag, frag_KS);
	else hipLaunchKernelGGL((k_panel_update_rows_mu<NC, false, false>), dim3(blocks), dim3(256), 0, stream, P, Pout, slabs, S, slab_stride, eps, ps, len_valid,
	                        sumsq_part, reinterpret_cast<const bf16x8*>(qx3), old_colsq, old_parts, frag, frag_KS);
	return hipGetLastError();
}

// long panels (k_panel_update_wide64_mu serves them): the multiplicative update may write its result to another panel
bool panel_update_long_available(int RP, int len_pad) { return (RP == 128 || RP == 256) && len_pad % 64 == 0 && len_pad >= 64 * 512; }

// P_out <- P_in * num / (P_in Q + eps) with Q given as its split image (k_pack_panel_x3 of Q); P_out may be P_in
hipError_t launch_panel_update_long_mu(const float* P_in, float* P_out, const float* slabs, int S, long slab_stride, const void* q_split, int RP, int len_pad,
                                       float eps, float* ps, int len_valid, float* sumsq_part, hipStream_t stream, const PanelTriExtras* tri) {
	if (!panel_update_long_available(RP, len_pad) || q_split == nullptr) return hipErrorInvalidValue;
	// (the 128-row kernel has no numerator transform, no fragment smoothing and, at rank 256, no error terms: callers route those to k_panel_update_wide_f32)
	if (tri != nullptr && (tri->num_transform || tri->frag_a != 1.0f || tri->frag_b != 0.0f || RP != 256 || len_pad % 128 != 0 || ps != nullptr)) return hipErrorInvalidValue;
#ifndef NMFAMD_UPDATE_ROWS
#define NMFAMD_UPDATE_ROWS 1
#endif
	if (NMFAMD_UPDATE_ROWS && len_pad % 128 == 0 && !(ps != nullptr && RP == 256))      // (rank 256 with error terms: that instantiation spills)
		return RP == 128 ? launch_rows_mu<4>(P_in, P_out, slabs, S, slab_stride, len_pad, eps, ps, len_valid, sumsq_part, stream, q_split, tri)
		                 : launch_rows_mu<8>(P_in, P_out, slabs, S, slab_stride, len_pad, eps, ps, len_valid, sumsq_part, stream, q_split, tri);
	if (tri != nullptr) return hipErrorInvalidValue;
	return RP == 128 ? launch_wide64<1>(P_in, P_out, slabs, S, slab_stride, RP, len_pad, eps, ps, len_valid, sumsq_part, stream, q_split)
	                 : launch_wide64<2>(P_in, P_out, slabs, S, slab_stride, RP, len_pad, eps, ps, len_valid, sumsq_part, stream, q_split);
}

bool panel_update_wide_available(int RP) { return RP >= 128 && RP % 128 == 0 && RP <= WIDE_MAX_RP; }

template <int MODE, int NCB, bool FX = false>
static hipError_t launch_wide(float* P, const float* slabs, int S, long slab_stride, const float* Q, int RP, int len_pad,
                              float eps, float* ps, int len_valid, float* sumsq_part, float* num_out, hipStream_t stream, const void* qx3, const PanelTriExtras& tri,
                              const PanelFusedF32& fx = PanelFusedF32()) {
	// two panels + [4][32] error terms + [4][32] row sums; PanelTriExtras::den_transform (rank 256): a third panel, the RP factors, [4][32] partial sums;
	// FX: the RP factors, [4][32] partial sums, two vectors of 32 row sums
	const size_t extra = NCB == 2 ? (size_t)WIDE_YB * (RP + 4) + RP + 128 : 0;
	const size_t fxe = FX ? (size_t)RP + 128 + 64 : 0, fxmax = FX ? (size_t)128 * NCB + 128 + 64 : 0;
	const size_t lds_bytes = sizeof(float) * (2 * (size_t)WIDE_YB * (RP + 4) + 256 + (tri.den_transform ? extra : 0) + fxe);
	const size_t max_bytes = sizeof(float) * (2 * (size_t)WIDE_YB * (128 * NCB + 4) + 256 + extra + fxmax);
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_panel_update_wide_f32<MODE, NCB, FX>), (int)max_bytes, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL((k_panel_update_wide_f32<MODE, NCB, FX>), dim3(len_pad / WIDE_YB), dim3(256), lds_bytes, stream,
	                   P, slabs, S, slab_stride, Q, RP, eps, ps, len_valid, sumsq_part, num_out, reinterpret_cast<const bf16x8*>(qx3), tri, fx);
	return hipGetLastError();
}

hipError_t launch_panel_update_wide_f32(int mode, float* P, const float* slabs, int S, long slab_stride, const float* Q, int RP, int len_pad,
                                        float eps, float* ps, int len_valid, float* sumsq_part, float* num_out, hipStream_t stream, void* q_split, const PanelTriExtras* tri,
                                        const PanelFusedF32* fused) {
	if (!panel_update_wide_available(RP) || (mode != PANEL_MU && mode != PANEL_LS) || len_pad % WIDE_YB != 0) return hipErrorInvalidValue;
	if (tri != nullptr && (RP != 256 || mode != PANEL_MU)) return hipErrorInvalidValue;
	if (fused != nullptr) {
		// the fused fp32 iteration at padded ranks 128 ... 512 (Engine::iterate_fused32w): 32-row kernel, multiplicative update, Q as its split image in q_split
		if (tri != nullptr || mode != PANEL_MU || q_split == nullptr || Q != nullptr || num_out != nullptr) return hipErrorInvalidValue;
		PanelFusedF32 fx = *fused;
		if (!fx.smooth) { fx.off = 0.f; fx.diag = 1.f; }
		const PanelTriExtras none = PanelTriExtras();
		switch (RP / 128) {
		case 1: return launch_wide<PANEL_MU, 1, true>(P, slabs, S, slab_stride, nullptr, RP, len_pad, eps, ps, len_valid, sumsq_part, nullptr, stream, q_split, none, fx);
		case 2: return launch_wide<PANEL_MU, 2, true>(P, slabs, S, slab_stride, nullptr, RP, len_pad, eps, ps, len_valid, sumsq_part, nullptr, stream, q_split, none, fx);
		case 3: return launch_wide<PANEL_MU, 3, true>(P, slabs, S, slab_stride, nullptr, RP, len_pad, eps, ps, len_valid, sumsq_part, nullptr, stream, q_split, none, fx);
		default: return launch_wide<PANEL_MU, 4, true>(P, slabs, S, slab_stride, nullptr, RP, len_pad, eps, ps, len_valid, sumsq_part, nullptr, stream, q_split, none, fx);
		}
	}
	const PanelTriExtras ext = tri ? *tri : PanelTriExtras();
	const void* qx3 = nullptr;
	if (q_split != nullptr) {
		// Q (RP x RP) split into three bf16 planes in fragment order: A(c, k) = Q(k, c) = Q[k * RP + c]
		// (Q == nullptr: q_split holds that image already -- k_smooth_gram writes it next to the matrix)
		if (Q != nullptr) { if (hipError_t e = launch_pack_panel_x3(Q, RP, RP, q_split, RP / 16, stream); e != hipSuccess) return e; }
		qx3 = q_split;
	} else if (Q == nullptr) return hipErrorInvalidValue;
	// long panels, multiplicative update, split operands: 64 rows per workgroup (k_panel_update_wide64_mu)
	// (the long form has no numerator transform and, at rank 256, no error terms: those launches stay with the 32-row kernel)
	if (mode == PANEL_MU && qx3 != nullptr && num_out == nullptr && panel_update_long_available(RP, len_pad) &&
	    !(tri != nullptr && (tri->num_transform || tri->frag_a != 1.0f || tri->frag_b != 0.0f || len_pad % 128 != 0 || ps != nullptr)))
		return launch_panel_update_long_mu(P, P, slabs, S, slab_stride, qx3, RP, len_pad, eps, ps, len_valid, sumsq_part, stream, tri);
#define NMFAMD_WIDE(NCB)                                                                                                                   \
	return mode == PANEL_MU ? launch_wide<PANEL_MU, NCB>(P, slabs, S, slab_stride, Q, RP, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream, qx3, ext) \
	                        : launch_wide<PANEL_LS, NCB>(P, slabs, S, slab_stride, Q, RP, len_pad, eps, ps, len_valid, sumsq_part, num_out, stream, qx3, ext)
	switch (RP / 128) {
	case 1: NMFAMD_WIDE(1);
	case 2: NMFAMD_WIDE(2);
	case 3: NMFAMD_WIDE(3);
	default: NMFAMD_WIDE(4);
	}
#undef NMFAMD_WIDE
}

} // namespace nmfamd
