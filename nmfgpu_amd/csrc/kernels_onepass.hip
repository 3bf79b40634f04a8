// kernels_onepass.hip -- the rank-64 multiplicative update with ONE pass over V per iteration.
//
// Reference sequence (source/nmf/AlgorithmMultiplicativeFrobenius.h:165-248): RN = W^T V (gemm TN, :187-188), RN2 = (W^T W) H
// (:176-183), H .*= RN ./ (RN2 + eps) (:191, KernelMultiplyDivide.cu:29-43), then MR = V H^T with the NEW H (:240-241).
// The element-wise step makes column j of the new H a function of column j of V and the r x r matrix W^T W only, so a
// column panel V(:, J) can be loaded once: reduce W^T V(:, J) over the rows, update H(:, J), and add V(:, J) H(:, J)^T to
// the running (V H^T) while the panel is still on chip.  Only the W update waits for all panels (kernels_mu64.hip, U_W).
//
// Cut (one persistent launch, 256 workgroups of 4 waves, one wave per SIMD, one workgroup per CU):
//   * the 32 workgroups of an XCD form a group that owns a contiguous range of 32-column panels and ALL rows: workgroup
//     `slot` of the group holds a slice of <= 320 rows (20 tiles of 16), wave w of it 80 of them -- for both products;
//   * the wave keeps its rows of the split image of W in registers (120) and its 80 x 64 block of (V H^T) in accumulators (80);
//   * tick t of a workgroup:  A(t): partial W^T V of panel t over its rows (operand V straight from the registers the
//     panel was loaded into), summed over the four waves through LDS and published to the group;  B(t - 3): (V H^T) +=
//     V(rows, panel t-3) Hnew(:, panel t-3)^T with V read back from LDS;  O(t - 1): the workgroup is the OWNER of column
//     `slot` of panel t - 1: it adds the 32 published partials, forms the new column of H and publishes its split form;
//   * a panel therefore stays in LDS for three ticks (3 x 40 KB), which is what hides the two hand-offs of its update.
// Hand-offs stay inside the XCD's L2: plain stores (the vector L1 is write-through), `sc1` loads (they bypass the reader's
// L1), every 8 bytes carry their own tag (tick number), so there is no flag, no fence and no drain -- a reader retries
// until all its tags match.  The group is formed from HW_REG_XCC_ID (the hardware's answer to "which L2 do I use"), not
// from blockIdx: a workgroup takes a ticket of ITS XCD; a launch whose XCDs do not get 32 workgroups each gives up (abort
// flag; every wait is bounded) and the engine falls back to the two-pass iteration.
//
// Arithmetic: both products are the six-term split-operand products of kernels_x3.hip (fp32 accuracy on the bf16 matrix
// pipe); the update itself is fp32 with the reference's formula.  Summation orders differ from the two-pass path (rows are
// cut per workgroup instead of per K slice), results agree to fp32 rounding; everything is deterministic (no atomics on data).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "split3.h"

namespace nmfamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

namespace {

constexpr int SLOTS = ONEPASS_SLOTS, TPW = ONEPASS_TILES_PER_WAVE;
constexpr int LDS_VBUF = 3 * 4 * TPW * 32 * 16 * 4;   // three panels x four wave regions of [tile][column][row] floats
constexpr int LDS_XCH = 4 * 8 * 64 * 16;              // partial W^T V of the four waves
constexpr int LDS_RED = 16 * 64 * 4;                  // owner: partial sums of 16 source pairs
constexpr int LDS_TAIL = 64 * 4 + 64;
constexpr int LDS_TOTAL = LDS_VBUF + LDS_XCH + LDS_RED + LDS_TAIL;
static_assert(LDS_TOTAL <= 163840, "LDS budget");

constexpr u64 GIVE_UP_TICKS = 3000000ull;             // 30 ms of the 100 MHz clock

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ inline u32x4 load_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
	return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);   // aux 16 = sc1: served by L2, not by this CU's L1
}

__device__ inline f32x4 six_terms_16(const bf16x8 (&x)[3], const bf16x8 (&y)[3], f32x4 acc) {
	acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[2], y[0], acc, 0, 0, 0);
	acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], y[2], acc, 0, 0, 0);
	acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], y[1], acc, 0, 0, 0);
	acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], y[0], acc, 0, 0, 0);
	acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], y[1], acc, 0, 0, 0);
	acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], y[0], acc, 0, 0, 0);
	return acc;
}

// hi + mid + lo = v exactly (round-to-nearest cuts, as split3)
__device__ inline void split3_scalar(float v, unsigned& hi, unsigned& mid, unsigned& lo) {
	const __bf16 h = (__bf16)v;
	const float r1 = v - (float)h;
	const __bf16 m = (__bf16)r1;
	const float r2 = r1 - (float)m;
	const __bf16 l = (__bf16)r2;
	hi = __builtin_bit_cast(unsigned short, h); mid = __builtin_bit_cast(unsigned short, m); lo = __builtin_bit_cast(unsigned short, l);
}

} // namespace

__global__ __launch_bounds__(256, 1) void k_mu64_onepass(const OnePassArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	float* const vbuf = reinterpret_cast<float*>(smem);
	f32x4* const xch = reinterpret_cast<f32x4*>(smem + LDS_VBUF);
	float* const s_red = reinterpret_cast<float*>(smem + LDS_VBUF + LDS_XCH);
	float* const s_hnew = s_red + 1024;
	float* const s_ps = s_hnew + 64;
	int* const s_ctl = reinterpret_cast<int*>(s_ps + 4);

	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int l31 = lane & 31, half = lane >> 5;          // 32 x 32 MFMA operand coordinates
	const int cl = lane & 15, grp = lane >> 4;            // 16 x 16 MFMA operand coordinates

	// ---- which group (XCD) and which slot of it -------------------------------------------------------------------------------
	if (tid == 0) {
		unsigned xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		xcc &= 15u;
		int sl = -1;
		if (xcc < (unsigned)ONEPASS_XCDS) sl = (int)(atomicAdd(a.ticket + xcc, 1u) - a.seq * (unsigned)ONEPASS_GROUP);
		s_ctl[0] = sl; s_ctl[1] = (int)xcc;
	}
	__syncthreads();
	const int slot_i = __builtin_amdgcn_readfirstlane(s_ctl[0]), xcd = __builtin_amdgcn_readfirstlane(s_ctl[1]);
	if (slot_i < 0 || slot_i >= ONEPASS_GROUP) {          // this XCD holds more workgroups than a group has members (or an unknown id)
		if (tid == 0) atomicOr(a.abort_flag, 1u);
		return;
	}
	// rows: the tile rows [tr0, tr1) of this slot, TPW per wave; columns: the panels [p0, p0 + T) of this group
	const int tr0 = (int)(((long)a.tile_rows * slot_i) / ONEPASS_GROUP), tr1 = (int)(((long)a.tile_rows * (slot_i + 1)) / ONEPASS_GROUP);
	const int p0 = (a.panels * xcd) / ONEPASS_XCDS, T = (a.panels * (xcd + 1)) / ONEPASS_XCDS - p0;
	const int tmax = (a.panels + ONEPASS_XCDS - 1) / ONEPASS_XCDS + 1;
	const unsigned tag0 = a.seq * (unsigned)tmax;          // tag of tick t: tag0 + t + 1 (never 0 in the first launch, always distinct from a slot's previous content)
	bool gave_up = false;                                  // wave-uniform: stop waiting, run to the end (outputs are discarded by the host)
	const u64 t_start = __builtin_amdgcn_s_memrealtime();

	const __amdgpu_buffer_rsrc_t rs_part = make_rsrc(a.part_scratch, (unsigned)onepass_part_bytes());
	const __amdgpu_buffer_rsrc_t rs_hf = make_rsrc(a.hfrag_scratch, (unsigned)onepass_hfrag_bytes());
	const unsigned part_group = (unsigned)xcd * SLOTS * ONEPASS_GROUP * 256u * 64u;      // byte offsets of this group's areas
	const unsigned hf_group = (unsigned)xcd * SLOTS * 64u * 32u * 8u;

	// ---- resident operands ------------------------------------------------------------------------------------------------------
	// this wave's rows of the split image of W: K-step ks = tile row tr0 + TPW wave + ks (zero beyond the slice: those steps add nothing)
	bf16x8 wf[TPW][2][3];
	int trw[TPW];
	{
		const bf16x8* F = reinterpret_cast<const bf16x8*>(a.Wx3);
#pragma unroll
		for (int ks = 0; ks < TPW; ++ks) {
			const int tr = tr0 + TPW * wave + ks;
			const bool valid = tr < tr1;
			trw[ks] = valid ? tr : tr1 - 1;
#pragma unroll
			for (int nb = 0; nb < 2; ++nb)
#pragma unroll
				for (int pl = 0; pl < 3; ++pl) {
					const int wk = trw[ks] < a.w_ks ? trw[ks] : a.w_ks;      // rows past the image: its closing all-zero step
					bf16x8 v = F[((long)(wk * 2 + nb) * 3 + pl) * 64 + lane];
					if (!valid) { const u32x4 z = {0u, 0u, 0u, 0u}; v = __builtin_bit_cast(bf16x8, z); }
					wf[ks][nb][pl] = v;
				}
		}
	}
	// owner arithmetic: lane (cl, grp) of wave w works on factor row c = 16 w + cl, reduction part k = 16 grp .. 16 grp + 15
	const int oc = 16 * wave + cl;
	float gq[16];
#pragma unroll
	for (int u = 0; u < 4; ++u) {
		const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.G + (long)oc * 64 + 16 * grp + 4 * u);
#pragma unroll
		for (int i = 0; i < 4; ++i) gq[4 * u + i] = g4[i];
	}
	const float sc_c = a.scale[oc];
	float hh[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) hh[i] = 0.f;
	f32x4 accB[TPW][4];
#pragma unroll
	for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
		for (int nt = 0; nt < 4; ++nt) accB[tl][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

	// the two landing slots of the panel stream: lane (l31 = column of the panel, half) takes rows 8 half .. 8 half + 7 of each tile
	f32x4 va[2][TPW][2];
	auto v_addr = [&](int t, int ks) -> const float* { return a.V + (long)trw[ks] * a.tile_stride + ((long)(p0 + t) * 32 + l31) * 16 + 8 * half; };
	auto prefetch = [&](int t, f32x4 (&dst)[TPW][2]) __attribute__((always_inline)) {
#pragma unroll
		for (int ks = 0; ks < TPW; ++ks) {
			const float* p = v_addr(t, ks);
			dst[ks][0] = *reinterpret_cast<const f32x4*>(p);
			dst[ks][1] = *reinterpret_cast<const f32x4*>(p + 4);
		}
	};
	if (T > 0) prefetch(0, va[0]);
	if (T > 1) prefetch(1, va[1]);

	float* const vw = vbuf + wave * (TPW * 32 * 16);      // this wave's region of a panel buffer (+ buffer * 4 * TPW * 512)
	int prev_owner_col = -1;                               // column whose error term / H H^T contribution is still to be booked

	// ---- one tick -----------------------------------------------------------------------------------------------------------------
	auto tick = [&](const int t, f32x4 (&vs)[TPW][2]) __attribute__((always_inline)) {
		const bool do_a = t < T;
		const bool do_b = t >= 3 && t - 3 < T;
		const bool do_o = t >= 1 && t - 1 < T;
		// A(t): D(c, j) = sum_i W(i, c) V(i, j) over this wave's rows
		if (do_a) {
			f32x16 accA[2];
#pragma unroll
			for (int nb = 0; nb < 2; ++nb)
#pragma unroll
				for (int g = 0; g < 16; ++g) accA[nb][g] = 0.f;
			bf16x8 op[2][3];
			{
				float v[8];
#pragma unroll
				for (int j = 0; j < 4; ++j) { v[j] = vs[0][0][j]; v[4 + j] = vs[0][1][j]; }
				split3(v, op[0][0], op[0][1], op[0][2]);
			}
#pragma unroll
			for (int ks = 0; ks < TPW; ++ks) {
				const int cur = ks & 1, nxt = cur ^ 1;
				if (ks + 1 < TPW) {
					float v[8];
#pragma unroll
					for (int j = 0; j < 4; ++j) { v[j] = vs[ks + 1][0][j]; v[4 + j] = vs[ks + 1][1][j]; }
					split3(v, op[nxt][0], op[nxt][1], op[nxt][2]);
				}
#pragma unroll
				for (int nb = 0; nb < 2; ++nb) {
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][nb][2], op[cur][0], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][nb][0], op[cur][2], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][nb][1], op[cur][1], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][nb][1], op[cur][0], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][nb][0], op[cur][1], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][nb][0], op[cur][0], accA[nb], 0, 0, 0);
				}
				if (ks + 1 < TPW) {
#pragma unroll
					for (int g = 0; g < 12; ++g) {
						__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
						__builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			// C/D map of the 32 x 32 MFMA: register 4 q + g of lane (l31, half) is row 8 q + 4 half + g (here c = 32 nb + that), column l31 (= j)
#pragma unroll
			for (int nb = 0; nb < 2; ++nb)
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					f32x4 v;
					v[0] = accA[nb][4 * q + 0]; v[1] = accA[nb][4 * q + 1]; v[2] = accA[nb][4 * q + 2]; v[3] = accA[nb][4 * q + 3];
					xch[(wave * 8 + nb * 4 + q) * 64 + lane] = v;
				}
		}
		__syncthreads();                                                                   // BAR_a
		// book the column this workgroup finished as owner in the previous tick: error term, H H^T
		if (prev_owner_col >= 0) {
			if (a.compute_error && tid == 0 && prev_owner_col < a.n) a.ps[prev_owner_col] = ((s_ps[0] + s_ps[1]) + s_ps[2]) + s_ps[3];
			const float hc = s_hnew[oc];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const f32x4 hk = *reinterpret_cast<const f32x4*>(s_hnew + 16 * grp + 4 * u);
#pragma unroll
				for (int i = 0; i < 4; ++i) hh[4 * u + i] = fmaf(hc, hk[i], hh[4 * u + i]);
			}
			prev_owner_col = -1;
		}
		// sum over the four waves (wave order) and publish: thread (w', lane) owns (nb, q) = 2 w', 2 w' + 1 of column l31, rows 4 half + g
		if (do_a) {
			const unsigned tg = tag0 + (unsigned)t + 1u;
			u32x4* dst = reinterpret_cast<u32x4*>(a.part_scratch) + ((long)((xcd * SLOTS + (t % SLOTS)) * ONEPASS_GROUP + slot_i) * 256 + tid) * 4;
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const int nbq = 2 * (tid >> 6) + u;
				f32x4 s = xch[(0 * 8 + nbq) * 64 + lane];
#pragma unroll
				for (int w = 1; w < 4; ++w) s += xch[(w * 8 + nbq) * 64 + lane];
				u32x4 g0, g1;
				g0[0] = __float_as_uint(s[0]); g0[1] = tg; g0[2] = __float_as_uint(s[1]); g0[3] = tg;
				g1[0] = __float_as_uint(s[2]); g1[1] = tg; g1[2] = __float_as_uint(s[3]); g1[3] = tg;
				dst[2 * u] = g0; dst[2 * u + 1] = g1;
			}
		}
		// B(t - 3): D(i, c) += sum_j V(i, j) Hnew(c, j) over the panel's 32 columns, K order j = 4 k + grp
		if (do_b) {
			const int tb = t - 3;
			const unsigned tg16 = (tag0 + (unsigned)tb + 1u) & 0xffffu;
			const unsigned hbase = hf_group + (unsigned)(tb % SLOTS) * (64u * 32u * 8u) + (unsigned)(cl * 4 + grp) * 64u;
			bf16x8 hf[4][3];
			const u64 w0 = __builtin_amdgcn_s_memrealtime();
			for (;;) {
				unsigned bad = 0;
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) {
					u32x4 d[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) d[q] = load_sc1(rs_hf, hbase + (unsigned)nt * (16u * 4u * 64u) + 16u * q);
					u32x4 o0, o1, o2;
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						// d[q] = two granules {p0 | p1 << 16, p2 | tag << 16} of k = 2 q, 2 q + 1
						o0[q] = (d[q][0] & 0xffffu) | (d[q][2] << 16);
						o1[q] = (d[q][0] >> 16) | (d[q][2] & 0xffff0000u);
						o2[q] = (d[q][1] & 0xffffu) | (d[q][3] << 16);
						bad |= ((d[q][1] >> 16) ^ tg16) | ((d[q][3] >> 16) ^ tg16);
					}
					hf[nt][0] = __builtin_bit_cast(bf16x8, o0); hf[nt][1] = __builtin_bit_cast(bf16x8, o1); hf[nt][2] = __builtin_bit_cast(bf16x8, o2);
				}
				if (__all(bad == 0) || gave_up) break;
				if (__builtin_amdgcn_s_memrealtime() - w0 > GIVE_UP_TICKS) { gave_up = true; if (lane == 0) atomicOr(a.abort_flag, 2u); break; }
				__builtin_amdgcn_s_sleep(8);
			}
			const float* vr = vw + (tb % 3) * (4 * TPW * 512) + grp * 16 + cl;                // + tile * 512 + k * 64
			bf16x8 op[2][3];
			float raw[2][8];
#pragma unroll
			for (int k = 0; k < 8; ++k) raw[0][k] = vr[k * 64];
#pragma unroll
			for (int k = 0; k < 8; ++k) raw[1][k] = vr[512 + k * 64];
			split3(raw[0], op[0][0], op[0][1], op[0][2]);
#pragma unroll
			for (int tl = 0; tl < TPW; ++tl) {
				const int cur = tl & 1, nxt = cur ^ 1;
				if (tl + 1 < TPW) split3(raw[nxt], op[nxt][0], op[nxt][1], op[nxt][2]);
				if (tl + 2 < TPW) {
#pragma unroll
					for (int k = 0; k < 8; ++k) raw[cur][k] = vr[(tl + 2) * 512 + k * 64];
				}
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) accB[tl][nt] = six_terms_16(op[cur], hf[nt], accB[tl][nt]);
				if (tl + 1 < TPW) {
#pragma unroll
					for (int g = 0; g < 24; ++g) {
						__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
						__builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
						if (g < 8 && tl + 2 < TPW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		// owner loads of O(t - 1), issued before the panel stream's next loads so that they can be waited for on their own
		const int to = t - 1;
		const int jc = do_o ? (p0 + to) * 32 + slot_i : 0;                                       // the column this workgroup owns in panel t - 1
		u32x4 od[4];
		f32x4 hold[4];
		float hcur = 0.f;
		const unsigned obase = part_group + (unsigned)((to & (SLOTS - 1)) * ONEPASS_GROUP + (tid >> 3)) * (256u * 64u) +
		                       (unsigned)(((tid & 7) >> 1) * 64 + slot_i + 32 * (tid & 1)) * 64u;
		if (do_o) {
#pragma unroll
			for (int q = 0; q < 4; ++q) od[q] = load_sc1(rs_part, obase + 16u * q);
#pragma unroll
			for (int u = 0; u < 4; ++u) hold[u] = *reinterpret_cast<const f32x4*>(a.H + (long)jc * 64 + 16 * grp + 4 * u);
			hcur = a.H[(long)jc * 64 + oc];
		}
		// the panel leaves its landing registers for LDS (the buffer B(t - 3) has just finished with); the slot takes panel t + 2
		if (do_a) {
			float* vd = vw + (t % 3) * (4 * TPW * 512) + l31 * 16 + 8 * half;
#pragma unroll
			for (int ks = 0; ks < TPW; ++ks) {
				*reinterpret_cast<f32x4*>(vd + ks * 512) = vs[ks][0];
				*reinterpret_cast<f32x4*>(vd + ks * 512 + 4) = vs[ks][1];
			}
			if (t + 2 < T) prefetch(t + 2, vs);
		}
		if (do_o) {
			const unsigned tg = tag0 + (unsigned)to + 1u;
			const u64 w0 = __builtin_amdgcn_s_memrealtime();
			for (;;) {
				unsigned bad = 0;
#pragma unroll
				for (int q = 0; q < 4; ++q) bad |= (od[q][1] ^ tg) | (od[q][3] ^ tg);
				if (__all(bad == 0) || gave_up) break;
				if (__builtin_amdgcn_s_memrealtime() - w0 > GIVE_UP_TICKS) { gave_up = true; if (lane == 0) atomicOr(a.abort_flag, 4u); break; }
				__builtin_amdgcn_s_sleep(8);
#pragma unroll
				for (int q = 0; q < 4; ++q) od[q] = load_sc1(rs_part, obase + 16u * q);
			}
			// (__uint_as_float, not __builtin_bit_cast(float, od[q][i]): hipcc 7.2 folds the bit cast of a vector ELEMENT of a
			//  buffer load's result to element 0)
			// thread (source s' = tid / 8, piece e = tid % 8 = (w', h)) holds D(c, slot column) of source s' for
			// c = 32 (w' / 2) + 16 (w' % 2) + 8 u + 4 h + g; sources 2 i and 2 i + 1 sit eight lanes apart
			f32x4 v0, v1;
			v0[0] = __uint_as_float(od[0][0]); v0[1] = __uint_as_float(od[0][2]);
			v0[2] = __uint_as_float(od[1][0]); v0[3] = __uint_as_float(od[1][2]);
			v1[0] = __uint_as_float(od[2][0]); v1[1] = __uint_as_float(od[2][2]);
			v1[2] = __uint_as_float(od[3][0]); v1[3] = __uint_as_float(od[3][2]);
#pragma unroll
			for (int i = 0; i < 4; ++i) { v0[i] += __shfl_xor(v0[i], 8); v1[i] += __shfl_xor(v1[i], 8); }
			if ((lane & 8) == 0) {
				const int e = tid & 7, wq = e >> 1, h = e & 1;
				const int cb = 32 * (wq >> 1) + 16 * (wq & 1) + 4 * h;
				float* r = s_red + (tid >> 4) * 64 + cb;
				*reinterpret_cast<f32x4*>(r) = v0;
				*reinterpret_cast<f32x4*>(r + 8) = v1;
			}
		}
		__syncthreads();                                                                   // BAR_b
		if (do_o) {
			// lane (cl, grp): source pairs 4 grp .. 4 grp + 3, then the four groups in order
			float sum = s_red[(4 * grp + 0) * 64 + oc];
#pragma unroll
			for (int i = 1; i < 4; ++i) sum += s_red[(4 * grp + i) * 64 + oc];
			float den = 0.f;
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int i = 0; i < 4; ++i) den = fmaf(gq[4 * u + i], hold[u][i], den);
			// (0 + 1) + (2 + 3) in every lane
			sum += __shfl_xor(sum, 16); den += __shfl_xor(den, 16);
			sum += __shfl_xor(sum, 32); den += __shfl_xor(den, 32);
			const float num = sum * sc_c;                                              // the pending column scale of W (kernels_mu64.hip)
			const float hn = hcur * num / (den + a.eps);                               // KernelMultiplyDivide.cu:39-42
			float psum = hn * num;                                                     // KernelTraceMultiplication.cu:43-80, per-column term
			psum += __shfl_xor(psum, 1); psum += __shfl_xor(psum, 2); psum += __shfl_xor(psum, 4); psum += __shfl_xor(psum, 8);
			if (grp == 0) {
				a.H[(long)jc * 64 + oc] = hn;
				s_hnew[oc] = hn;
				unsigned p0b, p1b, p2b;
				split3_scalar(hn, p0b, p1b, p2b);
				const u64 gr = (u64)p0b | ((u64)p1b << 16) | ((u64)p2b << 32) | ((u64)((tag0 + (unsigned)to + 1u) & 0xffffu) << 48);
				u64* hd = reinterpret_cast<u64*>(a.hfrag_scratch) + ((long)(xcd * SLOTS + (to % SLOTS)) * 64 + oc) * 32 + (slot_i & 3) * 8 + (slot_i >> 2);
				*hd = gr;
				if (cl == 0) s_ps[wave] = psum;
			}
			prev_owner_col = jc;
		}
	};

	for (int t = 0; t < T + 3; t += 2) {
		tick(t, va[0]);
		tick(t + 1, va[1]);
	}
	// the last owner column (booked after the barrier of a tick that does not come): one more barrier
	__syncthreads();
	if (prev_owner_col >= 0) {
		if (a.compute_error && tid == 0 && prev_owner_col < a.n) a.ps[prev_owner_col] = ((s_ps[0] + s_ps[1]) + s_ps[2]) + s_ps[3];
		const float hc = s_hnew[oc];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const f32x4 hk = *reinterpret_cast<const f32x4*>(s_hnew + 16 * grp + 4 * u);
#pragma unroll
			for (int i = 0; i < 4; ++i) hh[4 * u + i] = fmaf(hc, hk[i], hh[4 * u + i]);
		}
	}

	// ---- results ------------------------------------------------------------------------------------------------------------------
	// (V H^T)^T partial of this group: C/D map of the 16 x 16 MFMA: register g of lane (cl, grp) is row 4 grp + g (tile row i), column cl (c = 16 nt + cl)
	float* slab = a.slabs + (long)xcd * a.slab_stride;
#pragma unroll
	for (int tl = 0; tl < TPW; ++tl) {
		const int tr = tr0 + TPW * wave + tl;
		if (tr < tr1) {
#pragma unroll
			for (int nt = 0; nt < 4; ++nt)
#pragma unroll
				for (int g = 0; g < 4; ++g) slab[((long)tr * 16 + 4 * grp + g) * 64 + 16 * nt + cl] = accB[tl][nt][g];
		}
	}
	float* hp = a.hh_part + (long)(xcd * ONEPASS_GROUP + slot_i) * 4096 + (long)oc * 64 + 16 * grp;
#pragma unroll
	for (int u = 0; u < 4; ++u) {
		f32x4 v;
		v[0] = hh[4 * u]; v[1] = hh[4 * u + 1]; v[2] = hh[4 * u + 2]; v[3] = hh[4 * u + 3];
		*reinterpret_cast<f32x4*>(hp + 4 * u) = v;
	}
	(void)t_start;
}

bool onepass_available(long mpad, int num_cus) {
	return num_cus == ONEPASS_XCDS * ONEPASS_GROUP && mpad % 16 == 0 && mpad / 16 <= (long)ONEPASS_GROUP * 4 * TPW;
}

hipError_t launch_mu64_onepass(const OnePassArgs& a, hipStream_t stream) {
	if (a.tile_rows <= 0 || a.tile_rows > ONEPASS_GROUP * 4 * TPW || a.panels <= 0) return hipErrorInvalidValue;
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_mu64_onepass), LDS_TOTAL, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL(k_mu64_onepass, dim3(ONEPASS_XCDS * ONEPASS_GROUP), dim3(256), LDS_TOTAL, stream, a);
	return hipGetLastError();
}

} // namespace nmfamd
