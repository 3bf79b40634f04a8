// kernels_onepass.hip -- the rank-64 multiplicative update with ONE pass over V per iteration (one fetch from HBM).
//
// Reference sequence (source/nmf/AlgorithmMultiplicativeFrobenius.h:165-248): RN = W^T V (gemm TN, :187-188), RN2 = (W^T W) H
// (:176-183), H .*= RN ./ (RN2 + eps) (:191, KernelMultiplyDivide.cu:29-43), then MR = V H^T with the NEW H (:240-241).
// The element-wise step makes column j of the new H a function of column j of V and the r x r matrix W^T W only, so a
// column panel V(:, J) can be taken once: reduce W^T V(:, J) over the rows, update H(:, J), and add V(:, J) H(:, J)^T to
// the running (V H^T) while the panel is still close by.  Only the W update waits for all panels (kernels_mu64.hip, U_W).
//
// Cut (one persistent launch, 256 workgroups of 4 waves, one wave per SIMD, one workgroup per CU):
//   * the 32 workgroups of an XCD form a group that owns a contiguous range of 32-column panels and ALL rows: workgroup
//     `slot` of the group holds a slice of <= 320 rows (20 tiles of 16), wave w of it 80 of them -- for both products;
//   * the wave keeps its rows of the split image of W in LDS (30 KB per wave, read in place as MFMA operands) and its
//     80 x 64 block of (V H^T) in accumulators (80 registers);
//   * tick t of a workgroup:  A(t): partial W^T V of panel t over its rows (operand V from the registers the panel was
//     loaded into, once, from HBM), summed over the four waves through LDS and published to the group;  O(t - 1): the
//     workgroup is the OWNER of column `slot` of panel t - 1: it adds the 32 published partials, forms the new column of H
//     and publishes its split form;  B(t - LAG): (V H^T) += V(rows, panel) Hnew(:, panel)^T -- V read a second time, now
//     from the XCD's L2 / the memory-side cache it passed through LAG ticks ago (fully coalesced 256-byte rows).
// Hand-offs stay inside the XCD's L2: plain stores (the vector L1 is write-through), `sc1` loads (they bypass the reader's
// L1), every 8 bytes carry their own tag (tick number), so there is no flag, no fence and no drain -- a reader retries
// until all its tags match.  The group is formed from HW_REG_XCC_ID (the hardware's answer to "which L2 do I use"), not
// from blockIdx: a workgroup takes a ticket of ITS XCD; a launch whose XCDs do not get 32 workgroups each gives up (abort
// flag; every wait is bounded) and the engine falls back to the two-pass iteration.
// vmcnt completes in order: the far loads (next panel) are issued first in a tick, so that the waits for the near loads
// (hand-offs, second read of V) issued later never have to outwait a younger far load.
//
// Arithmetic: both products are the six-term split-operand products of kernels_x3.hip (fp32 accuracy on the bf16 matrix
// pipe); the update itself is fp32 with the reference's formula.  Summation orders differ from the two-pass path (rows are
// cut per workgroup instead of per K slice), results agree to fp32 rounding; everything is deterministic (no atomics on data).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.h"
#include "split3.h"

namespace nmfamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

namespace {

constexpr int SLOTS = ONEPASS_SLOTS, TPW = ONEPASS_TILES_PER_WAVE;
constexpr int LAG = 3;                                // ticks between a panel's first use (A) and its second (B): two hand-offs fit in between
constexpr int LDS_WF = 4 * TPW * 2 * 3 * 64 * 16;     // the four waves' rows of the split image of W, fragment order
constexpr int LDS_XCH = 4 * 8 * 64 * 16;              // partial W^T V of the four waves
constexpr int LDS_TAIL = 64 * 4 + 64;
constexpr int LDS_TOTAL = LDS_WF + LDS_XCH + LDS_TAIL;
static_assert(LDS_TOTAL <= 163840, "LDS budget");
static_assert(LAG >= 2 && LAG < SLOTS, "a slot must not be rewritten before its readers are done");

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

// Workgroup barrier for LDS traffic only: __syncthreads() also drains vmcnt (every load and STORE in flight: the next panel from HBM,
// the hand-off loads, the publishing stores) -- thousands of cycles per tick here.  The LDS operations of this wave are complete
// (lgkmcnt(0)) before it arrives; global memory is not ordered by this barrier (the hand-offs carry their own tags).
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// DIAG (diagnostic builds, tools/onepass_stamps.py): per-wave shader-cycle sums of the segments of a tick, see the stamps' layout at the end
template <bool DIAG>
__global__ __launch_bounds__(256, 1) void k_mu64_onepass(const OnePassArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	bf16x8* const wfl = reinterpret_cast<bf16x8*>(smem);
	f32x4* const xch = reinterpret_cast<f32x4*>(smem + LDS_WF);
	float* const s_hnew = reinterpret_cast<float*>(smem + LDS_WF + LDS_XCH);
	float* const s_ps = s_hnew + 64;
	int* const s_ctl = reinterpret_cast<int*>(s_ps + 4);

	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int grp = lane >> 4;                            // K group of the 16 x 16 MFMA operands

	// ---- which group (XCD) and which slot of it -------------------------------------------------------------------------------
	if (tid == 0) {
		unsigned xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		xcc &= 15u;
		int sl = -1;
		if (xcc < (unsigned)ONEPASS_XCDS) sl = (int)(atomicAdd(a.ticket + xcc, 1u) - a.seq * (unsigned)ONEPASS_GROUP);
		s_ctl[0] = sl; s_ctl[1] = (int)xcc;
	}
	if (tid < 64) s_hnew[tid] = 0.f;
	if (tid < 4) s_ps[tid] = 0.f;
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
	u64 seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c_last = 0, c_entry = 0, c_loop0 = 0, c_loop1 = 0;
	unsigned retries_f = 0, retries_o = 0;
	auto stamp = [&](int i) __attribute__((always_inline)) {
		if (DIAG) {
			__builtin_amdgcn_sched_barrier(0);
			const u64 c = __builtin_amdgcn_s_memtime();
			seg[i] += c - c_last; c_last = c;
			__builtin_amdgcn_sched_barrier(0);
		}
	};
	if (DIAG) c_entry = __builtin_amdgcn_s_memtime();

	const __amdgpu_buffer_rsrc_t rs_part = make_rsrc(a.part_scratch, (unsigned)onepass_part_bytes());
	const __amdgpu_buffer_rsrc_t rs_hf = make_rsrc(a.hfrag_scratch, (unsigned)onepass_hfrag_bytes());
	const unsigned part_group = (unsigned)xcd * SLOTS * ONEPASS_GROUP * 256u * 64u;      // byte offsets of this group's areas
	const unsigned hf_group = (unsigned)xcd * SLOTS * 64u * 32u * 8u;

	// ---- resident operands ------------------------------------------------------------------------------------------------------
	// this wave's rows of the split image of W, into its own part of LDS: K-step ks = tile row tr0 + TPW wave + ks (zero beyond the
	// slice: those steps add nothing); fragment (ks, nb, plane) of lane l at wl[((ks * 2 + nb) * 3 + plane) * 64 + l]
	bf16x8* const wl = wfl + wave * (TPW * 2 * 3 * 64) + lane;
	int trw[TPW];
	{
		const bf16x8* F = reinterpret_cast<const bf16x8*>(a.Wx3);
#pragma unroll
		for (int ks = 0; ks < TPW; ++ks) {
			const int tr = tr0 + TPW * wave + ks;
			const bool valid = tr < tr1;
			trw[ks] = valid ? tr : tr1 - 1;
			const int wk = trw[ks] < a.w_ks ? trw[ks] : a.w_ks;          // rows past the image: its closing all-zero step
#pragma unroll
			for (int f = 0; f < 6; ++f) {
				bf16x8 v = F[((long)wk * 6 + f) * 64 + lane];
				if (!valid) { const u32x4 z = {0u, 0u, 0u, 0u}; v = __builtin_bit_cast(bf16x8, z); }
				wl[(ks * 6 + f) * 64] = v;
			}
		}
	}
	// owner arithmetic: lane l of wave w works on factor row c = 16 w + 2 (l & 7) + ((l >> 3) & 1), reduction part k = 16 (l >> 4) .. + 15
	// (the row follows from how the owner's loads are dealt out, see O below)
	const int oc = 16 * wave + 2 * (lane & 7) + ((lane >> 3) & 1);
	const float* const grow = a.G + (long)oc * 64 + 16 * grp;      // this lane's part of row c of W^T W (read again every tick: L1 / L2)
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
	// (wave-uniform base + a 32-bit lane offset: the loads take the base in scalar registers)
	f32x4 va[2][TPW][2];
	const int v_lane = (lane & 31) * 16 + 8 * (lane >> 5);
	auto prefetch = [&](int t, f32x4 (&dst)[TPW][2]) __attribute__((always_inline)) {
#pragma unroll
		for (int ks = 0; ks < TPW; ++ks) {
			const float* p = a.V + ((long)trw[ks] * a.tile_stride + (long)(p0 + t) * (32 * 16));
			dst[ks][0] = *reinterpret_cast<const f32x4*>(p + v_lane);
			dst[ks][1] = *reinterpret_cast<const f32x4*>(p + v_lane + 4);
		}
	};
	if (T > 0) prefetch(0, va[0]);

	int prev_owner_col = -1;                               // column whose error term / H H^T contribution is still to be booked
	bf16x8 opn[3];                                         // split form of K-step 0 of the NEXT panel (prepared while the matrix pipe runs B)
	auto split_pair = [&](const f32x4& lo4, const f32x4& hi4, bf16x8 (&o)[3]) __attribute__((always_inline)) {
		float v[8];
#pragma unroll
		for (int j = 0; j < 4; ++j) { v[j] = lo4[j]; v[4 + j] = hi4[j]; }
		split3(v, o[0], o[1], o[2]);
	};
	// x + (x of the lane N places on, cyclically, in its row of 16 lanes) -- a DPP rotation, no LDS
	auto add_ror = [&](float x, auto ctrl_c) __attribute__((always_inline)) -> float {
		constexpr int CTRL = decltype(ctrl_c)::value;
		return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
	};
	typedef std::integral_constant<int, 0x128> ROR8;
	typedef std::integral_constant<int, 0x124> ROR4;
	typedef std::integral_constant<int, 0x122> ROR2;
	typedef std::integral_constant<int, 0x121> ROR1;
	if (T > 0) split_pair(va[0][0][0], va[0][0][1], opn);
	__builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): this wave's fragments of W are in LDS (nobody else reads them)

	// ---- one tick -----------------------------------------------------------------------------------------------------------------
	// vs: landing slot of panel t, vn: of panel t + 1 (the split of its first K-step rides in B's last tile).
	auto tick = [&](const int t, f32x4 (&vs)[TPW][2], f32x4 (&vn)[TPW][2]) __attribute__((always_inline)) {
		const bool do_a = t < T;
		const bool do_b = t >= LAG && t - LAG < T;
		const bool do_o = t >= 1 && t - 1 < T;
		const int tb = t - LAG, to = t - 1;
		stamp(7);
		// the next panel is requested first: the only loads from far away (memory-side cache / HBM) are then the oldest ones in flight
		if (t + 1 < T) prefetch(t + 1, vn);
		// ---- A(t): D(c, j) = sum_i W(i, c) V(i, j) over this wave's rows ---------------------------------------------------------
		if (do_a) {
			f32x16 accA[2];
#pragma unroll
			for (int nb = 0; nb < 2; ++nb)
#pragma unroll
				for (int g = 0; g < 16; ++g) accA[nb][g] = 0.f;
			bf16x8 op[2][3], wf[2][2][3];
#pragma unroll
			for (int pl = 0; pl < 3; ++pl) op[0][pl] = opn[pl];
#pragma unroll
			for (int f = 0; f < 6; ++f) wf[0][f / 3][f % 3] = wl[f * 64];
#pragma unroll
			for (int ks = 0; ks < TPW; ++ks) {
				const int cur = ks & 1, nxt = cur ^ 1;
				if (ks + 1 < TPW) {
#pragma unroll
					for (int f = 0; f < 6; ++f) wf[nxt][f / 3][f % 3] = wl[((ks + 1) * 6 + f) * 64];
					split_pair(vs[ks + 1][0], vs[ks + 1][1], op[nxt]);
				}
#pragma unroll
				for (int nb = 0; nb < 2; ++nb) {
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][nb][2], op[cur][0], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][nb][0], op[cur][2], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][nb][1], op[cur][1], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][nb][1], op[cur][0], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][nb][0], op[cur][1], accA[nb], 0, 0, 0);
					accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][nb][0], op[cur][0], accA[nb], 0, 0, 0);
				}
				if (ks + 1 < TPW) {
#pragma unroll
					for (int g = 0; g < 12; ++g) {
						__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
						__builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
						if (g < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			// C/D map of the 32 x 32 MFMA: register 4 q + g of lane (l31, half) is row 8 q + 4 half + g (here c = 32 nb + that), column l31 (= j).
			// Exchange image: float4 (wave, lane, nbq = 4 nb + q) at (wave * 64 + lane) * 8 + (nbq ^ (lane & 7)): a lane's eight
			// chunks are 128 contiguous bytes, the XOR spreads lanes over the banks for the writer and for the transposing reader
#pragma unroll
			for (int nb = 0; nb < 2; ++nb)
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					f32x4 v;
					v[0] = accA[nb][4 * q + 0]; v[1] = accA[nb][4 * q + 1]; v[2] = accA[nb][4 * q + 2]; v[3] = accA[nb][4 * q + 3];
					xch[(wave * 64 + lane) * 8 + ((nb * 4 + q) ^ (lane & 7))] = v;
				}
		}
		stamp(0);
		lds_barrier();                                                                     // BAR_a: the exchange image is complete
		stamp(1);
		// the split columns of H that B multiplies with were published a tick ago: requested now, looked at after the publishing step.
		// image of a slot: 16-byte piece ((nt * 4 + q) * 64 + lane) = the two granules k = 2 q, 2 q + 1 of (c = 16 nt + (lane & 15), j = 4 k + grp)
		const unsigned hbase = hf_group + (unsigned)((tb + SLOTS) % SLOTS) * (64u * 32u * 8u) + (unsigned)lane * 16u;
		u32x4 hraw[4][4];
		if (do_b) {
#pragma unroll
			for (int nt = 0; nt < 4; ++nt)
#pragma unroll
				for (int q = 0; q < 4; ++q) hraw[nt][q] = load_sc1(rs_hf, hbase + (unsigned)(nt * 4 + q) * 1024u);
		}
		// book the column this workgroup finished as owner in the previous tick: error term, H H^T
		{
			const bool have = prev_owner_col >= 0;
			if (have && a.compute_error && tid == 0 && prev_owner_col < a.n) a.ps[prev_owner_col] = ((s_ps[0] + s_ps[1]) + s_ps[2]) + s_ps[3];
			const float hc = have ? s_hnew[oc] : 0.f;
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const f32x4 hk = *reinterpret_cast<const f32x4*>(s_hnew + 16 * grp + 4 * u);
#pragma unroll
				for (int i = 0; i < 4; ++i) hh[4 * u + i] = fmaf(hc, hk[i], hh[4 * u + i]);
			}
			prev_owner_col = -1;
		}
		// sum over the four waves (wave order) and publish: thread (j = tid / 8, nbq = tid % 8) takes c = 8 nbq .. 8 nbq + 7 of column j,
		// i.e. the chunk nbq of lanes (j, half 0) and (j, half 1); its 64 bytes of granules are bytes [64 tid, 64 tid + 64) of the
		// workgroup's slot: slot image = [column j][c] granules {value, tag}
		if (do_a) {
			const unsigned tg = tag0 + (unsigned)t + 1u;
			u32x4* dst = reinterpret_cast<u32x4*>(a.part_scratch) + ((long)((xcd * SLOTS + (t % SLOTS)) * ONEPASS_GROUP + slot_i) * 256 + tid) * 4;
			const int pj = tid >> 3, pq = tid & 7;
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const int ln = pj + 32 * h;
				f32x4 s = xch[(0 * 64 + ln) * 8 + (pq ^ (ln & 7))];
#pragma unroll
				for (int w = 1; w < 4; ++w) s += xch[(w * 64 + ln) * 8 + (pq ^ (ln & 7))];
				u32x4 g0, g1;
				g0[0] = __float_as_uint(s[0]); g0[1] = tg; g0[2] = __float_as_uint(s[1]); g0[3] = tg;
				g1[0] = __float_as_uint(s[2]); g1[1] = tg; g1[2] = __float_as_uint(s[3]); g1[3] = tg;
				dst[2 * h] = g0; dst[2 * h + 1] = g1;
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		// owner of column `slot` of panel t - 1: its old values and the denominator (W^T W) H do not wait for anybody
		const int jc = do_o ? (p0 + to) * 32 + slot_i : 0;
		f32x4 hold[4], gq[4];
		float hcur = 0.f, den = 0.f;
		if (do_o) {
#pragma unroll
			for (int u = 0; u < 4; ++u) { hold[u] = *reinterpret_cast<const f32x4*>(a.H + (long)jc * 64 + 16 * grp + 4 * u); gq[u] = *reinterpret_cast<const f32x4*>(grow + 4 * u); }
			hcur = a.H[(long)jc * 64 + oc];
		}
		stamp(2);
		lds_barrier();                                                                     // BAR_b: every wave is done with the exchange image and the booking words
		// second read of V(rows, panel t - LAG): tile tl is 8 rows of 256 bytes, lane l of load k takes row i = l & 15 at column j = 4 k + (l >> 4)
		float rawb[3][8];
		const float* const vb0 = a.V + ((long)(p0 + tb) * (32 * 16) + lane);
		auto load_tile = [&](int tl, float (&dst)[8]) __attribute__((always_inline)) {
			const float* p = vb0 + (long)trw[tl] * a.tile_stride;
#pragma unroll
			for (int k = 0; k < 8; ++k) dst[k] = p[64 * k];
		};
		if (do_b) { load_tile(0, rawb[0]); load_tile(1, rawb[1]); load_tile(2, rawb[2]); }
		if (do_o) {
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int i = 0; i < 4; ++i) den = fmaf(gq[u][i], hold[u][i], den);
			den += __shfl_xor(den, 16);
			den += __shfl_xor(den, 32);                                                // (0 + 1) + (2 + 3) in every lane
		}
		__builtin_amdgcn_sched_barrier(0);
		// B's operand H: every granule must carry its tick's tag
		bf16x8 hf[4][3];
		if (do_b) {
			const unsigned tg16 = (tag0 + (unsigned)tb + 1u) & 0xffffu;
			unsigned bad = 0;
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) {
				u32x4 o0, o1, o2;
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					// two granules {p0 | p1 << 16, p2 | tag << 16} of k = 2 q, 2 q + 1
					const u32x4 d = hraw[nt][q];
					o0[q] = (d[0] & 0xffffu) | (d[2] << 16);
					o1[q] = (d[0] >> 16) | (d[2] & 0xffff0000u);
					o2[q] = (d[1] & 0xffffu) | (d[3] << 16);
					bad |= ((d[1] >> 16) ^ tg16) | ((d[3] >> 16) ^ tg16);
				}
				hf[nt][0] = __builtin_bit_cast(bf16x8, o0); hf[nt][1] = __builtin_bit_cast(bf16x8, o1); hf[nt][2] = __builtin_bit_cast(bf16x8, o2);
			}
			if (!__all(bad == 0) && !gave_up) {
				// (rare) an owner is late: read again until every tag matches
				const u64 w0 = __builtin_amdgcn_s_memrealtime();
				for (;;) {
					__builtin_amdgcn_s_sleep(4);
					if (DIAG) ++retries_f;
					bad = 0;
#pragma unroll
					for (int nt = 0; nt < 4; ++nt) {
						u32x4 d[4];
#pragma unroll
						for (int q = 0; q < 4; ++q) d[q] = load_sc1(rs_hf, hbase + (unsigned)(nt * 4 + q) * 1024u);
						u32x4 o0, o1, o2;
#pragma unroll
						for (int q = 0; q < 4; ++q) {
							o0[q] = (d[q][0] & 0xffffu) | (d[q][2] << 16);
							o1[q] = (d[q][0] >> 16) | (d[q][2] & 0xffff0000u);
							o2[q] = (d[q][1] & 0xffffu) | (d[q][3] << 16);
							bad |= ((d[q][1] >> 16) ^ tg16) | ((d[q][3] >> 16) ^ tg16);
						}
						hf[nt][0] = __builtin_bit_cast(bf16x8, o0); hf[nt][1] = __builtin_bit_cast(bf16x8, o1); hf[nt][2] = __builtin_bit_cast(bf16x8, o2);
					}
					if (__all(bad == 0)) break;
					if (__builtin_amdgcn_s_memrealtime() - w0 > GIVE_UP_TICKS) { gave_up = true; if (lane == 0) atomicOr(a.abort_flag, 2u); break; }
				}
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		stamp(3);
		// O(t - 1), this wave's sixteen factor rows c = 16 w .. 16 w + 15 of the owned column: the 32 sources' partial sums are requested
		// before B and looked at after it.  A source's 128 bytes (16 granules) are 8 pieces of 16 bytes (c = 16 w + 2 p, + 1); load q of
		// lane l takes piece p = l & 7 of source 8 q + (l >> 3): 128 contiguous bytes per source and instruction
		u32x4 od[4];
		const unsigned obase = part_group + (unsigned)(((to + SLOTS) % SLOTS) * ONEPASS_GROUP + (lane >> 3)) * (256u * 64u) + (unsigned)slot_i * 512u +
		                       (unsigned)wave * 128u + (unsigned)(lane & 7) * 16u;
		if (do_o) {
#pragma unroll
			for (int q = 0; q < 4; ++q) od[q] = load_sc1(rs_part, obase + (unsigned)q * (8u * 256u * 64u));
		}
		__builtin_amdgcn_sched_barrier(0);
		// ---- B(t - LAG): D(i, c) += sum_j V(i, j) Hnew(c, j) over the panel's 32 columns, K order j = 4 k + grp ---------------------------
		if (do_b) {
			bf16x8 op[2][3];
			split3(rawb[0], op[0][0], op[0][1], op[0][2]);
#pragma unroll
			for (int tl = 0; tl < TPW; ++tl) {
				const int cur = tl & 1, nxt = cur ^ 1;
				if (tl + 1 < TPW) split3(rawb[(tl + 1) % 3], op[nxt][0], op[nxt][1], op[nxt][2]);
				else split_pair(vn[0][0], vn[0][1], opn);                              // the next A's first operand (whatever the slot holds when there is no next A)
				if (tl + 3 < TPW) load_tile(tl + 3, rawb[tl % 3]);
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) accB[tl][nt] = six_terms_16(op[cur], hf[nt], accB[tl][nt]);
				{
#pragma unroll
					for (int g = 0; g < 24; ++g) {
						__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
						__builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
						if (g < 8 && tl + 3 < TPW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		if (!do_b && t + 1 < T) split_pair(vn[0][0], vn[0][1], opn);
		__builtin_amdgcn_sched_barrier(0);
		stamp(4);
		if (do_o) {
			const unsigned tg = tag0 + (unsigned)to + 1u;
			const u64 w0 = __builtin_amdgcn_s_memrealtime();
			for (;;) {
				unsigned bad = 0;
#pragma unroll
				for (int q = 0; q < 4; ++q) bad |= (od[q][1] ^ tg) | (od[q][3] ^ tg);
				if (__all(bad == 0) || gave_up) break;
				if (__builtin_amdgcn_s_memrealtime() - w0 > GIVE_UP_TICKS) { gave_up = true; if (lane == 0) atomicOr(a.abort_flag, 4u); break; }
				__builtin_amdgcn_s_sleep(4);
				if (DIAG) ++retries_o;
#pragma unroll
				for (int q = 0; q < 4; ++q) od[q] = load_sc1(rs_part, obase + (unsigned)q * (8u * 256u * 64u));
			}
			stamp(5);
			// (__uint_as_float, not __builtin_bit_cast(float, od[q][i]): hipcc 7.2 folds the bit cast of a vector ELEMENT of a
			//  buffer load's result to element 0)
			// sources 8 q + g in q order, then the eight lane groups g = l >> 3: g ^ 1 by a DPP rotation, g ^ 2 and g ^ 4 across rows
			float v0 = ((__uint_as_float(od[0][0]) + __uint_as_float(od[1][0])) + __uint_as_float(od[2][0])) + __uint_as_float(od[3][0]);
			float v1 = ((__uint_as_float(od[0][2]) + __uint_as_float(od[1][2])) + __uint_as_float(od[2][2])) + __uint_as_float(od[3][2]);
			v0 = add_ror(v0, ROR8()); v1 = add_ror(v1, ROR8());
			v0 += __shfl_xor(v0, 16); v1 += __shfl_xor(v1, 16);
			v0 += __shfl_xor(v0, 32); v1 += __shfl_xor(v1, 32);
			const float sum = (lane & 8) ? v1 : v0;                                    // this lane's factor row: c = 16 w + 2 (l & 7) + ((l >> 3) & 1)
			const float num = sum * sc_c;                                              // the pending column scale of W (kernels_mu64.hip)
			const float hn = hcur * num / (den + a.eps);                               // KernelMultiplyDivide.cu:39-42
			if (grp == 0) {
				a.H[(long)jc * 64 + oc] = hn;
				s_hnew[oc] = hn;
				unsigned p0b, p1b, p2b;
				split3_scalar(hn, p0b, p1b, p2b);
				const u64 gr = (u64)p0b | ((u64)p1b << 16) | ((u64)p2b << 32) | ((u64)((tag0 + (unsigned)to + 1u) & 0xffffu) << 48);
				const int kq = slot_i >> 2;                                           // j = slot_i = 4 k + (slot_i & 3)
				u64* hd = reinterpret_cast<u64*>(a.hfrag_scratch) + (long)(xcd * SLOTS + (to % SLOTS)) * (64 * 32) +
				          (((wave * 4 + (kq >> 1)) * 64 + (slot_i & 3) * 16 + (oc & 15)) * 2 + (kq & 1));
				*hd = gr;
			}
			if (a.compute_error) {
				// per-column term of tr(H^T W^T V) (KernelTraceMultiplication.cu:43-80): sixteen rows per wave by DPP rotations, waves by the booking step
				float psum = hn * num;
				psum = add_ror(psum, ROR8()); psum = add_ror(psum, ROR4()); psum = add_ror(psum, ROR2()); psum = add_ror(psum, ROR1());
				if (lane == 0) s_ps[wave] = psum;
			}
			prev_owner_col = jc;
		}
		__builtin_amdgcn_sched_barrier(0);
		stamp(6);
	};

	if (DIAG) { c_loop0 = __builtin_amdgcn_s_memtime(); c_last = c_loop0; }
	for (int t = 0; t < T + LAG; t += 2) {
		tick(t, va[0], va[1]);
		tick(t + 1, va[1], va[0]);
	}
	if (DIAG) c_loop1 = __builtin_amdgcn_s_memtime();
	// (the last owned column was booked in tick T + 1 <= T + LAG - 1)

	// ---- results ------------------------------------------------------------------------------------------------------------------
	// (V H^T)^T partial of this group: C/D map of the 16 x 16 MFMA: register g of lane (cl, grp) is row 4 grp + g (tile row i), column cl (c = 16 nt + cl)
	float* slab = a.slabs + (long)xcd * a.slab_stride;
	const int cl = lane & 15;
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
	if (DIAG && a.stamps != nullptr) {
		// per wave, 16 words: cycles in: next panel requested + A | wait BAR_a | H columns requested, booking, publish, owner's old values
		// requested | BAR_b, second read of V requested, denominator, H operand | owner's partials requested + B | wait for the partials |
		// owner: reduce, new column | (between ticks); retries of the two waits; cycles before the loop, in the loop, after it; 100 MHz
		// ticks of the whole kernel; XCD and slot; ticks
		__builtin_amdgcn_s_waitcnt(0);
		const u64 c_exit = __builtin_amdgcn_s_memtime();
		if (lane == 0) {
			unsigned long long* o = a.stamps + 16 * ((long)blockIdx.x * 4 + wave);
			for (int i = 0; i < 8; ++i) o[i] = seg[i];
			o[8] = retries_f; o[9] = retries_o; o[10] = c_loop0 - c_entry; o[11] = c_loop1 - c_loop0; o[12] = c_exit - c_loop1;
			o[13] = __builtin_amdgcn_s_memrealtime() - t_start; o[14] = (u64)xcd * 64 + slot_i; o[15] = (u64)T;
		}
	}
	(void)t_start;
}

bool onepass_available(long mpad, int num_cus) {
	return num_cus == ONEPASS_XCDS * ONEPASS_GROUP && mpad % 16 == 0 && mpad / 16 <= (long)ONEPASS_GROUP * 4 * TPW;
}

hipError_t launch_mu64_onepass(const OnePassArgs& a, hipStream_t stream) {
	if (a.tile_rows <= 0 || a.tile_rows > ONEPASS_GROUP * 4 * TPW || a.panels <= 0) return hipErrorInvalidValue;
	static std::atomic<unsigned long long> lds_done{0ull};
#ifdef NMFAMD_DIAG_BUILD
	if (a.stamps != nullptr) {
		static std::atomic<unsigned long long> lds_done_d{0ull};
		if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_mu64_onepass<true>), LDS_TOTAL, lds_done_d); e != hipSuccess) return e;
		hipLaunchKernelGGL(k_mu64_onepass<true>, dim3(ONEPASS_XCDS * ONEPASS_GROUP), dim3(256), LDS_TOTAL, stream, a);
		return hipGetLastError();
	}
#endif
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_mu64_onepass<false>), LDS_TOTAL, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL(k_mu64_onepass<false>, dim3(ONEPASS_XCDS * ONEPASS_GROUP), dim3(256), LDS_TOTAL, stream, a);
	return hipGetLastError();
}

} // namespace nmfamd
