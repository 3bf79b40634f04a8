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
constexpr int OLAG = 2;                               // the owner duty of a tick works on the panel published OLAG ticks before: its partial sums have arrived by then
constexpr int LDS_WF = 4 * TPW * 2 * 3 * 64 * 16;     // the four waves' rows of the split image of W, fragment order
constexpr int LDS_XCH = 4 * 8 * 64 * 16;              // partial W^T V of the four waves
constexpr int LDS_TAIL = 64 * 4 + 16 + 32 + 16;
constexpr int LDS_TOTAL = LDS_WF + LDS_XCH + LDS_TAIL;
static_assert(LDS_TOTAL <= 163840, "LDS budget");
static_assert(OLAG >= 1 && OLAG + 2 <= SLOTS, "a slot must not be rewritten before its readers are done");

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

// Spin on an LDS word until it reaches `target` (the A waves of a workgroup synchronise among themselves through counters in LDS: the
// workgroup barrier would tie the B waves in).  LDS operations of a wave are performed in order, so data written before an arrive
// is visible to whoever has seen the count.
__device__ inline void lds_arrive(unsigned* cnt, int lane) {
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS traffic is done (and the compiler keeps it above this line)
	if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	asm volatile("" ::: "memory");
}
__device__ inline bool lds_wait(const unsigned* cnt, unsigned target, bool gave_up, unsigned* abort_flag, unsigned code, int lane) {
	bool fail = false;
	if (!gave_up) {
		const u64 w0 = __builtin_amdgcn_s_memrealtime();
		while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) {
			if (__builtin_amdgcn_s_memrealtime() - w0 > GIVE_UP_TICKS) { if (lane == 0) atomicOr(abort_flag, code); fail = true; break; }
			__builtin_amdgcn_s_sleep(1);
		}
	}
	asm volatile("" ::: "memory");                          // LDS reads below stay below the poll
	return gave_up || fail;
}

} // namespace

// Workgroup = 8 waves, two per SIMD, 256 registers each, in two roles:
//   A waves (0..3, one per SIMD): A(t), the exchange + publishing step, and the owner duty O(t - 1);
//   B waves (4..7, the SIMD partners): B(panel) as soon as the panel's new H columns are published -- they run behind the A waves
//     by whatever the two hand-offs take and share the matrix pipe with them: while one role waits for memory, for LDS or for
//     another workgroup, the other one's MFMAs and operand splitting fill the SIMD.
// A wave w and B wave 4 + w hold the same 80 rows.
// DIAG (diagnostic builds, tools/onepass_stamps.py): per-wave shader-cycle sums of the segments of a tick, see the stamps' layout at the end
template <bool DIAG>
__global__ __launch_bounds__(512, 2) void k_mu64_onepass(const OnePassArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	bf16x8* const wfl = reinterpret_cast<bf16x8*>(smem);
	f32x4* const xch = reinterpret_cast<f32x4*>(smem + LDS_WF);
	float* const s_hnew = reinterpret_cast<float*>(smem + LDS_WF + LDS_XCH);
	float* const s_ps = s_hnew + 64;
	unsigned* const s_cnt = reinterpret_cast<unsigned*>(s_ps + 4);     // [0] exchange full, [1] exchange free, [2] new column full, [3] new column free, [4..7] B progress
	int* const s_ctl = reinterpret_cast<int*>(s_cnt + 8);

	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int role = wave >> 2, rw = wave & 3;            // rw: which 80 rows of the slice
	const int atid = tid & 255;                           // thread number inside the role
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
	if (tid < 8) s_cnt[tid] = 0u;
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

	int trw[TPW];                                          // this wave's tile rows (clamped into the slice)
#pragma unroll
	for (int ks = 0; ks < TPW; ++ks) { const int tr = tr0 + TPW * rw + ks; trw[ks] = tr < tr1 ? tr : tr1 - 1; }
	auto split_pair = [&](const f32x4& lo4, const f32x4& hi4, bf16x8 (&o)[3]) __attribute__((always_inline)) {
		float v[8];
#pragma unroll
		for (int j = 0; j < 4; ++j) { v[j] = lo4[j]; v[4 + j] = hi4[j]; }
		split3(v, o[0], o[1], o[2]);
	};

	if (role == 0) {
		// ================================================= A waves =================================================================
		// (they win the arbitration for the SIMD: the B waves fill what the A waves leave)
		__builtin_amdgcn_s_setprio(2);
		// this wave's rows of the split image of W, into its own part of LDS: K-step ks = tile row tr0 + TPW rw + ks (zero beyond the
		// slice: those steps add nothing); fragment (ks, nb, plane) of lane l at wl[((ks * 2 + nb) * 3 + plane) * 64]
		bf16x8* const wl = wfl + rw * (TPW * 2 * 3 * 64) + lane;
		{
			const bf16x8* F = reinterpret_cast<const bf16x8*>(a.Wx3);
#pragma unroll
			for (int ks = 0; ks < TPW; ++ks) {
				const bool valid = tr0 + TPW * rw + ks < tr1;
				const int wk = trw[ks] < a.w_ks ? trw[ks] : a.w_ks;          // rows past the image: its closing all-zero step
#pragma unroll
				for (int f = 0; f < 6; ++f) {
					bf16x8 v = F[((long)wk * 6 + f) * 64 + lane];
					if (!valid) { const u32x4 z = {0u, 0u, 0u, 0u}; v = __builtin_bit_cast(bf16x8, z); }
					wl[(ks * 6 + f) * 64] = v;
				}
			}
		}
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
		__builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): this wave's fragments of W are in LDS (nobody else reads them)

		auto tick = [&](const int t, f32x4 (&vs)[TPW][2], f32x4 (&vn)[TPW][2]) __attribute__((always_inline)) {
			stamp(7);
			// first use of the panel: whatever the compiler drains here (it cannot count loads across the loop's back edge) is old
			bf16x8 op0[3];
			split_pair(vs[0][0], vs[0][1], op0);
			__builtin_amdgcn_sched_barrier(0);
			// the next panel flies during A
			if (t + 1 < T) prefetch(t + 1, vn);
			__builtin_amdgcn_sched_barrier(0);
			stamp(6);
			// ---- A(t): D(c, j) = sum_i W(i, c) V(i, j) over this wave's rows -----------------------------------------------------
			f32x16 accA[2];
#pragma unroll
			for (int nb = 0; nb < 2; ++nb)
#pragma unroll
				for (int g = 0; g < 16; ++g) accA[nb][g] = 0.f;
			bf16x8 op[2][3], wf[2][2][3];
#pragma unroll
			for (int pl = 0; pl < 3; ++pl) op[0][pl] = op0[pl];
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
			stamp(0);
			// the exchange image must be free: every A wave has read what it publishes of tick t - 1
			gave_up = lds_wait(s_cnt + 1, 4u * (unsigned)t, gave_up, a.abort_flag, 8u, lane);
			// C/D map of the 32 x 32 MFMA: register 4 q + g of lane (l31, half) is row 8 q + 4 half + g (here c = 32 nb + that), column l31 (= j).
			// Exchange image: float4 (wave, lane, nbq = 4 nb + q) at (wave * 64 + lane) * 8 + (nbq ^ (lane & 7)): a lane's eight
			// chunks are 128 contiguous bytes, the XOR spreads lanes over the banks for the writer and for the transposing reader
#pragma unroll
			for (int nb = 0; nb < 2; ++nb)
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					f32x4 v;
					v[0] = accA[nb][4 * q + 0]; v[1] = accA[nb][4 * q + 1]; v[2] = accA[nb][4 * q + 2]; v[3] = accA[nb][4 * q + 3];
					xch[(rw * 64 + lane) * 8 + ((nb * 4 + q) ^ (lane & 7))] = v;
				}
			lds_arrive(s_cnt + 0, lane);
			stamp(1);
			// The slot tick t is published into held tick t - SLOTS: its readers -- the owners, B waves of every workgroup of the group -- are
			// done with it once the new H columns of panel t - SLOTS are complete, which is what this workgroup's B waves have seen when their
			// progress has passed that panel.  (The same condition keeps the H slots safe: the owners write the columns of panel t only after
			// every workgroup has published tick t.)
			if (t >= SLOTS && !gave_up) {
				const u64 w0 = __builtin_amdgcn_s_memrealtime();
				for (;;) {
					const unsigned b0 = __hip_atomic_load(s_cnt + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), b1 = __hip_atomic_load(s_cnt + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					const unsigned b2 = __hip_atomic_load(s_cnt + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), b3 = __hip_atomic_load(s_cnt + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					const unsigned mn = min(min(b0, b1), min(b2, b3));
					if ((int)(mn - (unsigned)(t - SLOTS + 1)) >= 0) break;
					if (__builtin_amdgcn_s_memrealtime() - w0 > GIVE_UP_TICKS) { gave_up = true; if (lane == 0) atomicOr(a.abort_flag, 16u); break; }
					__builtin_amdgcn_s_sleep(2);
				}
			}
			stamp(2);
			gave_up = lds_wait(s_cnt + 0, 4u * (unsigned)(t + 1), gave_up, a.abort_flag, 8u, lane);
			stamp(3);
			// sum over the four waves (wave order) and publish: thread (j = atid / 8, nbq = atid % 8) takes c = 8 nbq .. 8 nbq + 7 of column j,
			// i.e. the chunk nbq of lanes (j, half 0) and (j, half 1); its 64 bytes of granules are bytes [64 atid, 64 atid + 64) of the
			// workgroup's slot: slot image = [column j][c] granules {value, tag}
			const unsigned tg = tag0 + (unsigned)t + 1u;
			u32x4* dst = reinterpret_cast<u32x4*>(a.part_scratch) + ((long)((xcd * SLOTS + (t % SLOTS)) * ONEPASS_GROUP + slot_i) * 256 + atid) * 4;
			const int pj = atid >> 3, pq = atid & 7;
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
			lds_arrive(s_cnt + 1, lane);
			stamp(4);
		};

		if (DIAG) { c_loop0 = __builtin_amdgcn_s_memtime(); c_last = c_loop0; }
		for (int t = 0; t < T; t += 2) {
			tick(t, va[0], va[1]);
			if (t + 1 < T) tick(t + 1, va[1], va[0]);
		}
		if (DIAG) c_loop1 = __builtin_amdgcn_s_memtime();
	} else {
		// ================================================= B waves =================================================================
		// owner arithmetic: lane l of B wave w works on factor row c = 16 w + 2 (l & 7) + ((l >> 3) & 1), reduction part k = 16 (l >> 4) .. + 15
		// (the row follows from how the owner's loads are dealt out, see O below)
		const int oc = 16 * rw + 2 * (lane & 7) + ((lane >> 3) & 1);
		const float* const grow = a.G + (long)oc * 64 + 16 * grp;      // this lane's part of row c of W^T W (read again every tick: L1 / L2)
		const float sc_c = a.scale[oc];
		// x + (x of the lane N places on, cyclically, in its row of 16 lanes) -- a DPP rotation, no LDS
		auto add_ror = [&](float x, auto ctrl_c) __attribute__((always_inline)) -> float {
			constexpr int CTRL = decltype(ctrl_c)::value;
			return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
		};
		typedef std::integral_constant<int, 0x128> ROR8;
		typedef std::integral_constant<int, 0x124> ROR4;
		typedef std::integral_constant<int, 0x122> ROR2;
		typedef std::integral_constant<int, 0x121> ROR1;
		f32x4 accB[TPW][4], hht[4];                                    // hht: H H^T tiles (16 rw .., 16 nt' ..) of the panels this workgroup books
#pragma unroll
		for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) accB[tl][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int nt = 0; nt < 4; ++nt) hht[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
		// the denominator (W^T W) H and the old value of the owned column of panel k: they wait for nobody
		float den_next = 0.f, hcur_next = 0.f;
		auto prepare_owner = [&](int k) __attribute__((always_inline)) {
			const int jn = (p0 + k) * 32 + slot_i;
			f32x4 hold[4], gq[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) { hold[u] = *reinterpret_cast<const f32x4*>(a.H + (long)jn * 64 + 16 * grp + 4 * u); gq[u] = *reinterpret_cast<const f32x4*>(grow + 4 * u); }
			hcur_next = a.H[(long)jn * 64 + oc];
			float den = 0.f;
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int i = 0; i < 4; ++i) den = fmaf(gq[u][i], hold[u][i], den);
			den += __shfl_xor(den, 16);
			den += __shfl_xor(den, 32);                                                // (0 + 1) + (2 + 3) in every lane
			den_next = den;
		};
		if (T > 0) prepare_owner(0);
		if (DIAG) { c_loop0 = __builtin_amdgcn_s_memtime(); c_last = c_loop0; }
		// iteration k: owner duty O(k) (k < T), then B(k - 1) (k >= 1), whose H columns the owners published an iteration ago
		for (int k = 0; k < T + 1; ++k) {
			const bool do_o = k < T, do_b = k >= 1;
			const int tb = k - 1;
			stamp(7);
			// the 32 sources' partial sums of the owned column: a source's 128 bytes (16 granules, c = 16 rw .. + 15) are 8 pieces of 16 bytes
			// (c = 16 rw + 2 p, + 1); load q of lane l takes piece p = l & 7 of source 8 q + (l >> 3): 128 contiguous bytes per source and instruction
			u32x4 od[4];
			const unsigned obase = part_group + (unsigned)((k % SLOTS) * ONEPASS_GROUP + (lane >> 3)) * (256u * 64u) + (unsigned)slot_i * 512u +
			                       (unsigned)rw * 128u + (unsigned)(lane & 7) * 16u;
			if (do_o) {
#pragma unroll
				for (int q = 0; q < 4; ++q) od[q] = load_sc1(rs_part, obase + (unsigned)q * (8u * 256u * 64u));
			}
			// second read of V(rows, panel k - 1): tile tl is 8 rows of 256 bytes, lane l of load kk takes row i = l & 15 at column j = 4 kk + (l >> 4)
			float rawb[3][8];
			const float* const vb0 = a.V + ((long)(p0 + tb) * (32 * 16) + lane);
			auto load_tile = [&](int tl, float (&dst)[8]) __attribute__((always_inline)) {
				const float* p = vb0 + (long)trw[tl] * a.tile_stride;
#pragma unroll
				for (int kk = 0; kk < 8; ++kk) dst[kk] = p[64 * kk];
			};
			if (do_b) { load_tile(0, rawb[0]); load_tile(1, rawb[1]); load_tile(2, rawb[2]); }
			__builtin_amdgcn_sched_barrier(0);
			// ---- O(k): this workgroup owns column `slot` of panel k, this wave its factor rows c = 16 rw .. 16 rw + 15 ---------------------------
			if (do_o) {
				const int jc = (p0 + k) * 32 + slot_i;
				const unsigned tg = tag0 + (unsigned)k + 1u;
				{
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
				}
				stamp(0);
				// (__uint_as_float, not __builtin_bit_cast(float, od[q][i]): hipcc 7.2 folds the bit cast of a vector ELEMENT of a
				//  buffer load's result to element 0)
				// sources 8 q + g in q order, then the eight lane groups g = l >> 3: g ^ 1 by a DPP rotation, g ^ 2 and g ^ 4 across rows
				float v0 = ((__uint_as_float(od[0][0]) + __uint_as_float(od[1][0])) + __uint_as_float(od[2][0])) + __uint_as_float(od[3][0]);
				float v1 = ((__uint_as_float(od[0][2]) + __uint_as_float(od[1][2])) + __uint_as_float(od[2][2])) + __uint_as_float(od[3][2]);
				v0 = add_ror(v0, ROR8()); v1 = add_ror(v1, ROR8());
				v0 += __shfl_xor(v0, 16); v1 += __shfl_xor(v1, 16);
				v0 += __shfl_xor(v0, 32); v1 += __shfl_xor(v1, 32);
				const float sum = (lane & 8) ? v1 : v0;                                // this lane's factor row: c = 16 rw + 2 (l & 7) + ((l >> 3) & 1)
				const float num = sum * sc_c;                                          // the pending column scale of W (kernels_mu64.hip)
				const float hn = hcur_next * num / (den_next + a.eps);                 // KernelMultiplyDivide.cu:39-42
				if (grp == 0) {
					a.H_out[(long)jc * 64 + oc] = hn;                                  // (not in place: the other owner waves still read the old column)
					unsigned p0b, p1b, p2b;
					split3_scalar(hn, p0b, p1b, p2b);
					const u64 gr = (u64)p0b | ((u64)p1b << 16) | ((u64)p2b << 32) | ((u64)(tg & 0xffffu) << 48);
					const int kq = slot_i >> 2;                                       // j = slot_i = 4 kq + (slot_i & 3)
					// image of a slot: 16-byte piece ((nt * 4 + q) * 64 + lane) = the two granules k = 2 q, 2 q + 1 of (c = 16 nt + (lane & 15), j = 4 k + (lane >> 4))
					u64* hd = reinterpret_cast<u64*>(a.hfrag_scratch) + (long)(xcd * SLOTS + (k % SLOTS)) * (64 * 32) +
					          (((rw * 4 + (kq >> 1)) * 64 + (slot_i & 3) * 16 + (oc & 15)) * 2 + (kq & 1));
					*hd = gr;
				}
				if (a.compute_error) {
					// per-column term of tr(H^T W^T V) (KernelTraceMultiplication.cu:43-80): this wave's sixteen rows by DPP rotations; the four
					// waves' parts are added by the host side of the iteration (launch_reduce_partials, wave order)
					float psum = hn * num;
					psum = add_ror(psum, ROR8()); psum = add_ror(psum, ROR4()); psum = add_ror(psum, ROR2()); psum = add_ror(psum, ROR1());
					if (lane == 0) a.ps[(long)rw * a.ps_stride + jc] = psum;
				}
				stamp(1);
			}
			__builtin_amdgcn_sched_barrier(0);
			if (do_b) {
				// the split columns of H of panel k - 1: every granule must carry the panel's tag.
				// image of a slot: 16-byte piece ((nt * 4 + q) * 64 + lane) = the two granules kk = 2 q, 2 q + 1 of (c = 16 nt + (lane & 15), j = 4 kk + grp)
				const unsigned hbase = hf_group + (unsigned)(tb % SLOTS) * (64u * 32u * 8u) + (unsigned)lane * 16u;
				const unsigned tg16 = (tag0 + (unsigned)tb + 1u) & 0xffffu;
				bf16x8 hf[4][3];
				{
					const u64 w0 = __builtin_amdgcn_s_memrealtime();
					for (;;) {
						unsigned bad = 0;
#pragma unroll
						for (int nt = 0; nt < 4; ++nt) {
							u32x4 d[4];
#pragma unroll
							for (int q = 0; q < 4; ++q) d[q] = load_sc1(rs_hf, hbase + (unsigned)(nt * 4 + q) * 1024u);
							u32x4 o0, o1, o2;
#pragma unroll
							for (int q = 0; q < 4; ++q) {
								// d[q] = two granules {p0 | p1 << 16, p2 | tag << 16} of kk = 2 q, 2 q + 1
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
						if (DIAG) ++retries_f;
					}
				}
				// the columns of panel k - 1 are complete: every owner of the group is done with tick k - 1's partial sums, and this wave with
				// the H slot (the A waves wait for that before they let the slot rings go round)
				if (lane == 0) __hip_atomic_store(s_cnt + 4 + rw, (unsigned)(tb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				stamp(2);
				// H H^T of the panel (reference: syrk upper, AlgorithmMultiplicativeFrobenius.h:231-232): every B wave of the group holds the whole
				// panel of new columns as MFMA operands, so ONE workgroup per panel books it -- this wave the tiles (16 rw .., 16 nt ..)
				if ((tb & (ONEPASS_GROUP - 1)) == slot_i) {
#pragma unroll
					for (int nt = 0; nt < 4; ++nt) hht[nt] = rw == 0 ? six_terms_16(hf[0], hf[nt], hht[nt]) : rw == 1 ? six_terms_16(hf[1], hf[nt], hht[nt])
					                                       : rw == 2 ? six_terms_16(hf[2], hf[nt], hht[nt]) : six_terms_16(hf[3], hf[nt], hht[nt]);
				}
				// ---- B(k - 1): D(i, c) += sum_j V(i, j) Hnew(c, j) over the panel's 32 columns, K order j = 4 kk + grp -----------------------------
				bf16x8 op[2][3];
				split3(rawb[0], op[0][0], op[0][1], op[0][2]);
#pragma unroll
				for (int tl = 0; tl < TPW; ++tl) {
					const int cur = tl & 1, nxt = cur ^ 1;
					if (tl + 1 < TPW) split3(rawb[(tl + 1) % 3], op[nxt][0], op[nxt][1], op[nxt][2]);
					if (tl + 3 < TPW) load_tile(tl + 3, rawb[tl % 3]);
#pragma unroll
					for (int nt = 0; nt < 4; ++nt) accB[tl][nt] = six_terms_16(op[cur], hf[nt], accB[tl][nt]);
					if (tl + 1 < TPW) {
#pragma unroll
						for (int g = 0; g < 24; ++g) {
							__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
							__builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
							if (g < 8 && tl + 3 < TPW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
				stamp(3);
			}
			if (k + 1 < T) prepare_owner(k + 1);
			stamp(4);
		}
		if (DIAG) c_loop1 = __builtin_amdgcn_s_memtime();
		// (V H^T)^T partial of this group: C/D map of the 16 x 16 MFMA: register g of lane (cl, grp) is row 4 grp + g (tile row i), column cl (c = 16 nt + cl)
		float* slab = a.slabs + (long)xcd * a.slab_stride;
		const int cl = lane & 15;
#pragma unroll
		for (int tl = 0; tl < TPW; ++tl) {
			const int tr = tr0 + TPW * rw + tl;
			if (tr < tr1) {
#pragma unroll
				for (int nt = 0; nt < 4; ++nt)
#pragma unroll
					for (int g = 0; g < 4; ++g) slab[((long)tr * 16 + 4 * grp + g) * 64 + 16 * nt + cl] = accB[tl][nt][g];
			}
		}
		// this workgroup's part of H H^T (zero for most): rows 16 rw + 4 grp + g, columns 16 nt + cl
		float* hp = a.hh_part + (long)(xcd * ONEPASS_GROUP + slot_i) * 4096;
#pragma unroll
		for (int nt = 0; nt < 4; ++nt)
#pragma unroll
			for (int g = 0; g < 4; ++g) hp[(long)(16 * rw + 4 * grp + g) * 64 + 16 * nt + cl] = hht[nt][g];
	}
	if (DIAG && a.stamps != nullptr) {
		// per wave, 16 words.  A waves: cycles in: next panel requested + A | waits around the exchange (free, B waves, full) | publish |
		// owner: loads, wait for the partials | reduce, new column, wait for the others | booking | - | (between ticks).
		// B waves: H operand (wait + repack) | B | ... ; then retries of the two waits; cycles before the loop, in the loop, after it; 100 MHz
		// ticks of the whole kernel; XCD * 64 + slot; ticks
		__builtin_amdgcn_s_waitcnt(0);
		const u64 c_exit = __builtin_amdgcn_s_memtime();
		if (lane == 0) {
			unsigned long long* o = a.stamps + 16 * ((long)blockIdx.x * 8 + wave);
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
		hipLaunchKernelGGL(k_mu64_onepass<true>, dim3(ONEPASS_XCDS * ONEPASS_GROUP), dim3(512), LDS_TOTAL, stream, a);
		return hipGetLastError();
	}
#endif
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_mu64_onepass<false>), LDS_TOTAL, lds_done); e != hipSuccess) return e;
	hipLaunchKernelGGL(k_mu64_onepass<false>, dim3(ONEPASS_XCDS * ONEPASS_GROUP), dim3(512), LDS_TOTAL, stream, a);
	return hipGetLastError();
}

} // namespace nmfamd
