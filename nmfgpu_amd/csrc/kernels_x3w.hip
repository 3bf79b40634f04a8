// kernels_x3w.hip -- the split-operand product (kernels_x3.hip: fp32 operands cut exactly into three bf16 terms, six cross terms per product) with the factor
// fragments SHARED by the four waves of a workgroup -- round 5.
//
// What round 5 measured on k_factor_product_x3 (profiles/r05_x3_shape.md): its loop is paced by the memory system (5.55 TB/s whatever the clock), a kernel that
// only loads re-reads the image at 6.3 TB/s, and the difference is the factor fragments: 0.75 bytes per byte of V through the SAME vector-memory path, because
// the four waves of a workgroup split the reduction range and each streams its own K-steps' fragments (V only: 1.24 us per K-step; V + fragments: 1.40).
// Here the waves split the ROWS instead:
//   * workgroup = 4 waves = 256 rows x 64 panel columns x one K slice; wave w owns rows 64 w .. 64 w + 63 (4 row blocks of 16) for ALL the slice's K-steps;
//   * v_mfma_f32_16x16x32_bf16 (a 64-row wave tile needs no transposition on it: a lane's 16-byte load of V is four consecutive rows of one column = one row of
//     each of the four row blocks; and the chip holds a higher clock on this shape), reduction in double steps of 32 k;
//   * the 12 KB of factor fragments of a double step are fetched ONCE per workgroup: wave w requests the three planes of column block w by LDS-DMA
//     (global_load_lds: no registers) into a three-slot LDS ring; after one barrier per double step every wave reads all twelve fragments from LDS (ds_read_b128
//     straight into the accumulator half of the register file, where an MFMA takes its A operand from as well) -- a quarter of the fragment traffic on the
//     vector-memory path, the rest on the LDS path that the kernel does not otherwise use;
//   * V: a ring of two double steps per wave (8 loads each), so 16 + 3 requests in flight per wave;
//   * no cross-wave sum: the waves own disjoint rows; a lane's accumulator is four consecutive panel columns of one row = one 16-byte store into the slab.
// Every load and every wait is written out (asm), as in the 16 x 16 x 32 experiment this file grew from: hipcc loads into VGPRs only, and a wait it counts for its
// own loads would wait for the written-out ones as well (in-order vmcnt).  Loads sit behind MFMAs, tied ("+v") to the operand register the MFMAs on either side
// read; counted s_waitcnt statements name the registers they release.
// K slices: one per workgroup (grid = x-tiles x slices): twice as many slabs as the 128-row kernel for the same number of workgroups.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "tuning.h"
#include "split3.h"
#include "inverse_gj64.h"
#include "gram_image.h"

namespace nmfamd {

namespace {

constexpr int X3W_SLOTS = 3;                      // LDS ring of factor fragments (double steps)
constexpr int X3W_SLOT_BYTES = 12 * 1024;         // 4 column blocks x 3 planes x 64 lanes x 16 B
constexpr int X3W_LDS_BYTES = 40 * 1024;          // the ring (36 KB); the passengers of the launch need less (gram_image.h, inverse_gj64.h)

template <bool TR, int IMG>
constexpr int x3w_a_off(int i) {       // byte offset of load i (0..7) of a wave's four row blocks, from the lane's address
	return IMG == 16 ? (TR ? ((i >> 1) * 256 + 4 * (i & 1)) * 4 : i * 64) : i * 128 * 4;
}
constexpr int x3w_f_off(int cb, int pl) { return (pl * 64 + 16 * (cb & 1)) * 16; }     // fragment (column block cb, plane pl) from the address of its 32-column block

template <int OFF> __device__ inline void x3w_load_v(f32x4& dst, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:%c2" : "=v"(dst) : "v"(p), "i"(OFF) : "memory"); }
template <int OFF> __device__ inline void x3w_load_v_tied(f32x4& dst, const float* p, bf16x8& tie) {
	asm volatile("global_load_dwordx4 %0, %2, off offset:%c3" : "=v"(dst), "+v"(tie) : "v"(p), "i"(OFF) : "memory");
}
// LDS-DMA: 16 bytes per lane from p to LDS byte address lds_base (wave-uniform) + 16 lane.  M0 carries the base and belongs to hipcc: saved and restored inside the statement.
// (no instruction offset: the global address is complete in p, the LDS address complete in M0)
template <int OFF> __device__ inline void x3w_dma(const bf16x8* p, unsigned lds_base) {
	unsigned keep;
	const bf16x8* q = p + OFF / 16;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(q), "s"(lds_base) : "memory");
}
template <int OFF> __device__ inline void x3w_dma_tied(const bf16x8* p, unsigned lds_base, bf16x8& tie) {
	unsigned keep;
	const bf16x8* q = p + OFF / 16;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" : "=&s"(keep), "+v"(tie) : "v"(q), "s"(lds_base) : "memory");
}
// fragment from the ring into the accumulator file
template <int OFF> __device__ inline void x3w_lds_read(u32x4& dst, unsigned lds_addr) { asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=a"(dst) : "v"(lds_addr), "i"(OFF) : "memory"); }
template <int OFF> __device__ inline void x3w_lds_read_tied(u32x4& dst, unsigned lds_addr, bf16x8& tie) {
	asm volatile("ds_read_b128 %0, %2 offset:%c3" : "=a"(dst), "+v"(tie) : "v"(lds_addr), "i"(OFF) : "memory");
}
template <int N> __device__ inline void x3w_wait8(f32x4 (&g)[8]) {
	asm volatile("s_waitcnt vmcnt(%c8)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]) : "i"(N) : "memory");
}
__device__ inline void x3w_wait_lds(u32x4 (&f)[4][3]) {
	asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(f[0][0]), "+a"(f[0][1]), "+a"(f[0][2]), "+a"(f[1][0]), "+a"(f[1][1]), "+a"(f[1][2]),
	             "+a"(f[2][0]), "+a"(f[2][1]), "+a"(f[2][2]), "+a"(f[3][0]), "+a"(f[3][1]), "+a"(f[3][2]) :: "memory");
}
// this wave's LDS-DMA requests of the slot about to be read have landed (all but the N youngest requests of the wave are done), then the workgroup's barrier:
// everybody's have
template <int N> __device__ inline void x3w_ring_barrier() { asm volatile("s_waitcnt vmcnt(%c0)\n\ts_barrier" :: "i"(N) : "memory"); }

#ifndef NMFAMD_XCD_REMAP
#define NMFAMD_XCD_REMAP 1
#endif

template <bool TR, int IMG, int DIAG = 0>
__global__ __launch_bounds__(256, 1) void k_factor_product_x3w(
	const float* __restrict__ A, long tile_stride,
	const bf16x8* __restrict__ F, int NBT,
	float* __restrict__ slabs, long slab_stride, int RP,
	int steps_total, int xtiles, int splits, int rows_total, GramReduceArgs rg, unsigned long long* stamps) {
	static_assert(!TR || IMG == 16, "the y-tiled form reads 16-row tiles");
	constexpr int NC = 4;
	unsigned long long t_loop0 = 0, t_loop1 = 0, r_loop0 = 0, r_loop1 = 0, r_entry = 0, r_tail = 0;
	if (DIAG != 0) r_entry = __builtin_amdgcn_s_memrealtime();
	extern __shared__ __attribute__((aligned(16))) float lds[];
	const int pblocks = xtiles * splits;
	if (blockIdx.x >= (unsigned)pblocks) {
		// passengers, as in k_factor_product_x3: the 64 x 64 inverse of the least-squares algorithms, or a Gram matrix from the split image
		if (rg.inv_a != nullptr) inverse_gj64_body<float, 4>(rg.inv_a, 64, rg.inv_r, rg.inv_out, rg.inv_offdiag, rg.inv_diag);
		else if (rg.image != nullptr) gram_image_block(rg, blockIdx.x - pblocks, lds);
		return;
	}
	int vb = blockIdx.x;
	if (NMFAMD_XCD_REMAP) {
		// blocks b and b + 8 share an XCD: each XCD takes a contiguous range of (slice, x-tile) pairs, so one K slice of the factor image lives in one or two L2s
		const int q8 = pblocks / 8, r8 = pblocks % 8, xcd = vb % 8, idx = vb / 8;
		vb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
	}
	const int xt = vb % xtiles, sp = vb / xtiles;
	const long fstep = (long)NBT * 192;                 // factor fragments per K-step of 16
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	const int r16 = lane & 15, kq = lane >> 4;
	// the reduction range in double steps of 32 k, dealt to the slices; a double step past an odd range's end meets the all-zero K-step that closes the factor
	// image (index steps_total) with a re-read of the last valid K-step of A
	const int units = (steps_total + 1) / 2;
	const int S0 = (int)(((long)units * sp) / splits), S1 = (int)(((long)units * (sp + 1)) / splits);
	const int kend = steps_total - 1;
	// this wave's 64 rows; a wave beyond the image's rows (the last x-tile of an odd number of 128-row tiles) computes on the last 64 valid rows and stores nothing
	int xw = 256 * xt + 64 * wave;
	const bool stores = xw < rows_total;
	if (!stores) xw = rows_total - 64;

	f32x4 acc[4][NC];
#pragma unroll
	for (int rb = 0; rb < 4; ++rb)
#pragma unroll
		for (int cb = 0; cb < NC; ++cb)
#pragma unroll
			for (int e = 0; e < 4; ++e) acc[rb][cb][e] = 0.f;

	if (S1 > S0) {
		constexpr bool RA = DIAG == 0 || DIAG == 2 || DIAG >= 4, RF = DIAG == 0 || DIAG >= 3;       // refill V / the fragments (measurement forms switch them off)
		// lane part of the streamed operand's address; + the lane's K-step (2 S + kq / 2, clamped) times its stride; loads at constant offsets (x3w_a_off)
		const float* ap = IMG == 16 ? (TR ? A + ((long)xw + r16) * 16 + 8 * (kq & 1)
		                                  : A + ((long)(xw >> 4) + (r16 >> 2)) * tile_stride + (8 * (kq & 1)) * 16 + 4 * (r16 & 3))
		                            : A + (long)(xw >> 7) * tile_stride + (8 * (kq & 1)) * 128 + (xw & 127) + 4 * r16;
		const long kstride = IMG == 16 ? (TR ? tile_stride : 256) : 16 * 128;            // floats per K-step of 16
		auto a_step = [&](int S) -> const float* {
			int so = 2 * S + (kq >> 1);
			so = so < kend ? so : kend;
			return ap + (long)so * kstride;
		};
		// this wave's share of a double step's fragments: column block `wave`, three planes; lane part as the MFMA's operand map wants it
		const bf16x8* fp = F + (long)(kq >> 1) * fstep + (kq & 1) * 32 + r16 + (long)(wave >> 1) * 192;      // + 2 S * fstep; + x3w_f_off(wave, plane)
		const unsigned lds0 = (unsigned)(uintptr_t)lds;                 // (LDS byte address of the ring)
		const unsigned lds_lane = lds0 + 16u * (unsigned)lane;
		f32x4 va[2][8];
		u32x4 fb[2][NC][3];
		bf16x8 op[2][3];
#define X3W_IC(n) std::integral_constant<int, (n)>{}
		auto dma = [&](int S, int slot) __attribute__((always_inline)) {
			const bf16x8* f_ = fp + (long)(2 * S) * fstep;
			const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot * X3W_SLOT_BYTES + (unsigned)wave * 3072u);
			if (wave & 1) { x3w_dma<x3w_f_off(1, 0)>(f_, base); x3w_dma<x3w_f_off(1, 1)>(f_, base + 1024u); x3w_dma<x3w_f_off(1, 2)>(f_, base + 2048u); }
			else { x3w_dma<x3w_f_off(0, 0)>(f_, base); x3w_dma<x3w_f_off(0, 1)>(f_, base + 1024u); x3w_dma<x3w_f_off(0, 2)>(f_, base + 2048u); }
		};
		auto read_all = [&](u32x4 (&f)[NC][3], int slot) __attribute__((always_inline)) {
			const unsigned a_ = lds_lane + (unsigned)slot * X3W_SLOT_BYTES;
#define X3W_RD(i) x3w_lds_read<(i) * 1024>(f[(i) / 3][(i) % 3], a_);
			X3W_RD(0) X3W_RD(1) X3W_RD(2) X3W_RD(3) X3W_RD(4) X3W_RD(5) X3W_RD(6) X3W_RD(7) X3W_RD(8) X3W_RD(9) X3W_RD(10) X3W_RD(11)
#undef X3W_RD
		};
		// ---- prologue: two double steps of V, two of fragments; the first step's fragments into registers ----
		{
			const int Sb = S0 + 1 < S1 ? S0 + 1 : S1 - 1;
			const float* s_ = a_step(S0);
			const float* s1_ = a_step(Sb);
#define X3W_PRO(i) x3w_load_v<x3w_a_off<TR, IMG>(i)>(va[0][i], s_); x3w_load_v<x3w_a_off<TR, IMG>(i)>(va[1][i], s1_);
			X3W_PRO(0) X3W_PRO(1) X3W_PRO(2) X3W_PRO(3) X3W_PRO(4) X3W_PRO(5) X3W_PRO(6) X3W_PRO(7)
#undef X3W_PRO
			dma(S0, 0);
			dma(Sb, 1);
			x3w_wait8<0>(va[0]); x3w_wait8<0>(va[1]);
			x3w_ring_barrier<0>();
			read_all(fb[0], 0);
			if (!RF) read_all(fb[1], 0);
			x3w_wait_lds(fb[0]);
			if (!RF) x3w_wait_lds(fb[1]);
		}
		__builtin_amdgcn_sched_barrier(0);
		if (DIAG != 0) { t_loop0 = __builtin_amdgcn_s_memtime(); r_loop0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
		auto operand = [&](int set, int rb, float (&v)[8]) {
#pragma unroll
			for (int j = 0; j < 8; ++j) v[j] = TR ? va[set][2 * rb + (j >> 2)][j & 3] : va[set][j][rb];
		};
		{ float v[8]; operand(0, 0, v); split3(v, op[0][0], op[0][1], op[0][2]); }
		// One double step = four phases (row blocks).  Phase rb: the 24 MFMAs of its row block, interleaved with the split of row block rb + 1 (phase 3: row block 0 of
		// the NEXT double step).  u = register set (V and fragments) of THIS step; slot = its place in the LDS ring.
		// Requests of the wave, in issue order: phase 1: its share of the fragments of step S + 2 (3 LDS-DMA) into ring slot (slot + 2) % 3 -- last read during step S - 1,
		// behind two barriers; phase 3: V of step S + 2 (8 loads) into the set whose last split ran in phase 2.
		// Waits: phase 0: this wave's DMA of step S + 1 (requested in phase 1 of step S - 1; younger: 8 loads of V) + the barrier, then the twelve fragments of step
		// S + 1 are read from LDS into the other register set during phases 0 - 2 (used from the next step's head: lgkmcnt(0) there); phase 3: V of step S + 1
		// (requested in phase 3 of step S - 1; younger: the 3 DMA requests of phase 1).
		constexpr int NDMA = RF ? 3 : 0, NA = RA ? 8 : 0;
		auto step = [&](auto U, int S, int slot) __attribute__((always_inline)) {
			constexpr int u = decltype(U)::value;
			int Sn = S + 2;
			Sn = Sn < S1 ? Sn : S1 - 1;                                         // past this workgroup's slice: a harmless re-read
			const float* sn_ = a_step(Sn);
			const bf16x8* fn_ = fp + (long)(2 * Sn) * fstep;
			const int slot1 = slot + 1 < X3W_SLOTS ? slot + 1 : slot + 1 - X3W_SLOTS, slot2 = slot + 2 < X3W_SLOTS ? slot + 2 : slot + 2 - X3W_SLOTS;
			const unsigned rd_ = lds_lane + (unsigned)slot1 * X3W_SLOT_BYTES;
			const unsigned dma_ = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot2 * X3W_SLOT_BYTES + (unsigned)wave * 3072u);
			x3w_wait_lds(fb[u]);
			__builtin_amdgcn_sched_barrier(0);
			auto phase = [&](auto RB) __attribute__((always_inline)) {
				constexpr int rb = decltype(RB)::value, cur = rb & 1, nxt = cur ^ 1, nrb = (rb + 1) & 3;
				if constexpr (rb == 0) { if (RF) x3w_ring_barrier<NA>(); }
				if constexpr (rb == 3) x3w_wait8<NDMA>(va[u ^ 1]);
				{ float v[8]; operand(rb == 3 ? (u ^ 1) : u, nrb, v); split3(v, op[nxt][0], op[nxt][1], op[nxt][2]); }
				// MFMA (t, cb), t = term (smallest first: planes (fragment, operand) = (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)), then the request slot behind it: s = 4 t + cb
				auto issue = [&](auto SLOT, bf16x8& tie) __attribute__((always_inline)) {
					constexpr int s = decltype(SLOT)::value;
					// phases 0 - 2: four fragments each from LDS (slots 2, 7, 12, 17); phase 1: the wave's three DMA requests (slots 4, 9, 14); phase 3: eight loads of V
					if constexpr (RF && rb < 3 && (s == 2 || s == 7 || s == 12 || s == 17)) {
						constexpr int i = 4 * rb + (s == 2 ? 0 : s == 7 ? 1 : s == 12 ? 2 : 3);
						x3w_lds_read_tied<i * 1024>(fb[u ^ 1][i / 3][i % 3], rd_, tie);
					}
					if constexpr (RF && rb == 1 && (s == 4 || s == 9 || s == 14)) {
						constexpr int pl = s == 4 ? 0 : s == 9 ? 1 : 2;
						if (wave & 1) x3w_dma_tied<x3w_f_off(1, pl)>(fn_, dma_ + 1024u * pl, tie); else x3w_dma_tied<x3w_f_off(0, pl)>(fn_, dma_ + 1024u * pl, tie);
					}
					if constexpr (RA && rb == 3 && (s & 1) == 1 && s < 16) x3w_load_v_tied<x3w_a_off<TR, IMG>(s >> 1)>(va[u][s >> 1], sn_, tie);
				};
				auto term = [&](auto T, auto PF, auto PO) __attribute__((always_inline)) {
					constexpr int t = decltype(T)::value, pf = decltype(PF)::value, po = decltype(PO)::value;
					acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[u][0][pf]), op[cur][po], acc[rb][0], 0, 0, 0);
					issue(X3W_IC(t * NC + 0), op[cur][po]);
					acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[u][1][pf]), op[cur][po], acc[rb][1], 0, 0, 0);
					issue(X3W_IC(t * NC + 1), op[cur][po]);
					acc[rb][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[u][2][pf]), op[cur][po], acc[rb][2], 0, 0, 0);
					issue(X3W_IC(t * NC + 2), op[cur][po]);
					acc[rb][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[u][3][pf]), op[cur][po], acc[rb][3], 0, 0, 0);
					issue(X3W_IC(t * NC + 3), op[cur][po]);
				};
				term(X3W_IC(0), X3W_IC(0), X3W_IC(2));
				term(X3W_IC(1), X3W_IC(2), X3W_IC(0));
				term(X3W_IC(2), X3W_IC(1), X3W_IC(1));
				term(X3W_IC(3), X3W_IC(0), X3W_IC(1));
				term(X3W_IC(4), X3W_IC(1), X3W_IC(0));
				term(X3W_IC(5), X3W_IC(0), X3W_IC(0));
#pragma unroll
				for (int gi = 0; gi < 24; ++gi) {
					__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
					__builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU (44 of the split)
				}
				__builtin_amdgcn_sched_barrier(0);
			};
			phase(X3W_IC(0)); phase(X3W_IC(1)); phase(X3W_IC(2)); phase(X3W_IC(3));
		};
		int S = S0, slot = 0;
		for (; S + 2 <= S1; S += 2) {
			step(std::integral_constant<int, 0>{}, S, slot);
			slot = slot + 1 < X3W_SLOTS ? slot + 1 : 0;
			step(std::integral_constant<int, 1>{}, S + 1, slot);
			slot = slot + 1 < X3W_SLOTS ? slot + 1 : 0;
		}
		if (S < S1) step(std::integral_constant<int, 0>{}, S, slot);
		// the last steps' requests (re-reads that nobody uses) are still in flight and hipcc does not know of them: wait, and keep their destination registers live until then
		x3w_wait8<0>(va[0]); x3w_wait8<0>(va[1]);
		x3w_wait_lds(fb[0]); x3w_wait_lds(fb[1]);
		__builtin_amdgcn_sched_barrier(0);
#undef X3W_IC
		if (DIAG != 0) { __builtin_amdgcn_sched_barrier(0); t_loop1 = __builtin_amdgcn_s_memtime(); r_loop1 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
	}

	if (DIAG != 0) { __builtin_amdgcn_sched_barrier(0); r_tail = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); }
	// every wave stores its own 64 x 64 tile: accumulator register e of lane l is panel column 16 cb + 4 (l / 16) + e of the lane's row
	if (stores) {
		float* slab = slabs + (long)sp * slab_stride;
#pragma unroll
		for (int rb = 0; rb < 4; ++rb) {
			const int x = xw + (TR ? 16 * rb + r16 : 4 * r16 + rb);
#pragma unroll
			for (int cb = 0; cb < NC; ++cb) *reinterpret_cast<f32x4*>(slab + (long)x * RP + 16 * cb + 4 * kq) = acc[rb][cb];
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

static thread_local hipEvent_t t_ev_start = nullptr, t_ev_stop = nullptr;

template <bool TR, int IMG, int DIAG = 0>
hipError_t launch_x3w(const FactorProductPlan& p, const float* A, long tile_stride, const void* F, int RP, float* slabs, long slab_stride, int rows_total,
                      hipStream_t stream, const GramReduceArgs* rg, unsigned long long* stamps) {
	GramReduceArgs none = {nullptr, 0, nullptr, nullptr, 0};
	const bool wanted = rg != nullptr && (rg->inv_a != nullptr || rg->image != nullptr);
	const int passengers = !wanted ? 0 : (rg->inv_a != nullptr ? 1 : (rg->ksplit > 1 ? GRAM_IMAGE_TILES * rg->ksplit : GRAM_REDUCE_BLOCKS));
	dim3 grid(p.xtiles * p.splits + passengers), block(256);
	static std::atomic<unsigned long long> lds_done{0ull};
	if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_factor_product_x3w<TR, IMG, DIAG>), X3W_LDS_BYTES, lds_done); e != hipSuccess) return e;
	if (t_ev_start != nullptr && t_ev_stop != nullptr) {
		const hipEvent_t e0 = t_ev_start, e1 = t_ev_stop;
		t_ev_start = t_ev_stop = nullptr;
		hipExtLaunchKernelGGL((k_factor_product_x3w<TR, IMG, DIAG>), grid, block, (std::uint32_t)X3W_LDS_BYTES, stream, e0, e1, 0u,
		                      A, tile_stride, reinterpret_cast<const bf16x8*>(F), RP / 32, slabs, slab_stride, RP, p.steps_total, p.xtiles, p.splits, rows_total, wanted ? *rg : none, stamps);
		return hipGetLastError();
	}
	hipLaunchKernelGGL((k_factor_product_x3w<TR, IMG, DIAG>), grid, block, X3W_LDS_BYTES, stream,
	                   A, tile_stride, reinterpret_cast<const bf16x8*>(F), RP / 32, slabs, slab_stride, RP, p.steps_total, p.xtiles, p.splits, rows_total, wanted ? *rg : none, stamps);
	return hipGetLastError();
}

} // namespace

// x-tiles of 256 rows and K slices (one per workgroup) that fill the chip; at least 12 double steps per slice; `reserve` CUs stay free for passengers
void plan_x3w(long rows_total, int KS, int num_cus, int reserve, int* xtiles, int* splits) {
	*xtiles = (int)((rows_total + 255) / 256);
	const int by_fill = std::max(1, (num_cus - reserve) / std::max(1, *xtiles));
	const int by_depth = std::max(1, KS / 24);
	*splits = std::max(1, std::min(by_fill, by_depth));
}

// The split-operand product at padded rank 64 from 256-row workgroups (see the head of this file).  p.xtiles / p.splits from plan_x3w, p.steps_total = K-steps of 16;
// rows_total = rows of the output index the image and the slabs hold (a multiple of 128); passengers: the 64 x 64 inverse or a Gram matrix from a split image
// (not the partial-matrix form: callers that need it use launch_factor_product_x3).  image_tile 16 (either form) or 128 (x-tiled only).
hipError_t launch_factor_product_x3w(const FactorProductPlan& p, const float* A, long tile_stride, const void* F, int RP, float* slabs, long slab_stride, long rows_total,
                                     hipStream_t stream, const GramReduceArgs* rg, unsigned long long* stamps, bool y_tiled, int image_tile, hipEvent_t ev_start, hipEvent_t ev_stop) {
	if (RP != 64 || rows_total < 128 || rows_total % 128 != 0 || p.xtiles != (int)((rows_total + 255) / 256) || p.splits < 1 || (image_tile != 16 && image_tile != 128) || (y_tiled && image_tile != 16))
		return hipErrorInvalidValue;
	if (rg != nullptr && rg->partials != nullptr) return hipErrorInvalidValue;
	t_ev_start = ev_start; t_ev_stop = ev_stop;
	struct Clear { ~Clear() { t_ev_start = t_ev_stop = nullptr; } } clear_on_exit;
#ifdef NMFAMD_DIAG_BUILD
	if (stamps != nullptr) {
		// stamped forms (tools/stamp_x3.py): last digit of NMFAMD_X3_VARIANT: 0 = no refill, 1 = V only, 2 = fragments only, 3 = the production loop
		static const int yv = [] { const char* e = tuning_env("NMFAMD_X3_VARIANT"); return e ? std::atoi(e) : 0; }();
		if (image_tile != 16) return hipErrorNotSupported;
		if (!y_tiled) switch (yv % 10) {
			case 0: return launch_x3w<false, 16, 1>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, stamps);
			case 1: return launch_x3w<false, 16, 2>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, stamps);
			case 2: return launch_x3w<false, 16, 3>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, stamps);
			default: return launch_x3w<false, 16, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, stamps);
		}
		switch (yv % 10) {
			case 0: return launch_x3w<true, 16, 1>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, stamps);
			case 1: return launch_x3w<true, 16, 2>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, stamps);
			case 2: return launch_x3w<true, 16, 3>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, stamps);
			default: return launch_x3w<true, 16, 4>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, stamps);
		}
	}
#else
	if (stamps != nullptr) return hipErrorNotSupported;
#endif
	if (image_tile == 16) {
		return y_tiled ? launch_x3w<true, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, nullptr)
		               : launch_x3w<false, 16>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, nullptr);
	}
	return launch_x3w<false, 128>(p, A, tile_stride, F, RP, slabs, slab_stride, (int)rows_total, stream, rg, nullptr);
}

} // namespace nmfamd
