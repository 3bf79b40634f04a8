// tri_gram_tile.h -- the 256 x 256 Gram matrix of a factor panel from its bf16 fragments as PASSENGER workgroups of the bf16 factor product
// (kernels_bf16.hip) at padded rank 256 -- round 4.
//
// Stand-alone (kernels_tri.hip: k_gram_tri_bf16 + k_gram_tri_reduce_image) the Gram matrix of the operand a product multiplies V with costs two launches
// (15 + 9 us for W's 50 000 rows at config 4, 8 + 9 for H's 6 250 columns).  The product launch that follows does not depend on the result (its consumer is
// the update kernel behind that launch) and its grid leaves CUs free (config 4: 224 of 256 workgroups) -- so the Gram matrix rides there:
//   * TRI_PASSENGERS = 32 workgroups of four waves = 16 K slices x 2 halves of the 36 upper-triangle 32 x 32 tiles (18 tiles each, 4 - 5 per wave).  Per
//     K-step the workgroup fetches the eight fragment blocks once (8 KiB, two 16-byte loads per thread, TRI_RIDE_RING K-steps in flight), parks them in a
//     two-slot LDS ring, and every wave multiplies the blocks of its tiles from there -- k_gram_tri_bf16's loop on half the tiles: 0.8 MB per workgroup for
//     W's 3 125 K-steps (a tile per workgroup over the whole range, the first form tried, pulls 6.25 MB through one CU's L2 port: 180 us, longer than the product).
//   * each workgroup leaves its 18 partial tiles, releases them (agent scope) and counts itself in; the LAST of the sixteen of a half adds the sixteen partials in
//     slice order -- whoever is last, the order is fixed -- and writes the finished tiles: the fp32 matrix (both sides of the diagonal), the diagonal (the pending
//     column scale's sums of squares) and the tiles' fragments of the split image the update kernel multiplies with (store_split3, as k_gram_tri_reduce_image lays
//     them out).  Nobody waits for anybody: no co-residency assumption, safe beside other kernels and on a shared device.  The counters are left at zero.
// Measured at config 4 (bench.py --workload c4, us per iteration): no passengers 437.0, (S H)(S H)^T riding 429.2, W^T W riding (W^T V planned with 8 K slices
// instead of 9) 431.8, both 423.0 -- the four launches they replace took 41 us, but a product launch with busy passengers runs 10 - 12 us longer than with
// 32 idle CUs (224-workgroup V (S H)^T: 157 us alone, 160 with passengers that return at once, 165 without the fences, 170 as shipped).
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.h"
#include "split3.h"

namespace nmfamd {

#ifndef TRI_RIDE_RING
#define TRI_RIDE_RING 6               // K-steps in flight per thread (two 16-byte loads each)
#endif
constexpr int TRI_RIDE_SLICES = TRI_PASSENGERS / 2;
constexpr int TRI_RIDE_LDS_BYTES = 2 * 512 * 16 + 32 * 33 * 4;      // two K-steps of fragments + one finished tile

typedef float tg_f32x16 __attribute__((ext_vector_type(16)));

// tile t of the upper triangle, row by row: (i, j), i <= j  (kernels_tri.hip's tri_tile)
__device__ inline void tri_ride_tile(int t, int& i, int& j) {
	i = 0;
	while (t >= 8 - i) { t -= 8 - i; ++i; }
	j = i + t;
}

// the finished tile (i, j) from s_tile[r][33] -> G (both triangles), diag, split image; 256 threads
__device__ inline void tri_ride_emit(const float* s_tile, int i, int j, float* __restrict__ G, bf16x8* __restrict__ x3, float* __restrict__ diag) {
	const int tid = threadIdx.x;
	const bool offdiag = i != j;
	// a diagonal tile keeps its upper triangle and mirrors it (G exactly symmetric, as k_gram_tri_reduce_image leaves it)
	auto val = [&](int r, int c) -> float { return (!offdiag && r > c) ? s_tile[c * 33 + r] : s_tile[r * 33 + c]; };
	{
		const int r = tid >> 3, c4 = 4 * (tid & 7);
		f32x4 o;
#pragma unroll
		for (int k = 0; k < 4; ++k) o[k] = val(r, c4 + k);
		*reinterpret_cast<f32x4*>(G + (long)(32 * i + r) * 256 + 32 * j + c4) = o;
		if (offdiag) {
			// the mirrored tile: rows of block j, columns of block i
#pragma unroll
			for (int k = 0; k < 4; ++k) o[k] = s_tile[(c4 + k) * 33 + r];
			*reinterpret_cast<f32x4*>(G + (long)(32 * j + r) * 256 + 32 * i + c4) = o;
		} else if (diag != nullptr && tid < 32) diag[32 * i + tid] = s_tile[tid * 33 + tid];
	}
	if (x3 != nullptr) {
		// split image, fragment A(c, k) = G(k, c): K-step (k >> 4), half ((k >> 3) & 1), column block nb, lane = column within the block
		if (tid < 128) {
			const int q = tid >> 5, r = tid & 31;          // k = 32 i + 8 q + kk, c = 32 j + r
			float v8[8];
#pragma unroll
			for (int kk = 0; kk < 8; ++kk) v8[kk] = val(8 * q + kk, r);
			store_split3(x3, (32 * i + 8 * q) >> 4, 8, j, q & 1, r, v8);
		} else if (offdiag) {
			const int mm = (tid - 128) >> 5, r = tid & 31;  // k' = 32 j + 8 mm + kk, c = 32 i + r: G(k', c) = tile(r, 8 mm + kk)
			float v8[8];
#pragma unroll
			for (int kk = 0; kk < 8; ++kk) v8[kk] = s_tile[r * 33 + 8 * mm + kk];
			store_split3(x3, (32 * j + 8 * mm) >> 4, 8, i, mm & 1, r, v8);
		}
	}
}

// passenger p of TRI_PASSENGERS (256 threads): half = p & 1 takes tiles t = 2 k + half, slice = p >> 1 the K-steps [slice, slice + 1) * steps_total / 16.
// frags: [(ks * 8 + nb) * 64 + lane]; partial: [slice][36][1024] floats; counters: two unsigned, zero between launches; lds: TRI_RIDE_LDS_BYTES, 16-byte aligned
// (always inlined: a second instantiation of the product kernel in the translation unit once made hipcc emit a real call here -- in a kernel that holds 224
//  accumulator registers that was 178 -> 248 VGPRs and 372 bytes of scratch per lane; profiles/r05_c4_product_timeline.txt)
__device__ __forceinline__ void tri_gram_passenger(const bf16x8* __restrict__ frags, int steps_total, int p, float* __restrict__ partial, unsigned* __restrict__ counters,
                                          float* __restrict__ G, bf16x8* __restrict__ x3, float* __restrict__ diag, void* lds) {
	constexpr int GB = TRI_RIDE_RING, TPW = 5;
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int half = p & 1, slice = p >> 1;
	const int s0 = (int)(((long)steps_total * slice) / TRI_RIDE_SLICES), s1 = (int)(((long)steps_total * (slice + 1)) / TRI_RIDE_SLICES);
	const int steps = s1 - s0;
	bf16x8* buf = reinterpret_cast<bf16x8*>(lds);                                         // [2][512]
	float* s_tile = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + 2 * 512 * 16);  // [32][33]
	// wave w of this half: its tiles are the half's k-th ones, k = w, w + 4, ...  (18 per half: waves 0 and 1 take five, 2 and 3 four)
	int ti[TPW], tj[TPW];
	bool on[TPW];
#pragma unroll
	for (int q = 0; q < TPW; ++q) {
		const int k = wave + 4 * q;
		on[q] = k < 18;
		tri_ride_tile(on[q] ? 2 * k + half : 0, ti[q], tj[q]);
	}
	tg_f32x16 acc[TPW];
#pragma unroll
	for (int q = 0; q < TPW; ++q)
#pragma unroll
		for (int g = 0; g < 16; ++g) acc[q][g] = 0.f;
	if (steps > 0) {
		const bf16x8* src = frags + (long)s0 * 512 + tid;
		bf16x8 v[GB][2];
		auto fetch = [&](int s, int h) { s = s < steps ? s : steps - 1; return src[(long)s * 512 + 256 * h]; };      // past the slice: a harmless re-load
#pragma unroll
		for (int d = 0; d < GB; ++d) { v[d][0] = fetch(d, 0); v[d][1] = fetch(d, 1); }
		buf[tid] = v[0][0]; buf[256 + tid] = v[0][1];
		v[0][0] = fetch(GB, 0); v[0][1] = fetch(GB, 1);
		for (int s = 0; s < steps; s += GB) {
#pragma unroll
			for (int d = 0; d < GB; ++d) {
				if (s + d < steps) {
					__syncthreads();                       // K-step s + d is in slot d & 1; the other slot has been read by everybody
					if (s + d + 1 < steps) { buf[((d + 1) & 1) * 512 + tid] = v[(d + 1) % GB][0]; buf[((d + 1) & 1) * 512 + 256 + tid] = v[(d + 1) % GB][1]; }
					v[(d + 1) % GB][0] = fetch(s + d + 1 + GB, 0); v[(d + 1) % GB][1] = fetch(s + d + 1 + GB, 1);
					const bf16x8* f = buf + (d & 1) * 512;
#pragma unroll
					for (int q = 0; q < TPW; ++q)
						if (on[q]) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ti[q] * 64 + lane], f[tj[q] * 64 + lane], acc[q], 0, 0, 0);
				}
			}
		}
	}
	float* out = partial + (long)slice * 36 * 1024;
#pragma unroll
	for (int q = 0; q < TPW; ++q)
		if (on[q]) {
			const int t = 2 * (wave + 4 * q) + half;
#pragma unroll
			for (int g = 0; g < 16; ++g) out[(long)t * 1024 + g * 64 + lane] = acc[q][g];
		}
	// release this workgroup's partial tiles, count it in; the last of the sixteen of this half finishes the half's tiles.
	// Round 5: ONE release per workgroup (every storing wave drains its stores, the barrier, then lane 0 writes the XCD's L2 back once and adds to the counter) and ONE
	// acquire by the finishing workgroup's first wave -- round 4 had all 256 threads run __threadfence() (write-back + invalidate per wave: 128 L2 write-backs per launch
	// beside the product's own stores) and every thread of the last arriver invalidate.  TRI_RIDE_FENCE_ALL=1 (A/B) restores that form.
	__shared__ unsigned s_last;
#if defined(TRI_RIDE_FENCE_ALL) && TRI_RIDE_FENCE_ALL
	__threadfence();
	__syncthreads();
	if (tid == 0) {
		const unsigned old = __hip_atomic_fetch_add(counters + half, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		s_last = old == (unsigned)(TRI_RIDE_SLICES - 1) ? 1u : 0u;
		if (s_last) __hip_atomic_store(counters + half, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the next launch finds zero)
	}
	__syncthreads();
	if (!s_last) return;
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (tid == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (hipcc may drop the wait behind the write-back when it knows the counter empty: written out)
		const unsigned old = __hip_atomic_fetch_add(counters + half, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		s_last = old == (unsigned)(TRI_RIDE_SLICES - 1) ? 1u : 0u;
		if (s_last) {
			__hip_atomic_store(counters + half, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the next launch finds zero)
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // this CU's L1 holds nothing of the partials yet that another CU has since rewritten ...
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ... once the invalidate has completed: the barrier below holds the other waves until then
		}
	}
	__syncthreads();
	if (!s_last) return;
#endif
	for (int k = 0; k < 18; ++k) {
		const int t = 2 * k + half;
		int i, j;
		tri_ride_tile(t, i, j);
		// thread tid: elements 4 tid .. 4 tid + 3 of the tile's 1 024 (register g = tid >> 4, lanes 4 (tid & 15) ..), sixteen slices in order
		const float* pp = partial + (long)t * 1024 + 4 * tid;
		f32x4 part[TRI_RIDE_SLICES];
#pragma unroll
		for (int u = 0; u < TRI_RIDE_SLICES; ++u) part[u] = *reinterpret_cast<const f32x4*>(pp + (long)u * 36 * 1024);
		f32x4 sum = part[0];
#pragma unroll
		for (int u = 1; u < TRI_RIDE_SLICES; ++u) sum += part[u];
		// C/D map of the 32 x 32 MFMA: register g of lane l is row (g & 3) + 8 (g >> 2) + 4 (l >> 5) of block i, column l & 31 of block j
		const int g = tid >> 4;
		__syncthreads();                       // (the previous tile's emit has read s_tile)
#pragma unroll
		for (int e = 0; e < 4; ++e) {
			const int l = 4 * (tid & 15) + e;
			s_tile[((g & 3) + 8 * (g >> 2) + 4 * (l >> 5)) * 33 + (l & 31)] = sum[e];
		}
		__syncthreads();
		tri_ride_emit(s_tile, i, j, G, x3, diag);
	}
}

} // namespace nmfamd
