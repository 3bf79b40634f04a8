// split3.h -- exact three-way bf16 splitting of fp32 values (device code shared by kernels_x3.hip and the
// update kernels that emit the split image of a factor panel).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nmfamd {

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// v = hi + mid + lo exactly (each residual of a round-to-nearest bf16 cut fits the next cut).
// Written stage by stage over the four pairs of the operand so that consecutive instructions are
// independent: a pair-by-pair chain (cvt -> widen -> subtract -> cvt ...) exposes the VALU latency of
// every step to the one wave a SIMD runs here (measured: 49 instead of 32 cycles per MFMA gap).
// Per pair: three v_cvt_pk_bf16_f32, four widenings (shift / mask), four subtractions.
// (The operand comes as eight scalars: an eight-wide vector value would be assembled in consecutive registers
// with a v_mov per element.)
__device__ inline void split3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
	u32x4 h, m, l;
	float r1[8], r2[8];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		f32x2 pr; pr[0] = v[2 * i]; pr[1] = v[2 * i + 1];
		h[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2));
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		r1[2 * i] = v[2 * i] - __builtin_bit_cast(float, h[i] << 16);
		r1[2 * i + 1] = v[2 * i + 1] - __builtin_bit_cast(float, h[i] & 0xffff0000u);
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		f32x2 pr; pr[0] = r1[2 * i]; pr[1] = r1[2 * i + 1];
		m[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2));
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		r2[2 * i] = r1[2 * i] - __builtin_bit_cast(float, m[i] << 16);
		r2[2 * i + 1] = r1[2 * i + 1] - __builtin_bit_cast(float, m[i] & 0xffff0000u);
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		f32x2 pr; pr[0] = r2[2 * i]; pr[1] = r2[2 * i + 1];
		l[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2));
	}
	hi = __builtin_bit_cast(bf16x8, h); mid = __builtin_bit_cast(bf16x8, m); lo = __builtin_bit_cast(bf16x8, l);
}

// The three 16-byte fragments of one (K-step ks, column block nb, half h, lane r) slot of the split panel image:
// Fx[(((ks * NBT + nb) * 3 + plane) * 2 + h) * 32 + r][8] = plane of F(c = 32 nb + r, y = 16 ks + 8 h + j), j = 0..7.
__device__ inline void store_split3(bf16x8* __restrict__ dst, long ks, int NBT, int nb, int h, int r, const float (&v)[8]) {
	bf16x8 hi, mid, lo;
	split3(v, hi, mid, lo);
	bf16x8* o = dst + ((ks * NBT + nb) * 3) * 64 + h * 32 + r;
	o[0] = hi; o[64] = mid; o[128] = lo;
}

// gfx950 rule (docs/DESIGN_r05.md section 11, "packed fp32 with a scalar source"): v_pk_fma / mul / add_f32 must not take a scalar register as a source --
// with waves of OTHER kernels on the same SIMD the low half of the result came out wrong in lanes 48..63 (found with tools/shared_device_diff.py;
// tools/probe/pk_scalar_probe.hip narrows it to v_pk_fma_f32 with a scalar source and an op_sel bit on a vector source, beside a wave that receives scalar-load data).
// The build disables the SLP vectoriser (which forms those instructions from uniform kernel arguments) and checks every object for such
// an instruction (build.py); a uniform value that meets a vector type in the source goes through here first.
__device__ inline float in_vgpr(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ inline double in_vgpr(double x) { return x; }

// d(c) of a pending column scale handed over as staged sums of squares (PanelTriExtras, kernels.h): the vectors are added in order, every consumer
// through this one function
__device__ inline float tri_pending_scale(const float* __restrict__ colsq, int parts, int RP, int c) {
	// (all loads of a batch of 16 in flight together: a dependent chain of `parts` round trips otherwise)
	float s = 0.f;
	for (int k0 = 0; k0 < parts; k0 += 16) {
		float v[16];
#pragma unroll
		for (int u = 0; u < 16; ++u) v[u] = k0 + u < parts ? colsq[(long)(k0 + u) * RP + c] : 0.f;
#pragma unroll
		for (int u = 0; u < 16; ++u) s += v[u];
	}
	return s > 0.f ? 1.0f / sqrtf(s) : 1.0f;
}

} // namespace nmfamd
