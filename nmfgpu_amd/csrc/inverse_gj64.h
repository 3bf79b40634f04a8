// inverse_gj64.h -- 64 x 64 Gauss-Jordan inverse as device code shared by the stand-alone kernel (kernels.hip)
// and the factor-product kernels that carry it as a passenger workgroup (kernels.hip, kernels_x3.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nmfamd {

// Fast form for r <= 64: Gauss-Jordan elimination with partial pivoting, double precision, register resident,
// no row swaps (the pivot row stays where it is and the permutation is undone when the result is written).
// Same result as the QR route (cusolver geqrf + ormqr + trsm, Matrix.h:565-618) up to rounding for
// the non-singular normal matrices the LS algorithms produce.
// maximum of a 32-bit key over the 64 lanes of a wave with DPP row shifts / row broadcasts (seven VALU ops and
// one readlane) instead of six ds_bpermute round trips
__device__ inline unsigned wave_max_u32(unsigned v) {
	unsigned t;
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:1
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:2
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:4
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:8
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;   // row_bcast:15 -> rows 1, 3
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;   // row_bcast:31 -> rows 2, 3
	return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ inline double readlane_f64(double v, int lane_uniform) {
	const long long bits = __double_as_longlong(v);
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), lane_uniform);
	const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), lane_uniform);
	return __longlong_as_double(((long long)hi << 32) | lo);
}

// Pivot search of one Gauss-Jordan step, wave-local: lane = row, v = the pivot column.  The (near-)largest |v|
// among the rows not used yet wins -- one 32-bit key per lane, the float bits of |v| with the low six bits
// replaced by 63 - lane (the first of equal maxima wins).  Publishes the multipliers f / piv, the pivot row,
// 1 / piv (hardware estimate + two Newton steps: the matrix carries fp32 data) and the bookkeeping.
__device__ inline void gj_search(double v, unsigned long long used, int lane, int k, int b,
                                 double (*s_f)[64], double* s_pivinv, int* s_p, int* s_rowof, int* s_pivrow) {
	unsigned key = 0u;
	if (!((used >> lane) & 1ull)) key = (__float_as_uint((float)fabs(v)) & ~63u) | (unsigned)(63 - lane);
	key = wave_max_u32(key);
	const int p = 63 - (int)(key & 63u);
	const double piv = readlane_f64(v, p);
	double pivinv = __builtin_amdgcn_rcp(piv);
	pivinv = pivinv * (2.0 - piv * pivinv);
	pivinv = pivinv * (2.0 - piv * pivinv);
	s_f[b][lane] = v * pivinv;
	if (lane == 0) { s_p[b] = p; s_pivinv[b] = pivinv; s_rowof[p] = k; s_pivrow[k] = p; }
}

// In-place Gauss-Jordan with partial pivoting in fp64, r <= 64, one workgroup of NW waves (8, or 4 when it rides in
// a 256-thread kernel): wave w owns columns CPW w .. CPW w + CPW - 1 (CPW = 64 / NW) of the (identity-padded)
// 64 x 64 matrix, lane = row, so
//   * the pivot search of step k is local to wave k / CPW (register k % CPW: static index, no LDS),
//   * one barrier per step carries the multipliers, the pivot row index and 1 / piv to the other waves,
//   * every wave fetches its CPW pivot-row values with v_readlane (same-address ds_read_b128 is serialised)
//     and updates CPW columns: rows i != p: M_i -= (f_i / piv) row_p; row p: row_p / piv = row_p - (1 - 1/piv) row_p,
//   * the search of step k + 1 is issued as soon as its column is final, ahead of the next barrier.
// Physical column k ends up as the column of the inverse that belongs to pivot row p_k; physical row p_k as row k.
// (Measured: 26 us against 55 us for the two-barrier, LDS-broadcast form this replaces: tools/probe/gj_probe.hip.)
template <typename T, int NW = 8>
__device__ inline void inverse_gj64_body(const T* __restrict__ A, int RP, int r, T* __restrict__ Ainv, T offdiag, T diag) {
	__shared__ double s_f[2][64];
	__shared__ double s_pivinv[2];
	__shared__ int s_p[2];
	__shared__ int s_rowof[64], s_pivrow[64];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	constexpr int CPW = 64 / NW;
	double M[CPW];
#pragma unroll
	for (int q = 0; q < CPW; ++q) {
		const int i = lane, j = CPW * wave + q;
		// the regulariser (kernel::addConstantToMatrix, KernelFillMatrix.cu:29-45) is added in T on the way in
		M[q] = (i < r && j < r) ? (double)(T)(A[(long)j * RP + i] + (i == j ? diag : offdiag)) : (i == j ? 1.0 : 0.0);
	}
	unsigned long long used = 0ull;
	if (wave == 0) gj_search(M[0], used, lane, 0, 0, s_f, s_pivinv, s_p, s_rowof, s_pivrow);
#pragma unroll 1
	for (int kg = 0; kg < NW; ++kg) {
#pragma unroll
		for (int kc = 0; kc < CPW; ++kc) {
			const int b = kc & 1;
			__syncthreads();
			const double fm = s_f[b][lane];                  // f / piv
			const int p = __builtin_amdgcn_readfirstlane(s_p[b]);
			const double pivinv = s_pivinv[b];
			used |= 1ull << p;
			double prow[CPW];
#pragma unroll
			for (int q = 0; q < CPW; ++q) prow[q] = readlane_f64(M[q], p);
			const double fadj = (lane == p) ? 1.0 - pivinv : fm;
#pragma unroll
			for (int q = 0; q < CPW; ++q) M[q] = M[q] - fadj * prow[q];
			if (wave == kg) M[kc] = (lane == p) ? pivinv : -fm;      // the eliminated column becomes a column of the inverse
			if (kc < CPW - 1) { if (wave == kg) gj_search(M[kc + 1 < CPW ? kc + 1 : 0], used, lane, CPW * kg + kc + 1, b ^ 1, s_f, s_pivinv, s_p, s_rowof, s_pivrow); }
			else if (kg < NW - 1) { if (wave == kg + 1) gj_search(M[0], used, lane, CPW * kg + CPW, b ^ 1, s_f, s_pivinv, s_p, s_rowof, s_pivrow); }
		}
	}
	__syncthreads();
	{
		const int kk = s_rowof[lane];                            // this physical row is row kk of the inverse
#pragma unroll
		for (int q = 0; q < CPW; ++q) {
			const int j = s_pivrow[CPW * wave + q];
			if (kk < r && j < r) Ainv[(long)j * RP + kk] = (T)M[q];
		}
	}
	// zero padding of the RP x RP output outside the r x r block
	for (int e = tid; e < RP * RP; e += 64 * NW) {
		const int i = e % RP, j = e / RP;
		if (i >= r || j >= r) Ainv[e] = T(0);
	}
}

} // namespace nmfamd
