// Micro-probe (not part of the library): where do the cycles of one Gauss-Jordan step go?
// Variants of the 64 x 64 fp64 register-resident inverse (k_inverse_gj64's structure), one workgroup.
//   V0 full; V1 no pivot search (p = k); V2 V1 + no reciprocal (pivinv = 1); V3 V2 + no row broadcast barrier
//   (wrong results; timing only); V4 update arithmetic only (no LDS, no barriers)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cmath>
__device__ inline unsigned wave_max_u32(unsigned v) {
	unsigned t;
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
	t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
	return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
template <int V>
__global__ __launch_bounds__(256) void gj(const float* __restrict__ A, float* __restrict__ Ainv, unsigned long long* cyc) {
	__shared__ double s_col[2][64];
	__shared__ double s_row[2][128];
	__shared__ int s_inv[64];
	const int tid = threadIdx.x, lane = tid & 63;
	const int ti = tid >> 4, tj = tid & 15;
	double M[4][8];
	for (int rr = 0; rr < 4; ++rr) for (int cc = 0; cc < 8; ++cc) {
		const int i = 4 * ti + rr, j = 8 * tj + cc;
		M[rr][cc] = j < 64 ? (double)A[j * 64 + i] : (j - 64 == i ? 1.0 : 0.0);
	}
	unsigned long long used = 0ull;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
	for (int kg = 0; kg < 8; ++kg)
#pragma unroll
	for (int kc = 0; kc < 8; ++kc) {
		const int k = 8 * kg + kc, b = kc & 1;
		int p = k; double pivinv = 1.0;
		if (V <= 3) {
			if (tj == kg) for (int rr = 0; rr < 4; ++rr) s_col[b][4 * ti + rr] = M[rr][kc];
			__syncthreads();
			if (V == 0) {
				unsigned key = 0u;
				if (!((used >> lane) & 1ull)) key = (__float_as_uint((float)fabs(s_col[b][lane])) & ~63u) | (unsigned)(63 - lane);
				key = wave_max_u32(key);
				p = 63 - (int)(key & 63u);
			}
			if (V <= 1) {
				const double piv = s_col[b][p];
				pivinv = __builtin_amdgcn_rcp(piv);
				pivinv = pivinv * (2.0 - piv * pivinv);
				pivinv = pivinv * (2.0 - piv * pivinv);
			}
			if (ti == (p >> 2)) {
				for (int rr = 0; rr < 4; ++rr) if (rr == (p & 3)) for (int cc = 0; cc < 8; ++cc) s_row[b][8 * tj + cc] = M[rr][cc] * pivinv;
			}
			if (tid == 0) s_inv[p] = k;
			used |= 1ull << p;
			if (V <= 2) __syncthreads();
		}
		double prow[8], f[4];
		if (V <= 3) { for (int cc = 0; cc < 8; ++cc) prow[cc] = s_row[b][8 * tj + cc]; for (int rr = 0; rr < 4; ++rr) f[rr] = s_col[b][4 * ti + rr]; }
		else { for (int cc = 0; cc < 8; ++cc) prow[cc] = M[0][cc] * 0.5; for (int rr = 0; rr < 4; ++rr) f[rr] = M[rr][0] * 0.25; }
		for (int rr = 0; rr < 4; ++rr) {
			const int i = 4 * ti + rr;
			for (int cc = 0; cc < 8; ++cc) M[rr][cc] = (i == p) ? prow[cc] : M[rr][cc] - f[rr] * prow[cc];
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	__syncthreads();
	if (tid == 0) cyc[0] = t1 - t0;
	if (tj >= 8) for (int rr = 0; rr < 4; ++rr) { const int kk = (V <= 3) ? s_inv[4 * ti + rr] : 4 * ti + rr; for (int cc = 0; cc < 8; ++cc) Ainv[(8 * tj + cc - 64) * 64 + (kk & 63)] = (float)M[rr][cc]; }
}

// V5: row-per-lane layout.  Wave w holds columns 32w .. 32w+31 of [A | I] for all 64 rows (lane = row), so the pivot
// search is wave-local (no LDS), one barrier per step, the pivot row is broadcast (unscaled) inside each wave, and the
// search of step k+1 is issued right after column k+1 has been updated in step k (its dependent chain overlaps the
// other 31 column updates).
__device__ inline void gj5_search(double v, unsigned long long used, int lane, int k, int b, double (*s_f)[64], double* s_pivinv, int* s_p, int* s_inv) {
	unsigned key = 0u;
	if (!((used >> lane) & 1ull)) key = (__float_as_uint((float)fabs(v)) & ~63u) | (unsigned)(63 - lane);
	key = wave_max_u32(key);
	const int p = 63 - (int)(key & 63u);
	const unsigned lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) & 0xffffffffll), p);
	const unsigned hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), p);
	const double piv = __longlong_as_double(((long long)hi << 32) | lo);
	double pivinv = __builtin_amdgcn_rcp(piv);
	pivinv = pivinv * (2.0 - piv * pivinv);
	pivinv = pivinv * (2.0 - piv * pivinv);
	s_f[b][lane] = v * pivinv;                       // multiplier of the UNSCALED pivot row
	if (lane == 0) { s_p[b] = p; s_pivinv[b] = pivinv; s_inv[p] = k; }
}
template <bool STAMP>
__global__ __launch_bounds__(256) void gj5(const float* __restrict__ A, float* __restrict__ Ainv, unsigned long long* cyc) {
	__shared__ double s_f[2][64];
	__shared__ double s_pivinv[2];
	__shared__ int s_p[2];
	__shared__ __attribute__((aligned(16))) double s_rowp[4][32];
	__shared__ int s_inv[64];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	double M[32];
#pragma unroll
	for (int q = 0; q < 32; ++q) {
		const int c = 32 * wave + q;
		M[q] = c < 64 ? (double)A[c * 64 + lane] : (c - 64 == lane ? 1.0 : 0.0);
	}
	unsigned long long used = 0ull;
	unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
	for (int kg = 0; kg < 8; ++kg) {
		const int wk = kg >> 2;
		if (wave == wk) gj5_search(M[0], used, lane, 8 * kg, 0, s_f, s_pivinv, s_p, s_inv);
#pragma unroll
		for (int kc = 0; kc < 8; ++kc) {
			const int b = kc & 1;
			const unsigned long long ta = STAMP ? __builtin_amdgcn_s_memtime() : 0;
			__syncthreads();
			const unsigned long long tb = STAMP ? __builtin_amdgcn_s_memtime() : 0;
			const double fm = s_f[b][lane];                  // f * pivinv
			const int p = __builtin_amdgcn_readfirstlane(s_p[b]);
			const double pivinv = s_pivinv[b];
			used |= 1ull << p;
			const unsigned long long tc = STAMP ? __builtin_amdgcn_s_memtime() : 0;
			// pivot row to every lane through SGPRs (v_readlane): same-address ds_read_b128 is serialised on this LDS
			double prow[32];
#pragma unroll
			for (int q = 0; q < 32; ++q) {
				const long long bits = __double_as_longlong(M[q]);
				const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), p);
				const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), p);
				prow[q] = __longlong_as_double(((long long)hi << 32) | lo);
			}
			unsigned long long td = 0;
			if (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); td = __builtin_amdgcn_s_memtime(); }
			// rows i != p: M_i -= (f_i / piv) row_p;  row p: row_p / piv = row_p - (1 - 1/piv) row_p
			const double fadj = (lane == p) ? 1.0 - pivinv : fm;
			if (kc < 7) {
				M[kc + 1] = M[kc + 1] - fadj * prow[kc + 1];
				if (wave == wk) gj5_search(M[kc + 1], used, lane, 8 * kg + kc + 1, b ^ 1, s_f, s_pivinv, s_p, s_inv);
			}
			unsigned long long te = 0;
			if (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); te = __builtin_amdgcn_s_memtime(); }
#pragma unroll
			for (int q = 0; q < 32; ++q) if (!(kc < 7 && q == kc + 1)) M[q] = M[q] - fadj * prow[q];
			if (STAMP) { a0 += tb - ta; a1 += tc - tb; a2 += td - tc; a3 += te - td; }
		}
		// rotate the register block by eight columns: the next group of pivot columns moves to M[0..7]
		double t8[8];
#pragma unroll
		for (int q = 0; q < 8; ++q) t8[q] = M[q];
#pragma unroll
		for (int q = 0; q < 24; ++q) M[q] = M[q + 8];
#pragma unroll
		for (int q = 0; q < 8; ++q) M[24 + q] = t8[q];
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	__syncthreads();
	if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = a0; cyc[2] = a1; cyc[3] = a2; cyc[4] = a3; }
	if (wave >= 2) {
		const int kk = s_inv[lane];
#pragma unroll
		for (int q = 0; q < 32; ++q) Ainv[(32 * (wave - 2) + q) * 64 + kk] = (float)M[q];
	}
}

// V6: in-place Gauss-Jordan (64 columns, no identity block), 8 waves x 8 columns, lane = row.  Step k: active wave
// k / 8, its register k % 8 is the pivot column (static index, no rotation).  Physical column k ends up holding the
// column of the inverse that belongs to pivot row p_k; physical row p_k holds row k of the inverse.
__device__ inline void gj6_search(double v, unsigned long long used, int lane, int k, int b, double (*s_f)[64], double* s_pivinv, int* s_p, int* s_rowof, int* s_pivrow) {
	unsigned key = 0u;
	if (!((used >> lane) & 1ull)) key = (__float_as_uint((float)fabs(v)) & ~63u) | (unsigned)(63 - lane);
	key = wave_max_u32(key);
	const int p = 63 - (int)(key & 63u);
	const unsigned lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) & 0xffffffffll), p);
	const unsigned hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), p);
	const double piv = __longlong_as_double(((long long)hi << 32) | lo);
	double pivinv = __builtin_amdgcn_rcp(piv);
	pivinv = pivinv * (2.0 - piv * pivinv);
	pivinv = pivinv * (2.0 - piv * pivinv);
	s_f[b][lane] = v * pivinv;
	if (lane == 0) { s_p[b] = p; s_pivinv[b] = pivinv; s_rowof[p] = k; s_pivrow[k] = p; }
}
__global__ __launch_bounds__(512) void gj6(const float* __restrict__ A, float* __restrict__ Ainv, unsigned long long* cyc) {
	__shared__ double s_f[2][64];
	__shared__ double s_pivinv[2];
	__shared__ int s_p[2];
	__shared__ int s_rowof[64], s_pivrow[64];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	double M[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) M[q] = (double)A[(8 * wave + q) * 64 + lane];
	unsigned long long used = 0ull;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	if (wave == 0) gj6_search(M[0], used, lane, 0, 0, s_f, s_pivinv, s_p, s_rowof, s_pivrow);
#pragma unroll 1
	for (int kg = 0; kg < 8; ++kg) {
#pragma unroll
		for (int kc = 0; kc < 8; ++kc) {
			const int b = kc & 1;
			__syncthreads();
			const double fm = s_f[b][lane];                  // f / piv
			const int p = __builtin_amdgcn_readfirstlane(s_p[b]);
			const double pivinv = s_pivinv[b];
			used |= 1ull << p;
			double prow[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) {
				const long long bits = __double_as_longlong(M[q]);
				const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), p);
				const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), p);
				prow[q] = __longlong_as_double(((long long)hi << 32) | lo);
			}
			const double fadj = (lane == p) ? 1.0 - pivinv : fm;
#pragma unroll
			for (int q = 0; q < 8; ++q) M[q] = M[q] - fadj * prow[q];
			if (wave == kg) M[kc] = (lane == p) ? pivinv : -fm;      // the eliminated column becomes a column of the inverse
			// search of the next step as soon as its column is final
			if (kc < 7) { if (wave == kg) gj6_search(M[kc + 1], used, lane, 8 * kg + kc + 1, b ^ 1, s_f, s_pivinv, s_p, s_rowof, s_pivrow); }
			else if (kg < 7) { if (wave == kg + 1) gj6_search(M[0], used, lane, 8 * kg + 8, b ^ 1, s_f, s_pivinv, s_p, s_rowof, s_pivrow); }
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	__syncthreads();
	if (tid == 0) cyc[0] = t1 - t0;
	{
		const int kk = s_rowof[lane];                            // this physical row is row kk of the inverse
#pragma unroll
		for (int q = 0; q < 8; ++q) Ainv[s_pivrow[8 * wave + q] * 64 + kk] = (float)M[q];
	}
}

template <int V> static void run(const char* name, const float* dA, float* dI, unsigned long long* dc) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gj<V>, dim3(1), dim3(256), 0, 0, dA, dI, dc);
	hipEventRecord(e0, 0);
	for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(gj<V>, dim3(1), dim3(256), 0, 0, dA, dI, dc);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
	printf("%-44s %7.1f us per launch, loop %8llu cycles = %6.0f per step\n", name, ms * 1e3 / 50, c, c / 64.0);
}
int main() {
	std::vector<float> h(4096);
	for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) h[j * 64 + i] = (i == j ? 70.f : 0.f) + (float)rand() / RAND_MAX;
	float *dA, *dI; unsigned long long* dc;
	hipMalloc(&dA, 16384); hipMalloc(&dI, 16384); hipMalloc(&dc, 64);
	hipMemcpy(dA, h.data(), 16384, hipMemcpyHostToDevice);
	run<0>("V0 full", dA, dI, dc);
	std::vector<float> r0(4096), r5(4096);
	hipMemcpy(r0.data(), dI, 16384, hipMemcpyDeviceToHost);
	{
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gj5<false>, dim3(1), dim3(256), 0, 0, dA, dI, dc);
		hipEventRecord(e0, 0);
		for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(gj5<false>, dim3(1), dim3(256), 0, 0, dA, dI, dc);
		hipEventRecord(e1, 0); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
		hipMemcpy(r5.data(), dI, 16384, hipMemcpyDeviceToHost);
		double md = 0, mx = 0; for (int i = 0; i < 4096; ++i) { md = fmax(md, fabs((double)r0[i] - r5[i])); mx = fmax(mx, fabs((double)r0[i])); }
		// residual of V5: || A * Ainv - I ||_max
		double res = 0; for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) { double sacc = 0; for (int t = 0; t < 64; ++t) sacc += (double)h[t * 64 + i] * r5[j * 64 + t]; res = fmax(res, fabs(sacc - (i == j))); }
		{
			hipLaunchKernelGGL(gj5<true>, dim3(1), dim3(256), 0, 0, dA, dI, dc);
			unsigned long long cs[5]; hipMemcpy(cs, dc, 40, hipMemcpyDeviceToHost);
			printf("V5 stamped (wave 0): total %llu; per step: barrier wait %.0f, read f/p %.0f, row write+read %.0f, lookahead column+search %.0f, rest (31 FMA etc.) %.0f\n",
			       cs[0], cs[1] / 64.0, cs[2] / 64.0, cs[3] / 64.0, cs[4] / 64.0, (cs[0] - cs[1] - cs[2] - cs[3] - cs[4]) / 64.0);
		}
		{
			for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gj6, dim3(1), dim3(512), 0, 0, dA, dI, dc);
			hipEvent_t f0, f1; hipEventCreate(&f0); hipEventCreate(&f1);
			hipEventRecord(f0, 0);
			for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(gj6, dim3(1), dim3(512), 0, 0, dA, dI, dc);
			hipEventRecord(f1, 0); hipEventSynchronize(f1);
			float ms6; hipEventElapsedTime(&ms6, f0, f1);
			unsigned long long c6; hipMemcpy(&c6, dc, 8, hipMemcpyDeviceToHost);
			std::vector<float> r6(4096); hipMemcpy(r6.data(), dI, 16384, hipMemcpyDeviceToHost);
			double md6 = 0, res6 = 0; for (int i = 0; i < 4096; ++i) md6 = fmax(md6, fabs((double)r0[i] - r6[i]));
			for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) { double sacc = 0; for (int t = 0; t < 64; ++t) sacc += (double)h[t * 64 + i] * r6[j * 64 + t]; res6 = fmax(res6, fabs(sacc - (i == j))); }
			printf("%-44s %7.1f us per launch, loop %8llu cycles = %6.0f per step; max |V6 - V0| = %.3g, residual %.3g\n", "V6 in-place, 8 waves x 8 columns", ms6 * 1e3 / 50, c6, c6 / 64.0, md6, res6);
		}
		printf("%-44s %7.1f us per launch, loop %8llu cycles = %6.0f per step; max |V5 - V0| = %.3g (max |V0| %.3g), residual %.3g\n", "V5 row-per-lane, one barrier per step", ms * 1e3 / 50, c, c / 64.0, md, mx, res);
	}
	run<1>("V1 no pivot search", dA, dI, dc);
	run<2>("V2 + no reciprocal", dA, dI, dc);
	run<3>("V3 + no second barrier", dA, dI, dc);
	run<4>("V4 update arithmetic only", dA, dI, dc);
	return 0;
}
