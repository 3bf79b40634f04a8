// pk_scalar_probe.hip -- does a packed fp32 FMA with a scalar-register source return wrong low halves when waves of ANOTHER kernel
// share the SIMD?  (profiles/r03_packed_scalar_source.md: seen in the rank-256 H update; this is the attempt to see it without the library.)
// Victim: one workgroup of four waves per CU (a 100 KB LDS array keeps a second one out), every lane runs a chain of
//   v_pk_fma_f32 acc, s[a:b], acc, t  op_sel:[0,0,1] op_sel_hi:[0,1,1]      (lo = a * acc.lo + t.hi, hi = a * acc.hi + t.hi)
// next to the same chain in plain v_fma_f32 and counts the steps at which the two disagree (per lane group of 16, per half).
// A second form takes the same factors from VGPRs.  Aggressor (other stream): MFMA + LDS + VALU work in 256-thread workgroups without the LDS pad.
// Build: hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -o pk_scalar_probe pk_scalar_probe.hip   (SLP off: the reference chain must stay scalar)       Run: ./pk_scalar_probe [rounds]
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// FORM: 0 v_pk_fma_f32 with a scalar pair, 1 the same from VGPRs, 2 v_pk_mul_f32 scalar, 3 v_pk_add_f32 scalar, 4 v_fma_f64 scalar, 5 v_lshl_add_u64 scalar,
//       6 v_fma_f32 (plain, 32-bit scalar), 7 v_pk_mul_f32 with the scalar pair as SECOND source, 8 - 13 operand-select variants, 14 - 17 the scalar-operand
//       forms the shipped kernels use most (mask of a select, 64-bit multiply-add, carry chain, 32-bit multiply)
template <int FORM>
__global__ __launch_bounds__(256, 2) void k_victim(const float* __restrict__ in, float a, float b, int steps, unsigned* __restrict__ bad, float* __restrict__ out) {
	extern __shared__ float pad[];
	const int lane = threadIdx.x & 63;
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	pad[threadIdx.x] = in[2 * i];
	pad[256 + threadIdx.x] = in[2 * i + 1];
	__syncthreads();
	const uint64_t ab = (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32);
	const double dab = (double)a * 3.0 + (double)b;
	float av = a, bv = b;            // the same factors, opaque and in VGPRs: the reference side (derived from av / bv: a copy of ab or dab would pull those into VGPRs too)
	asm volatile("" : "+v"(av), "+v"(bv));
	const double dv = (double)av * 3.0 + (double)bv;
	const uint64_t abv = (uint64_t)__float_as_uint(av) | ((uint64_t)__float_as_uint(bv) << 32);
	unsigned wrong_lo = 0, wrong_hi = 0;
	float keep = 0.f;
	for (int s = 0; s < steps; ++s) {
		const float x0 = pad[(threadIdx.x + s) & 255], x1 = pad[256 + ((threadIdx.x + 3 * s) & 255)];
		const float tau = x0 * 0.25f + 0.5f;
		f32x2 acc = {x0, x1};
		f32x2 t = {0.f, bv * tau};
		float r0, r1;
		if (FORM == 0) {
			asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(ab), "v"(t));
			r0 = __builtin_fmaf(av, x0, t[1]); r1 = __builtin_fmaf(av, x1, t[1]);
		} else if (FORM == 1) {
			f32x2 avv = {av, bv};
			asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(avv), "v"(t));
			r0 = __builtin_fmaf(av, x0, t[1]); r1 = __builtin_fmaf(av, x1, t[1]);
		} else if (FORM == 2) {
			asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc) : "s"(ab));
			r0 = av * x0; r1 = bv * x1;
		} else if (FORM == 3) {
			asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc) : "s"(ab));
			r0 = av + x0; r1 = bv + x1;
		} else if (FORM == 4) {
			double d = (double)x0 + (double)x1 * 1e-3, td = (double)tau;
			const double ref = __builtin_fma(dv, d, td);
			const uint64_t db = __double_as_longlong(dab);      // (computed by the VALU: brought back into a scalar pair)
			const uint64_t dabs = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)db) | ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(db >> 32)) << 32);
			asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(d) : "s"(dabs), "v"(td));
			const uint64_t u = __double_as_longlong(d), ur = __double_as_longlong(ref);
			acc[0] = __uint_as_float((unsigned)u); acc[1] = __uint_as_float((unsigned)(u >> 32));
			r0 = __uint_as_float((unsigned)ur); r1 = __uint_as_float((unsigned)(ur >> 32));
		} else if (FORM == 5) {
			uint64_t v = ((uint64_t)__float_as_uint(x1) << 32) | __float_as_uint(x0);
			const uint64_t ref = (abv << 1) + v;
			asm volatile("v_lshl_add_u64 %0, %1, 1, %0" : "+v"(v) : "s"(ab));
			acc[0] = __uint_as_float((unsigned)v); acc[1] = __uint_as_float((unsigned)(v >> 32));
			r0 = __uint_as_float((unsigned)ref); r1 = __uint_as_float((unsigned)(ref >> 32));
		} else if (FORM == 6) {
			float y0 = x0, y1 = x1;
			asm volatile("v_fma_f32 %0, %2, %0, %3\n\tv_fma_f32 %1, %2, %1, %3" : "+v"(y0), "+v"(y1) : "s"(a), "v"(t[1]));
			acc[0] = y0; acc[1] = y1;
			r0 = __builtin_fmaf(av, x0, t[1]); r1 = __builtin_fmaf(av, x1, t[1]);
		} else if (FORM == 7) {
			asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc) : "s"(ab));
			r0 = av * x0; r1 = bv * x1;
		} else if (FORM == 8) {       // default selects: lo = a x0 + t0, hi = b x1 + t1
			t[0] = bv * tau * 0.5f;
			asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc) : "s"(ab), "v"(t));
			r0 = __builtin_fmaf(av, x0, t[0]); r1 = __builtin_fmaf(bv, x1, t[1]);
		} else if (FORM == 9) {       // the low scalar for both halves, nothing else selected
			t[0] = bv * tau * 0.5f;
			asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(ab), "v"(t));
			r0 = __builtin_fmaf(av, x0, t[0]); r1 = __builtin_fmaf(av, x1, t[1]);
		} else if (FORM == 10) {      // v_pk_mul_f32, the low scalar for both halves
			asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(acc) : "s"(ab));
			r0 = av * x0; r1 = av * x1;
		} else if (FORM == 11) {      // v_pk_add_f32, the high scalar for both halves (the update kernels' `+ eps`)
			asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0]" : "+v"(acc) : "s"(ab));
			r0 = bv + x0; r1 = bv + x1;
		} else if (FORM == 12) {      // scalar pair as the ADDEND: lo = x0 * t1 + a, hi = x1 * t1 + b
			asm volatile("v_pk_fma_f32 %0, %0, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(ab), "v"(t));
			r0 = __builtin_fmaf(x0, t[1], av); r1 = __builtin_fmaf(x1, t[1], bv);
		} else if (FORM == 14) {      // a scalar pair as the LANE MASK of a select
			const uint64_t mask = ab ^ (0x9e3779b97f4a7c15ull * (uint64_t)(s + 1));       // uniform, differs per step
			const uint64_t ms = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)mask) | ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(mask >> 32)) << 32);
			float y0;
			asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(y0) : "v"(x0), "v"(x1), "s"(ms));
			const uint64_t mv = (uint64_t)__float_as_uint(av) | ((uint64_t)__float_as_uint(bv) << 32);
			const uint64_t mref = mv ^ (0x9e3779b97f4a7c15ull * (uint64_t)(s + 1));
			acc[0] = y0; acc[1] = 0.f;
			r0 = ((mref >> lane) & 1) ? x1 : x0; r1 = 0.f;
		} else if (FORM == 15) {      // 64-bit multiply-add with a scalar factor
			uint64_t v = ((uint64_t)__float_as_uint(x1) << 32) | __float_as_uint(x0);
			const unsigned sa = __float_as_uint(a);
			const uint64_t ref = (uint64_t)__float_as_uint(av) * (uint64_t)__float_as_uint(x0) + v;
			asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "s"(sa), "v"(__float_as_uint(x0)) : "vcc");
			acc[0] = __uint_as_float((unsigned)v); acc[1] = __uint_as_float((unsigned)(v >> 32));
			r0 = __uint_as_float((unsigned)ref); r1 = __uint_as_float((unsigned)(ref >> 32));
		} else if (FORM == 16) {      // carry chain through VCC: 64-bit add of a scalar pair in two halves
			const unsigned lo = __float_as_uint(x0) | 0x80000000u, hi = __float_as_uint(x1);      // (the low add carries about half the time)
			unsigned o0, o1;
			asm volatile("v_add_co_u32_e32 %0, vcc, %2, %3\n\tv_addc_co_u32_e32 %1, vcc, %4, %5, vcc" : "=&v"(o0), "=v"(o1) : "s"((unsigned)ab), "v"(lo), "v"(__float_as_uint(bv)), "v"(hi) : "vcc");      // (VCC is the second scalar of the high add: its addend comes from a VGPR)
			const uint64_t ref = (((uint64_t)hi << 32) | lo) + abv;
			acc[0] = __uint_as_float(o0); acc[1] = __uint_as_float(o1);
			r0 = __uint_as_float((unsigned)ref); r1 = __uint_as_float((unsigned)(ref >> 32));
		} else if (FORM == 17) {      // 32-bit integer multiply with a scalar
			unsigned o0 = __float_as_uint(x0), o1 = __float_as_uint(x1);
			asm volatile("v_mul_lo_u32 %0, %2, %0\n\tv_mul_lo_u32 %1, %2, %1" : "+v"(o0), "+v"(o1) : "s"(__float_as_uint(a)));
			acc[0] = __uint_as_float(o0); acc[1] = __uint_as_float(o1);
			r0 = __uint_as_float(__float_as_uint(av) * __float_as_uint(x0)); r1 = __uint_as_float(__float_as_uint(av) * __float_as_uint(x1));
		} else {                      // (FORM == 13) scalar pair in the middle: lo = x0 * a + t1, hi = x1 * a + t1  (commuted form 0)
			asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(ab), "v"(t));
			r0 = __builtin_fmaf(x0, av, t[1]); r1 = __builtin_fmaf(x1, av, t[1]);
		}
		wrong_lo += __float_as_uint(acc[0]) != __float_as_uint(r0);
		wrong_hi += __float_as_uint(acc[1]) != __float_as_uint(r1);
		keep += acc[0] + acc[1];
	}
	if (wrong_lo) atomicAdd(&bad[(lane >> 4)], wrong_lo);
	if (wrong_hi) atomicAdd(&bad[4 + (lane >> 4)], wrong_hi);
	out[i] = keep;
}

// MFMA + LDS + packed VALU work, eight waves per workgroup, no LDS pad: lands on the victim's SIMDs
__global__ __launch_bounds__(512) void k_aggressor(const float* __restrict__ in, float* __restrict__ out, int steps, float s0, float s1) {
	__shared__ float buf[4096];
	const int tid = threadIdx.x;
	for (int k = tid; k < 4096; k += 512) buf[k] = in[k];
	__syncthreads();
	f32x16 acc;
	for (int g = 0; g < 16; ++g) acc[g] = 0.f;
	bf16x8 x, y;
	for (int j = 0; j < 8; ++j) { x[j] = (__bf16)buf[(tid + j) & 4095]; y[j] = (__bf16)buf[(tid * 3 + j) & 4095]; }
	f32x2 v = {buf[tid], buf[tid + 512]};
	for (int s = 0; s < steps; ++s) {
		acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
		const float w = buf[(tid * 7 + s) & 4095];
		v[0] = v[0] * s0 + w; v[1] = v[1] * s1 + w;
		if ((s & 63) == 63) { buf[(tid + s) & 4095] = v[0]; __syncthreads(); }
	}
	float sum = v[0] + v[1];
	for (int g = 0; g < 16; ++g) sum += acc[g];
	out[(long)blockIdx.x * 512 + tid] = sum;
}

// second aggressor: uniform (scalar) loads every step -- return data written into the SGPR file of the shared SIMD -- and scalar ALU traffic
__global__ __launch_bounds__(512) void k_aggressor_scalar(const float* __restrict__ in, float* __restrict__ out, int steps, float s0, float s1) {
	const int tid = threadIdx.x;
	float v0 = in[tid], v1 = in[tid + 512];
	f32x16 acc;
	for (int g = 0; g < 16; ++g) acc[g] = 0.f;
	bf16x8 x, y;
	for (int j = 0; j < 8; ++j) { x[j] = (__bf16)in[(tid + j) & 4095]; y[j] = (__bf16)in[(tid * 3 + j) & 4095]; }
	for (int s = 0; s < steps; ++s) {
		const int u = (s * 37 + blockIdx.x) & 4080;      // uniform index: s_load_dwordx4 and friends
		const float c0 = in[u], c1 = in[u + 1], c2 = in[u + 2], c3 = in[u + 3], c4 = in[u + 4], c5 = in[u + 5], c6 = in[u + 6], c7 = in[u + 7];
		v0 = v0 * c0 + c1; v1 = v1 * c2 + c3; v0 = v0 * c4 + c5; v1 = v1 * c6 + c7;
		if ((s & 7) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
	}
	float sum = v0 + v1;
	for (int g = 0; g < 16; ++g) sum += acc[g];
	out[(long)blockIdx.x * 512 + tid] = sum + s0 + s1;
}

int main(int argc, char** argv) {
	const int rounds = argc > 1 ? atoi(argv[1]) : 20;
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const long nv = (long)cus * 256;
	std::vector<float> h(2 * nv);
	for (long i = 0; i < 2 * nv; ++i) h[i] = 0.001f + 1e-6f * (float)(i % 9973);
	float *din, *dout, *ain, *aout;
	unsigned* dbad;
	hipMalloc(&din, sizeof(float) * 2 * nv); hipMalloc(&dout, sizeof(float) * nv); hipMalloc(&dbad, 64);
	hipMalloc(&ain, sizeof(float) * 4096); hipMalloc(&aout, sizeof(float) * 512l * cus * 8);
	hipMemcpy(din, h.data(), sizeof(float) * 2 * nv, hipMemcpyHostToDevice);
	hipMemcpy(ain, h.data(), sizeof(float) * 4096, hipMemcpyHostToDevice);
	hipStream_t sv, sa;
	hipStreamCreateWithFlags(&sv, hipStreamNonBlocking); hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
	const size_t lds = 100 * 1024;
	static const char* form_name[18] = {"v_pk_fma_f32 scalar pair", "v_pk_fma_f32 vector", "v_pk_mul_f32 scalar pair (src0)", "v_pk_add_f32 scalar pair", "v_fma_f64 scalar pair",
	                                   "v_lshl_add_u64 scalar pair", "v_fma_f32 scalar (32-bit)", "v_pk_mul_f32 scalar pair (src1)", "v_pk_fma_f32 scalar, no op_sel",
	                                    "v_pk_fma_f32 scalar, op_sel_hi src0", "v_pk_mul_f32 scalar, op_sel_hi src0", "v_pk_add_f32 scalar, op_sel src0", "v_pk_fma_f32 scalar as addend",
	                                    "v_pk_fma_f32 scalar as src1", "v_cndmask_b32 scalar-pair mask", "v_mad_u64_u32 scalar factor", "v_add_co / v_addc_co scalar + VCC", "v_mul_lo_u32 scalar"};
	static const char* agg_name[3] = {"alone", "beside the MFMA / LDS aggressor", "beside the scalar-load aggressor"};
	typedef void (*victim_t)(const float*, float, float, int, unsigned*, float*);
	victim_t victims[18] = {k_victim<0>, k_victim<1>, k_victim<2>, k_victim<3>, k_victim<4>, k_victim<5>, k_victim<6>, k_victim<7>, k_victim<8>, k_victim<9>, k_victim<10>, k_victim<11>, k_victim<12>, k_victim<13>, k_victim<14>, k_victim<15>, k_victim<16>, k_victim<17>};
	for (int form = 0; form < 18; ++form) {
		hipFuncSetAttribute(reinterpret_cast<const void*>(victims[form]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		for (int with_aggressor = (form >= 2 && form != 14 ? 2 : 0); with_aggressor < 3; ++with_aggressor) {
			hipMemset(dbad, 0, 64);
			for (int r = 0; r < rounds; ++r) {
				if (with_aggressor == 1) hipLaunchKernelGGL(k_aggressor, dim3(cus * 4), dim3(512), 0, sa, ain, aout, 4000, 0.999f, 1.001f);
				if (with_aggressor == 2) hipLaunchKernelGGL(k_aggressor_scalar, dim3(cus * 4), dim3(512), 0, sa, ain, aout, 4000, 0.999f, 1.001f);
				hipLaunchKernelGGL(victims[form], dim3(cus), dim3(256), lds, sv, din, 0.5f, 0.5f / 256.f, 20000, dbad, dout);
			}
			hipDeviceSynchronize();
			unsigned bad[8];
			hipMemcpy(bad, dbad, 32, hipMemcpyDeviceToHost);
			printf("%-34s %-34s wrong low words per lane group [0-15 16-31 32-47 48-63] = %u %u %u %u, wrong high words = %u %u %u %u  (%s)\n",
			       form_name[form], agg_name[with_aggressor], bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7], hipGetErrorString(hipGetLastError()));
		}
	}
	return 0;
}
