// Micro-probe (not part of the library): fp32 MFMA shape vs sustained rate/clock on all CUs,
// random operands in registers, 2 waves per SIMD, 128 accumulator registers per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void probe(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* stamps) {
	const int tid = blockIdx.x * blockDim.x + threadIdx.x;
	float a[8], b[8];
	for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 16 + i) & 0xFFFFF]; b[i] = in[(tid * 16 + 8 + i) & 0xFFFFF]; }
	unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	float sum = 0.f;
	if (SHAPE == 32) {
		f32x16 acc[8];
		for (int k = 0; k < 8; ++k) for (int g = 0; g < 16; ++g) acc[k][g] = 0.f;
		for (int it = 0; it < iters; ++it) {
#pragma unroll
			for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k & 3], b[k >> 2], acc[k], 0, 0, 0);
			// 8 MFMAs = 8 * 4096 FLOP
		}
		for (int k = 0; k < 8; ++k) for (int g = 0; g < 16; ++g) sum += acc[k][g];
	} else {
		f32x4 acc[32];
		for (int k = 0; k < 32; ++k) for (int g = 0; g < 4; ++g) acc[k][g] = 0.f;
		for (int it = 0; it < iters; ++it) {
#pragma unroll
			for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k & 7], b[k >> 3], acc[k], 0, 0, 0);
			// 16 MFMAs = 16 * 2048 FLOP (same FLOP per iteration as the 32x32 body)
		}
		for (int k = 0; k < 32; ++k) for (int g = 0; g < 4; ++g) sum += acc[k][g];
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	out[tid] = sum;
	if ((threadIdx.x & 63) == 0) { stamps[2 * (tid >> 6)] = t1 - t0; stamps[2 * (tid >> 6) + 1] = r1 - r0; }
}

int main() {
	const int blocks = 256, threads = 512, iters = 20000;
	float *in, *out; unsigned long long* st;
	hipMalloc(&in, 4 << 20); hipMalloc(&out, blocks * threads * 4); hipMalloc(&st, blocks * 8 * 16);
	std::vector<float> h(1 << 20);
	for (auto& v : h) v = (float)rand() / RAND_MAX;
	hipMemcpy(in, h.data(), 4 << 20, hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int shape : {32, 16, 32, 16}) {
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(e0);
			if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, st);
			else hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, st);
			hipEventRecord(e1); hipEventSynchronize(e1);
		}
		float ms; hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> hs(blocks * 8 * 2);
		hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
		double cyc = 0, ns = 0; for (size_t i = 0; i < hs.size(); i += 2) { cyc += hs[i]; ns += hs[i + 1] * 10.0; }
		double flops = (double)blocks * 8 * iters * 8 * 4096.0;
		printf("shape %2d: %.3f ms  %.1f TFLOP/s  clock %.3f GHz  cycles/iter/wave %.1f\n", shape, ms, flops / ms / 1e9, cyc / ns, cyc / hs.size() * 2 / iters);
	}
	return 0;
}
