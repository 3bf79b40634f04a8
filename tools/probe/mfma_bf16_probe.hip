// Micro-probe (not part of the library): sustained v_mfma_f32_32x32x16_bf16 rate and clock on all CUs,
// random operands in registers, 2 waves per SIMD, 128 accumulator registers per wave; short launches
// (the length of a real product kernel) and long ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 2) void probe(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* stamps) {
	const int tid = blockIdx.x * blockDim.x + threadIdx.x;
	bf16x8 a[4], b[2];
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)in[(tid * 64 + i * 8 + j) & 0xFFFFF];
	for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)in[(tid * 64 + 32 + i * 8 + j) & 0xFFFFF];
	unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	f32x16 acc[8];
	for (int k = 0; k < 8; ++k) for (int g = 0; g < 16; ++g) acc[k][g] = 0.f;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k & 3], b[k >> 2], acc[k], 0, 0, 0);
	}
	float sum = 0.f;
	for (int k = 0; k < 8; ++k) for (int g = 0; g < 16; ++g) sum += acc[k][g];
	unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	out[tid] = sum;
	if ((threadIdx.x & 63) == 0) { stamps[2 * (tid >> 6)] = t1 - t0; stamps[2 * (tid >> 6) + 1] = r1 - r0; }
}

int main() {
	const int blocks = 256, threads = 512;
	float *in, *out; unsigned long long* st;
	hipMalloc(&in, 4 << 20); hipMalloc(&out, blocks * threads * 4); hipMalloc(&st, blocks * 8 * 16);
	std::vector<float> h(1 << 20);
	for (auto& v : h) v = (float)rand() / RAND_MAX;
	hipMemcpy(in, h.data(), 4 << 20, hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int iters : {312, 312, 3000, 30000, 312}) {
		for (int rep = 0; rep < 3; ++rep) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, in, out, iters, st);
			hipEventRecord(e1); hipEventSynchronize(e1);
		}
		float ms; hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> hs(blocks * 8 * 2);
		hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
		double cyc = 0, ns = 0; for (size_t i = 0; i < hs.size(); i += 2) { cyc += hs[i]; ns += hs[i + 1] * 10.0; }
		double flops = (double)blocks * 8 * iters * 8 * 32768.0;
		printf("iters %5d: %.3f ms  %.1f TFLOP/s  clock %.3f GHz  cycles/iter/wave %.1f (ideal 512 at 2 waves/SIMD)\n", iters, ms, flops / ms / 1e9, cyc / ns, cyc / hs.size() * 2 / iters);
	}
	return 0;
}
