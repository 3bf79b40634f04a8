// Micro-probe (not part of the library): the split-operand product's inner loop (operand split on the VALU interleaved with six bf16 MFMAs per
// product term, one wave per SIMD, 128 accumulator registers, no memory traffic inside the loop) in the two bf16 MFMA shapes:
//   32: v_mfma_f32_32x32x16_bf16, phase = one 32-row block x 64 columns = 12 MFMAs of 32 cycles (what kernels_x3.hip issues)
//   16: v_mfma_f32_16x16x32_bf16, phase = one 16-row block x 64 columns x 32 k = 24 MFMAs of 16 cycles
// Same FLOP, same VALU work (44 instructions per 8 split values), same registers.  The chip runs this loop far below its maximum clock (power), so what
// decides is wall time per launch on random data: back-to-back launches of the length of a real product launch.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o x3_shape_probe x3_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../nmfgpu_amd/csrc/split3.h"

using namespace nmfamd;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int VALU>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* stamps) {
	const int tid = blockIdx.x * blockDim.x + threadIdx.x;
	f32x4 va[2][8];                    // two 16-k steps (or one 32-k step) of the streamed operand: 64 floats per lane
	bf16x8 fb[4][3];                   // factor fragments of the same k range: 12 x 16 B
#pragma unroll
	for (int d = 0; d < 2; ++d)
#pragma unroll
		for (int j = 0; j < 8; ++j)
#pragma unroll
			for (int q = 0; q < 4; ++q) va[d][j][q] = in[(tid * 64 + d * 32 + j * 4 + q) & 0xFFFFF];
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int p = 0; p < 3; ++p)
#pragma unroll
			for (int j = 0; j < 8; ++j) fb[i][p][j] = (__bf16)(in[(tid * 96 + i * 24 + p * 8 + j + 77) & 0xFFFFF] * (p == 0 ? 1.f : p == 1 ? 0.004f : 0.00002f));
	float sum = 0.f;
	unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	__builtin_amdgcn_sched_barrier(0);
	if (SHAPE == 32) {
		f32x16 acc[4][2];
#pragma unroll
		for (int b = 0; b < 4; ++b)
#pragma unroll
			for (int nb = 0; nb < 2; ++nb)
#pragma unroll
				for (int g = 0; g < 16; ++g) acc[b][nb][g] = 0.f;
		bf16x8 op[2][3];
		{ float v[8];
#pragma unroll
		  for (int j = 0; j < 8; ++j) v[j] = va[0][j][0];
		  split3(v, op[0][0], op[0][1], op[0][2]); }
		for (int it = 0; it < iters; ++it) {
#pragma unroll
			for (int d = 0; d < 2; ++d) {
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const int cur = (d * 4 + b) & 1, nxt = cur ^ 1;
					const int nd = b == 3 ? (d + 1) % 2 : d, nbk = (b + 1) & 3;
					if (VALU) {
						float v[8];
#pragma unroll
						for (int j = 0; j < 8; ++j) v[j] = va[nd][j][nbk];
						split3(v, op[nxt][0], op[nxt][1], op[nxt][2]);
					}
#pragma unroll
					for (int nb = 0; nb < 2; ++nb) {
						const int f = d * 2 + nb;
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][2], fb[f][0], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][0], fb[f][2], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][1], fb[f][1], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][1], fb[f][0], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][0], fb[f][1], acc[b][nb], 0, 0, 0);
						acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[cur][0], fb[f][0], acc[b][nb], 0, 0, 0);
					}
					if (b == 3) {
						// (what the ring refill would overwrite: keeps the split inside the loop)
#pragma unroll
						for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(va[d][j]));
					}
#pragma unroll
					for (int g = 0; g < 12; ++g) {
						__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
						__builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
					}
					__builtin_amdgcn_sched_barrier(0);
				}
			}
		}
#pragma unroll
		for (int b = 0; b < 4; ++b)
#pragma unroll
			for (int nb = 0; nb < 2; ++nb)
#pragma unroll
				for (int g = 0; g < 16; ++g) sum += acc[b][nb][g];
	} else {
		// row block rb = 4 q + b (q = 0, 1: the 64-row group = ring slot q; b = element of the lane's 16-byte load), column block cb of 16
		f32x4 acc[8][4];
#pragma unroll
		for (int rb = 0; rb < 8; ++rb)
#pragma unroll
			for (int cb = 0; cb < 4; ++cb)
#pragma unroll
				for (int g = 0; g < 4; ++g) acc[rb][cb][g] = 0.f;
		bf16x8 op[2][3];
		{ float v[8];
#pragma unroll
		  for (int j = 0; j < 8; ++j) v[j] = va[0][j][0];
		  split3(v, op[0][0], op[0][1], op[0][2]); }
		for (int it = 0; it < iters; ++it) {
#pragma unroll
			for (int rb = 0; rb < 8; ++rb) {
				const int cur = rb & 1, nxt = cur ^ 1;
				const int nrb = (rb + 1) & 7;
				if (VALU) {
					float v[8];
#pragma unroll
					for (int j = 0; j < 8; ++j) v[j] = va[nrb >> 2][j][nrb & 3];
					split3(v, op[nxt][0], op[nxt][1], op[nxt][2]);
				}
#pragma unroll
				for (int cb = 0; cb < 4; ++cb) {
					acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[cur][2], fb[cb][0], acc[rb][cb], 0, 0, 0);
					acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[cur][0], fb[cb][2], acc[rb][cb], 0, 0, 0);
					acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[cur][1], fb[cb][1], acc[rb][cb], 0, 0, 0);
					acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[cur][1], fb[cb][0], acc[rb][cb], 0, 0, 0);
					acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[cur][0], fb[cb][1], acc[rb][cb], 0, 0, 0);
					acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[cur][0], fb[cb][0], acc[rb][cb], 0, 0, 0);
				}
				if (rb == 3 || rb == 7) {
#pragma unroll
					for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(va[rb >> 2][j]));
				}
#pragma unroll
				for (int g = 0; g < 24; ++g) {
					__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
					__builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
#pragma unroll
		for (int rb = 0; rb < 8; ++rb)
#pragma unroll
			for (int cb = 0; cb < 4; ++cb)
#pragma unroll
				for (int g = 0; g < 4; ++g) sum += acc[rb][cb][g];
	}
	__builtin_amdgcn_sched_barrier(0);
	unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	out[tid] = sum;
	if ((threadIdx.x & 63) == 0) { stamps[2 * (tid >> 6)] = t1 - t0; stamps[2 * (tid >> 6) + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
	const int blocks = 256, threads = 256;
	const int iters = argc > 1 ? atoi(argv[1]) : 13;          // 13 double K-steps = the 26 K-steps a wave of config 2's V H^T runs
	const int reps = argc > 2 ? atoi(argv[2]) : 400;
	float *in, *out; unsigned long long* st;
	hipMalloc(&in, 4 << 20); hipMalloc(&out, blocks * threads * 4); hipMalloc(&st, blocks * 4 * 16);
	std::vector<float> h(1 << 20);
	for (auto& v : h) v = (float)rand() / RAND_MAX;
	hipMemcpy(in, h.data(), 4 << 20, hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	auto run = [&](int shape, int valu) {
		if (shape == 32 && valu) hipLaunchKernelGGL((probe<32, 1>), dim3(blocks), dim3(threads), 0, 0, in, out, iters, st);
		else if (shape == 32) hipLaunchKernelGGL((probe<32, 0>), dim3(blocks), dim3(threads), 0, 0, in, out, iters, st);
		else if (valu) hipLaunchKernelGGL((probe<16, 1>), dim3(blocks), dim3(threads), 0, 0, in, out, iters, st);
		else hipLaunchKernelGGL((probe<16, 0>), dim3(blocks), dim3(threads), 0, 0, in, out, iters, st);
	};
	for (int round = 0; round < 3; ++round)
		for (int valu : {1, 0})
			for (int shape : {32, 16}) {
				for (int rep = 0; rep < reps / 4; ++rep) run(shape, valu);      // (warm: the clock settles over many launches)
				hipEventRecord(e0);
				for (int rep = 0; rep < reps; ++rep) run(shape, valu);
				hipEventRecord(e1); hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1);
				std::vector<unsigned long long> hs(blocks * 4 * 2);
				hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
				double cyc = 0, ns = 0; for (size_t i = 0; i < hs.size(); i += 2) { cyc += hs[i]; ns += hs[i + 1] * 10.0; }
				const double mfma_cycles = (double)iters * 3072.0;
				printf("shape %2d valu %d: %.2f us per launch, in-loop clock %.3f GHz, cycles per 32x32x16-equivalent MFMA %.1f (loop %.0f cycles, ideal %.0f)\n", shape, valu, ms * 1e3 / reps, cyc / ns,
				       cyc / (hs.size() / 2) / iters / 96.0, cyc / (hs.size() / 2), mfma_cycles);
			}
	return 0;
}
