// probe_stream.hip -- how much HBM bandwidth can W waves per CU pull with D 1-KiB loads in flight each?
// One workgroup per CU (a 100 KB LDS array keeps a second one out), W waves, every wave streams its own contiguous
// share of a 2 GB buffer with a D-deep register ring of 16-byte loads.  Build: hipcc --offload-arch=gfx950 -O3 -o probe_stream probe_stream.hip
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ void k_stream(const f32x4* __restrict__ src, long frags_per_wave, float* __restrict__ out) {
	__shared__ float pad[25000];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int nw = blockDim.x >> 6;
	const long w = (long)blockIdx.x * nw + wave;
	const f32x4* p = src + w * frags_per_wave + lane;
	f32x4 ring[D];
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
	const long steps = frags_per_wave / 64;
#pragma unroll
	for (int d = 0; d < D; ++d) ring[d] = p[(long)d * 64];
	for (long s = 0; s < steps; s += D) {
#pragma unroll
		for (int d = 0; d < D; ++d) {
			acc += ring[d];
			long n = s + D + d;
			n = n < steps ? n : steps - 1;
			ring[d] = p[n * 64];
		}
	}
	if (threadIdx.x == 0) pad[0] = acc[0];
	out[(long)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + pad[threadIdx.x & 1] * 0.f;
}

// The shape of k_factor_product_bf16_r2's A path: 4 waves, each two 1-KiB loads per step into a D-deep ring, a landed step is parked in a
// three-slot LDS ring, ONE barrier per step; MODE 1: every wave also reads the 7 blocks of the step back from LDS.
template <int D, int MODE>
__global__ void k_stream_sync(const f32x4* __restrict__ src, long steps, long wg_stride, float* __restrict__ out) {
	__shared__ f32x4 l8[3 * 512];
	__shared__ float pad[18000];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const f32x4* p = src + (long)blockIdx.x * wg_stride + (2 * wave) * 64 + lane;      // step s: + s * 512 (8 KiB per step and workgroup)
	f32x4 ring[D][2];
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int d = 0; d < D; ++d) { ring[d][0] = p[(long)d * 512]; ring[d][1] = p[(long)d * 512 + 64]; }
	int wr = 0;
	for (long s = 0; s < steps; s += D) {
#pragma unroll
		for (int d = 0; d < D; ++d) {
			__syncthreads();
			l8[wr * 512 + 2 * wave * 64 + lane] = ring[d][0];
			l8[wr * 512 + 2 * wave * 64 + 64 + lane] = ring[d][1];
			long n = s + D + d;
			n = n < steps ? n : steps - 1;
			ring[d][0] = p[n * 512]; ring[d][1] = p[n * 512 + 64];
			if (MODE == 1) {
				const int rd = wr == 0 ? 2 : wr - 1;
#pragma unroll
				for (int b = 0; b < 7; ++b) acc += l8[rd * 512 + b * 64 + lane];
			}
			wr = wr == 2 ? 0 : wr + 1;
		}
	}
	if (threadIdx.x == 0) pad[0] = acc[0];
	out[(long)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + l8[threadIdx.x][0] + pad[threadIdx.x & 1] * 0.f;
}

template <int D, int MODE>
static void run_sync(const f32x4* src, long total_frags, float* out, int cus) {
	const long steps = (total_frags / cus / 512) / D * D;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int it = 0; it < 5; ++it) {
		hipEventRecord(e0, nullptr);
		hipLaunchKernelGGL((k_stream_sync<D, MODE>), dim3(cus), dim3(256), 0, nullptr, src, steps, steps * 512, out);
		hipEventRecord(e1, nullptr);
		hipEventSynchronize(e1);
		float ms = 0.f;
		hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	const double bytes = 16.0 * 512 * steps * cus;
	printf("sync form, mode %d, depth %2d: %7.1f us  %6.2f TB/s  (%.0f ns per step)\n", MODE, D, best * 1e3, bytes / (best * 1e-3) / 1e12, best * 1e6 / steps);
	fflush(stdout);
}

template <int D>
static void run(int waves, const f32x4* src, long total_frags, float* out, int cus) {
	const long per_wave = (total_frags / ((long)cus * waves)) / (64 * D) * (64 * D);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int it = 0; it < 5; ++it) {
		hipEventRecord(e0, nullptr);
		hipLaunchKernelGGL((k_stream<D>), dim3(cus), dim3(64 * waves), 0, nullptr, src, per_wave, out);
		hipEventRecord(e1, nullptr);
		hipEventSynchronize(e1);
		float ms = 0.f;
		hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	const double bytes = 16.0 * per_wave * cus * waves;
	printf("waves/CU %2d  depth %2d  (%5.1f KB in flight per CU)  %7.1f us  %6.2f TB/s\n", waves, D, waves * D * 1.0, best * 1e3, bytes / (best * 1e-3) / 1e12);
	fflush(stdout);
}

int main() {
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const long total_frags = (2l << 30) / 16;
	f32x4* src; float* out;
	hipMalloc(&src, total_frags * 16);
	hipMalloc(&out, sizeof(float) * cus * 1024);
	hipMemset(src, 0, total_frags * 16);
	run_sync<4, 0>(src, total_frags, out, cus); run_sync<8, 0>(src, total_frags, out, cus); run_sync<12, 0>(src, total_frags, out, cus);
	run_sync<4, 1>(src, total_frags, out, cus); run_sync<8, 1>(src, total_frags, out, cus); run_sync<12, 1>(src, total_frags, out, cus);
	for (int waves : {4}) {
		run<8>(waves, src, total_frags, out, cus);
		run<12>(waves, src, total_frags, out, cus);
		run<16>(waves, src, total_frags, out, cus);
		run<20>(waves, src, total_frags, out, cus);
		run<24>(waves, src, total_frags, out, cus);
		run<28>(waves, src, total_frags, out, cus);
		run<32>(waves, src, total_frags, out, cus);
		run<48>(waves, src, total_frags, out, cus);
	}
	return 0;
}
