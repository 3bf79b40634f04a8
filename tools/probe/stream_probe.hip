// stream_probe.hip -- read-only HBM streaming rates for three address-to-wave mappings (diagnostic).
//   A: chip-wide grid-stride (adjacent waves read adjacent KiB)
//   B: one contiguous region per WAVE
//   C: one contiguous region per WORKGROUP, waves interleaved by KiB
// build: hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int U>
__global__ __launch_bounds__(512, 2) void k_read(const f32x4* __restrict__ p, long n16, float* out) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const long nwaves = (long)gridDim.x * 8, gw = (long)blockIdx.x * 8 + wave;
	const long chunks = n16 / 64;   // KiB chunks
	long c0, c1, stride;
	if (MODE == 0) { c0 = gw; c1 = chunks; stride = nwaves; }
	else if (MODE == 1) { c0 = chunks * gw / nwaves; c1 = chunks * (gw + 1) / nwaves; stride = 1; }
	else { const long b0 = chunks * blockIdx.x / gridDim.x, b1 = chunks * (blockIdx.x + 1) / gridDim.x; c0 = b0 + wave; c1 = b1; stride = 8; }
	f32x4 acc = {0, 0, 0, 0};
	long c = c0;
	for (; c + (U - 1) * stride < c1; c += U * stride) {
		f32x4 v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) v[u] = p[(c + u * stride) * 64 + lane];
#pragma unroll
		for (int u = 0; u < U; ++u) acc += v[u];
	}
	for (; c < c1; c += stride) acc += p[c * 64 + lane];
	if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
template <int MODE, int U>
static void run(const char* name, const f32x4* p, long bytes, int grid, float* out) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f, sum = 0.f;
	for (int i = 0; i < 12; ++i) {
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL((k_read<MODE, U>), dim3(grid), dim3(512), 0, 0, p, bytes / 16, out);
		hipEventRecord(e1, 0); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (i >= 2) { sum += ms; if (ms < best) best = ms; }
	}
	printf("%-10s U=%d grid=%4d %7.1f MB  best %7.1f us (%5.2f TB/s)  avg %7.1f us (%5.2f TB/s)\n", name, U, grid, bytes / 1e6, best * 1e3,
	       bytes / (best * 1e-3) / 1e12, sum / 10 * 1e3, bytes / (sum / 10 * 1e-3) / 1e12);
}
int main() {
	const long cap = 1l << 30;
	f32x4* p; float* out;
	hipMalloc(&p, cap); hipMalloc(&out, 4); hipMemset(p, 0, cap);
	for (long bytes : {104l << 20, 207l << 20, 1000l << 20}) {
		for (int grid : {240, 480, 1024}) {
			run<0, 4>("gridstride", p, bytes, grid, out);
			run<1, 4>("per-wave", p, bytes, grid, out);
			run<2, 4>("per-wg", p, bytes, grid, out);
		}
		run<0, 8>("gridstride", p, bytes, 480, out);
		run<1, 8>("per-wave", p, bytes, 240, out);
		run<1, 8>("per-wave", p, bytes, 480, out);
		run<1, 16>("per-wave", p, bytes, 240, out);
	}
	return 0;
}
