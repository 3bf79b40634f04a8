// Micro-probe (not part of the library): how fast can the chip re-read ONE buffer of B megabytes, launch after launch -- i.e. what a product kernel could get from
// the 256 MiB memory-side cache (and HBM behind it) if it did nothing but load.  256 workgroups x 4 waves, each wave streams a contiguous share with D 1-KiB
// loads in flight; optional "other traffic" of O megabytes (written and read by a second kernel) between two passes, as the slabs / panels of an iteration are.
//   hipcc --offload-arch=gfx950 -O3 -o mall_stream_probe mall_stream_probe.hip ; ./mall_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ src, long frags_per_wave, float* __restrict__ out) {
	__shared__ float pad[25000];          // one workgroup per CU
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const long w = (long)blockIdx.x * 4 + wave;
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
__global__ void k_other(f32x4* __restrict__ buf, long n) {
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { f32x4 v = buf[i]; v += 1.f; buf[i] = v; }
}

int main() {
	float* out; hipMalloc(&out, 256 * 256 * 4);
	f32x4* other; hipMalloc(&other, 64l << 20);
	hipMemset(other, 0, 64l << 20);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (double mb : {100.0, 150.0, 207.0, 230.0, 260.0, 414.0}) {
		const long frags_per_wave = (long)(mb * 1e6 / 16 / 1024) / 64 * 64;
		const long bytes = frags_per_wave * 16 * 1024;
		f32x4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
		for (double omb : {0.0, 10.0, 30.0}) {
			const long on = (long)(omb * 1e6 / 2 / 16);      // read + write = omb
			const int reps = 60;
			for (int i = 0; i < 10; ++i) { hipLaunchKernelGGL(k_read<14>, dim3(256), dim3(256), 0, 0, buf, frags_per_wave, out); if (on) hipLaunchKernelGGL(k_other, dim3(512), dim3(256), 0, 0, other, on); }
			float tot = 0.f;
			for (int i = 0; i < reps; ++i) {
				hipEventRecord(e0);
				hipLaunchKernelGGL(k_read<14>, dim3(256), dim3(256), 0, 0, buf, frags_per_wave, out);
				hipEventRecord(e1);
				if (on) hipLaunchKernelGGL(k_other, dim3(512), dim3(256), 0, 0, other, on);
				hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
			}
			printf("buffer %.0f MB, other traffic %.0f MB between passes: %.1f us per pass = %.2f TB/s (event pair included)\n", bytes / 1e6, omb, tot * 1e3 / reps, bytes / (tot / reps * 1e-3) / 1e12);
		}
		hipFree(buf);
	}
	return 0;
}
