// Microbenchmark: what rate of v_mfma_f64_16x16x4_f64 does an MI355X sustain with nothing else in the loop?
// N waves per SIMD, each issuing independent matrix instructions on 8 accumulator tiles (the recommend kernels' shape).
// Prints TFLOP/s (2048 flop per instruction) and the shader clock seen by s_memtime against the 100 MHz s_memrealtime.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_peak tools/micro/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(512) mfma_loop(double *out, int iters, unsigned long long *clk)
{
	d4 acc[NACC];
#pragma unroll
	for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
	double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	double s = 0.0;
#pragma unroll
	for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	if (s == 12345.678) out[0] = s;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		clk[0] = t1 - t0;
		clk[1] = r1 - r0;
	}
}

template <int NACC>
static void run(int threads, int blocks_per_cu, int iters)
{
	double *out;
	unsigned long long *clk, h[2];
	hipMalloc(&out, 8);
	hipMalloc(&clk, 16);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int grid = 256 * blocks_per_cu;
	hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, 1000, clk);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, clk);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
	const double flop = 2048.0 * NACC * (double) iters * (threads / 64) * grid;
	printf("acc tiles %d  waves/SIMD %.1f  %.3f ms  %.2f TFLOP/s  shader clock %.0f MHz  cycles per instruction and SIMD %.1f\n", NACC,
	       threads / 64 * blocks_per_cu / 4.0, ms, flop / ms / 1e9, (double) h[0] / ((double) h[1] / 100.0),
	       (double) h[0] / ((double) NACC * iters * (threads / 64 * blocks_per_cu / 4.0)));
	hipFree(out);
	hipFree(clk);
}

int main(int argc, char **argv)
{
	const int iters = argc > 1 ? atoi(argv[1]) : 200000;
	run<8>(256, 1, iters);
	run<8>(512, 1, iters / 2);
	run<8>(256, 2, iters / 2);
	run<8>(256, 4, iters / 4);
	run<4>(256, 1, iters);
	run<4>(512, 1, iters / 2);
	run<2>(512, 1, iters / 2);
	return 0;
}
