// Microbenchmark: a wave streaming v_mfma_f64_16x16x4_f64 back to back starves the other wave of its SIMD (one VALU / LDS
// instruction per ~128 cycles, tools/micro/valu_under_mfma.hip).  Does pacing the stream with s_nop -- so that the wave is
// not parked at the issue stage while the matrix pipe executes -- give the other wave its issue cycles back, and at what
// cost to the matrix rate?   Workgroup of 8 waves: 0..3 matrix stream with P s_nop 15 after every instruction, 4..7 a
// dependent chain of v_max_f64 / ds_read_b64.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_pacing tools/micro/mfma_pacing.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int P, int LDSCHAIN>
__global__ void __launch_bounds__(512) probe(double *out, int iters, unsigned long long *clk)
{
	const int wave = threadIdx.x >> 6;
	__shared__ double lds[4096];
	if (wave < 4) {
		d4 acc[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
		double a = threadIdx.x * 1e-3, b = 1.0;
		const unsigned long long t0 = __builtin_amdgcn_s_memtime();
		for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
				if (P >= 1) asm volatile("s_nop 15");
				if (P >= 2) asm volatile("s_nop 15");
				if (P >= 3) asm volatile("s_nop 15");
				if (P >= 4) asm volatile("s_nop 7");
			}
		}
		const unsigned long long t1 = __builtin_amdgcn_s_memtime();
		double s = 0.0;
#pragma unroll
		for (int i = 0; i < 8; ++i) s += acc[i][0];
		if (s == 12345.678) out[0] = s;
		if (blockIdx.x == 0 && threadIdx.x == 0) clk[1] = t1 - t0;
		return;
	}
	double x = threadIdx.x * 0.5, y = 1.000001;
	const double *lp = lds + (threadIdx.x & 63);
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 16; ++u) {
			if (LDSCHAIN)
				asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"((unsigned) (unsigned long long) lp), "i"(u * 512));
			else
				asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(y));
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (x == 12345.678) out[1] = x;
	if (blockIdx.x == 0 && threadIdx.x == 256) clk[0] = t1 - t0;
}

template <int P, int LDSCHAIN>
static void run()
{
	double *out;
	unsigned long long *clk, h[2];
	hipMalloc(&out, 16);
	hipMalloc(&clk, 16);
	const int iters = 2000;
	for (int rep = 0; rep < 3; ++rep) {
		hipLaunchKernelGGL((probe<P, LDSCHAIN>), dim3(256), dim3(512), 0, 0, out, iters, clk);
		hipDeviceSynchronize();
		hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
		printf("pacing %d x s_nop  chain of %-12s: %.1f cycles per chain instruction, %.1f cycles per matrix instruction\n", P,
		       LDSCHAIN ? "ds_read_b64" : "v_max_f64", (double) h[0] / (16.0 * iters), (double) h[1] / (32.0 * iters));
	}
	hipFree(out);
	hipFree(clk);
}

int main()
{
	run<0, 0>();
	run<1, 0>();
	run<2, 0>();
	run<3, 0>();
	run<4, 0>();
	run<0, 1>();
	run<3, 1>();
	return 0;
}
