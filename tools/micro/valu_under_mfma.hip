// Microbenchmark: what does a wave's VALU work cost while the OTHER wave of its SIMD keeps the FP64 matrix pipe busy?
// Workgroup of 8 waves: waves 0..3 (one per SIMD) loop v_mfma_f64_16x16x4_f64; waves 4..7 run a dependent chain of one
// VALU instruction kind.  Prints shader cycles per instruction of the chain with the matrix waves running and idle.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_under_mfma tools/micro/valu_under_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(512) probe(double *out, int iters, int mfma_on, unsigned long long *clk)
{
	const int wave = threadIdx.x >> 6;
	if (mfma_on == 2 && wave >= 4) __builtin_amdgcn_s_setprio(3);   // the chain wave above the matrix wave
	if (wave < 4) {
		if (!mfma_on) return;
		d4 acc[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
		double a = threadIdx.x * 1e-3, b = 1.0;
		for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
			for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
		}
		double s = 0.0;
#pragma unroll
		for (int i = 0; i < 8; ++i) s += acc[i][0];
		if (s == 12345.678) out[0] = s;
		if (blockIdx.x == 0 && threadIdx.x == 0) clk[1] = __builtin_amdgcn_s_memtime();
		return;
	}
	if (blockIdx.x == 0 && threadIdx.x == 256) clk[2] = __builtin_amdgcn_s_memtime();
	double x = threadIdx.x * 0.5, y = 1.000001;
	int xi = threadIdx.x, yi = 3;
	unsigned long long xl = threadIdx.x, sh = 1;
	double xs[16];
#pragma unroll
	for (int u = 0; u < 16; ++u) xs[u] = threadIdx.x * 0.25 + u;
	__shared__ double lds[4096];
	const double *lp = lds + (threadIdx.x & 63);
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
		if (KIND == 9) {   // 16 INDEPENDENT FP64 maxima
#pragma unroll
			for (int u = 0; u < 16; ++u) asm volatile("v_max_f64 %0, %0, %1" : "+v"(xs[u]) : "v"(y));
			continue;
		}
		if (KIND == 10) {   // 16 independent LDS reads and one wait
#pragma unroll
			for (int u = 0; u < 16; ++u) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xs[u]) : "v"((unsigned) (unsigned long long) lp), "i"(u * 512));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			continue;
		}
#pragma unroll
		for (int u = 0; u < 16; ++u) {
			if (KIND == 0) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(y));
			if (KIND == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(xi) : "v"(yi));
			if (KIND == 2) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(xl) : "v"((int) sh));
			if (KIND == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
			if (KIND == 4) asm volatile("v_cmp_gt_f64 vcc, %0, %1" ::"v"(x), "v"(y) : "vcc");
			if (KIND == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(xi) : "v"(yi));
			if (KIND == 6) asm volatile("v_cmp_gt_i32 vcc, %0, %1" ::"v"(xi), "v"(yi) : "vcc");
			if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(xi) : "v"(yi));
			if (KIND == 8) asm volatile("s_add_u32 %0, %0, 1" : "+s"(yi));
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	for (int u = 0; u < 16; ++u) x += xs[u];
	if (x == 12345.678 || xi == 123457 || xl == 987654321ull || yi == -5) out[1] = x;
	if (blockIdx.x == 0 && threadIdx.x == 256) clk[0] = t1 - t0;
}

template <int KIND>
static void run(const char *name)
{
	double *out;
	unsigned long long *clk, h[3];
	hipMalloc(&out, 16);
	hipMalloc(&clk, 24);
	const int iters = 2000;
	double r[3], m[3];
	for (int on = 0; on < 3; ++on) {
		hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, out, iters, on, clk);
		hipDeviceSynchronize();
		hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost);
		r[on] = (double) h[0] / (16.0 * iters);
		m[on] = (double) (h[1] - h[2]) / (32.0 * iters);   // matrix wave: cycles per matrix instruction, start to end
	}
	printf("%-18s cycles per instruction: matrix waves idle %.1f   running %.1f (matrix instr. %.1f)   running, chain wave at s_setprio 3: %.1f (matrix instr. %.1f)\n",
	       name, r[0], r[1], m[1], r[2], m[2]);
	hipFree(out);
	hipFree(clk);
}

int main()
{
	run<0>("v_max_f64");
	run<3>("v_add_f64");
	run<4>("v_cmp_gt_f64");
	run<2>("v_lshlrev_b64");
	run<1>("v_max_i32");
	run<5>("v_xor_b32");
	run<6>("v_cmp_gt_i32");
	run<7>("v_mov_b32");
	run<8>("s_add_u32");
	run<9>("16 indep v_max_f64");
	run<10>("16 ds_read_b64");
	return 0;
}
