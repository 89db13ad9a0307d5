// Microbenchmark: latency of a dependent chain of v_add_f64 on one wave (the critical path of ordered_sum_kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void chain(double *out, const double *in, int n)
{
	double acc[CH];
	for (int c = 0; c < CH; ++c) acc[c] = in[threadIdx.x + c];
	const double x = in[100 + threadIdx.x];
	for (int i = 0; i < n; ++i) {
#pragma unroll
		for (int u = 0; u < 16; ++u)
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				acc[c] = acc[c] + x;
				asm volatile("" : "+v"(acc[c]));
			}
	}
	double s = 0;
	for (int c = 0; c < CH; ++c) s += acc[c];
	out[threadIdx.x] = s;
}
template <int CH> void run(double *out, double *in, int lanes)
{
	const int n = 200000;
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	chain<CH><<<1, lanes>>>(out, in, 1000);
	hipDeviceSynchronize();
	hipEventRecord(a);
	chain<CH><<<1, lanes>>>(out, in, n);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	printf("chains=%d lanes=%d: %.2f ns per dependent add step (%d adds in flight per step)\n", CH, lanes, ms * 1e6 / (16.0 * n), CH);
}
int main()
{
	double *in, *out;
	hipMalloc(&in, 4096); hipMalloc(&out, 4096);
	hipMemset(in, 0, 4096);
	run<1>(out, in, 64); run<2>(out, in, 64); run<4>(out, in, 64); run<8>(out, in, 64);
	run<1>(out, in, 8); run<2>(out, in, 8);
	return 0;
}
