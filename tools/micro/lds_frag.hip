// Microbenchmark: cycles per ds_read_b64 for the fragment-read address patterns of recommend_mfma_kernel.
//   A  [k-pair][row] of double2, 130 rows per pair (the kernel's image): lane (lq, lr) -> pair lq>>1, half lq&1, row lr
//   B  fully contiguous (lane * 8)
//   C  [k][row] of double, 128 rows per k: lane -> k = lq, row lr
//   D  as A with 2 waves of a workgroup reading different row blocks (A and B operands alternate)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) probe(unsigned long long *out, int pattern, int reps)
{
	extern __shared__ char lds[];
	const int lane = threadIdx.x, lq = lane >> 4, lr = lane & 15;
	for (int i = lane; i < 8192; i += 64) reinterpret_cast<double *>(lds)[i] = i;
	__syncthreads();
	unsigned addr;
	if (pattern == 0) addr = (lq >> 1) * 2080 + (lq & 1) * 8 + lr * 16;
	else if (pattern == 1) addr = lane * 8;
	else addr = lq * 1024 + lr * 8;
	double acc = 0.0;
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; ++r) {
		double v0, v1, v2, v3;
		asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:256\n\tds_read_b64 %2, %4 offset:512\n\tds_read_b64 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
		             : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory");
		acc += v0 + v1 + v2 + v3;
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0) out[pattern] = t1 - t0;
	if (acc == 1.2345) out[7] = 1;
}
int main()
{
	unsigned long long *out;
	hipMalloc(&out, 64);
	const char *names[3] = {"A [k-pair][row] double2 (kernel)", "B contiguous", "C [k][row] double"};
	for (int p = 0; p < 3; ++p) {
		probe<<<1, 64, 65536>>>(out, p, 100);
		hipDeviceSynchronize();
		probe<<<1, 64, 65536>>>(out, p, 20000);
		hipDeviceSynchronize();
		unsigned long long c;
		hipMemcpy(&c, out + p, 8, hipMemcpyDeviceToHost);
		printf("%-36s %.2f clock ticks per ds_read_b64 (4 back to back + wait)\n", names[p], (double) c / (20000.0 * 4));
	}
	return 0;
}
