// Microbenchmark / semantics probe: where does `global_load_lds_dwordx4 v, s[base:base+1] offset:N` land in LDS on gfx950 --
// at M0 + 16*lane, or at M0 + N + 16*lane (the instruction offset added to the LDS address as well as to the global one)?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_dma_offset tools/micro/lds_dma_offset.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) probe(const double *src, double *out)
{
	__shared__ __attribute__((aligned(16))) double lds[1024];   // 8 KB
	for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.0;
	__syncthreads();
	const unsigned base = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) lds;
	const unsigned voff = threadIdx.x * 16;
	const unsigned m0 = __builtin_amdgcn_readfirstlane(base + 2048);   // LDS destination 2048 B into the array
	asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(src), "s"(m0) : "memory");
	__syncthreads();
	for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main()
{
	std::vector<double> h(4096);
	for (int i = 0; i < 4096; ++i) h[i] = i;
	double *src, *out;
	hipMalloc(&src, 4096 * 8);
	hipMalloc(&out, 1024 * 8);
	hipMemcpy(src, h.data(), 4096 * 8, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out);
	std::vector<double> o(1024);
	hipMemcpy(o.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
	int first = -1, last = -1;
	for (int i = 0; i < 1024; ++i)
		if (o[i] >= 0) {
			if (first < 0) first = i;
			last = i;
		}
	printf("M0 = array + 2048 B, offset:1024 -> LDS doubles [%d, %d] written (bytes %d..%d), first value %.0f (source double index; 128 = global offset applied)\n",
	       first, last, first * 8, last * 8 + 7, first >= 0 ? o[first] : -1.0);
	printf("%s\n", first * 8 == 2048 ? "LDS address = M0 + 16*lane: the instruction offset is NOT added to the LDS side"
	                                   : first * 8 == 3072 ? "LDS address = M0 + offset + 16*lane: the instruction offset moves BOTH sides" : "unexpected");
	return 0;
}
