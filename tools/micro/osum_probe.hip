// Probe: ordered_sum_kernel alone on a synthetic set of extreme rows shaped like the item side of the power-law cfg3
// instance (126 rows, 5947 ... 727 entries, K=100) or its user side (214 rows of <= 2324 entries).  Tells what the
// launch costs when nothing else runs beside it, over the LDS request (waves per CU) and the grid order.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o osum_probe osum_probe.hip
#include "../../recommender-system_amd/csrc/mf_sweep.hip.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
	do {                                                                           \
		hipError_t e_ = (x);                                                       \
		if (e_ != hipSuccess) {                                                    \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
			return 1;                                                              \
		}                                                                          \
	} while (0)

// OSUM_PLAIN=1: the plain-add form (MF_OS_DPP=0 in the library) instead of the DPP broadcast form
static void (*const osum)(mf::OrderedSumArgs) = getenv("OSUM_PLAIN") ? mf::ordered_sum_kernel<false> : mf::ordered_sum_kernel<true>;

// a stand-in for the sweep running beside the ordered sums: single-wave workgroups with a large LDS tile, read 16 bytes
// per lane over and over (phase A / B of sweep_dma_kernel), plus some global traffic
__global__ void __launch_bounds__(64) hammer(double *dst, const double *src, size_t n, int rounds)
{
	extern __shared__ __attribute__((aligned(16))) char tile[];
	const int lane = threadIdx.x;
	double2 acc = make_double2(0.0, 0.0);
	for (int r = 0; r < rounds; ++r) {
		const size_t base = ((size_t) blockIdx.x * 977 + (size_t) r * 131071) % (n - 8192);
		for (int i = 0; i < 64; ++i) *reinterpret_cast<double2 *>(tile + i * 816 + 16 * (lane % 50)) = *reinterpret_cast<const double2 *>(src + base + i * 100 + 2 * (lane % 50));
		__syncthreads();
		for (int i = 0; i < 64; ++i) {
			const double2 t = *reinterpret_cast<const double2 *>(tile + i * 816 + 16 * (lane % 50));
			acc.x = acc.x + t.x;
			acc.y = acc.y + t.y;
		}
		__syncthreads();
	}
	if (acc.x == 1.2345) dst[lane] = acc.y;
}

int main(int argc, char **argv)
{
	const int K = 100, ld = 100;
	const char *shape = argc > 1 ? argv[1] : "item";
	std::vector<int> len;
	if (shape[0] == 'i')
		for (int i = 0; i < 126; ++i) len.push_back((int) (5947.0 / pow(1.0 + i, 0.4346)));
	else
		for (int i = 0; i < 214; ++i) len.push_back(i < 40 ? 2324 : (int) (2324.0 / pow(1.0 + (i - 40) / 8.0, 0.52)));
	const int nrows = (int) len.size(), nsl = (K + mf::kSliceCols - 1) / mf::kSliceCols;
	std::vector<long long> sbeg;
	std::vector<int> row;
	long long tot = 0;
	for (int i = 0; i < nrows; ++i) {
		sbeg.push_back(tot);
		row.push_back(i);
		tot += len[i];
	}
	const size_t scratch_entries = (size_t) tot + mf::kBlockEntries;
	const size_t sbytes = scratch_entries * mf::kSliceCols * nsl * 8;
	double *scratch, *Xo, *Xn;
	int *drow, *dcnt;
	long long *dsbeg;
	CK(hipMalloc(&scratch, sbytes));
	CK(hipMalloc(&Xo, (size_t) nrows * ld * 8));
	CK(hipMalloc(&Xn, (size_t) nrows * ld * 8));
	CK(hipMalloc(&drow, nrows * 4));
	CK(hipMalloc(&dcnt, nrows * 4));
	CK(hipMalloc(&dsbeg, nrows * 8));
	std::vector<double> hs(sbytes / 8), hx((size_t) nrows * ld);
	for (size_t i = 0; i < hs.size(); ++i) hs[i] = (double) ((i * 2654435761u) & 1023) * 1.0000001e-3 - 0.3;
	for (size_t i = 0; i < hx.size(); ++i) hx[i] = (double) ((i * 40503u) & 255) * 0.37;
	CK(hipMemcpy(scratch, hs.data(), sbytes, hipMemcpyHostToDevice));
	CK(hipMemcpy(Xo, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
	CK(hipMemcpy(drow, row.data(), nrows * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(dcnt, len.data(), nrows * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(dsbeg, sbeg.data(), nrows * 8, hipMemcpyHostToDevice));
	mf::OrderedSumArgs o;
	o.nrows = nrows;
	o.K = K;
	o.seed = 1;
	o.nslices = nsl;
	o.ldx = ld;
	o.row = drow;
	o.sbeg = dsbeg;
	o.cnt = dcnt;
	o.scratch = scratch;
	o.scratch_entries = scratch_entries;
	o.X_old = Xo;
	o.X_new = Xn;
	o.stamps = nullptr;
	o.max_cnt = *std::max_element(len.begin(), len.end());
	unsigned long long *dst;
	CK(hipMalloc(&dst, (size_t) nrows * nsl * 32));
	// a second buffer written between the timed launches, so the scratch is not served from L2 as a leftover
	double *flush;
	const size_t fbytes = 512u << 20;
	CK(hipMalloc(&flush, fbytes));
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	const size_t lds_list[] = {mf::kOrderedSumLds, 40000, 53000, 80000, 160000};
	for (size_t lds : lds_list) {
		CK(hipFuncSetAttribute((const void *) osum, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		float best = 1e9f, sum = 0.f;
		const int reps = 10;
		for (int r = 0; r < reps + 2; ++r) {
			CK(hipMemsetAsync(flush, r, fbytes, 0));
			CK(hipEventRecord(a, 0));
			osum<<<nrows * nsl, mf::kWave, lds, 0>>>(o);
			CK(hipEventRecord(b, 0));
			CK(hipEventSynchronize(b));
			float ms;
			CK(hipEventElapsedTime(&ms, a, b));
			if (r >= 2) {
				best = ms < best ? ms : best;
				sum += ms;
			}
		}
		printf("%s side: %d rows, %lld entries, %d waves, lds %zu: avg %.1f us, best %.1f us = %.2f TB/s of scratch\n", shape, nrows,
		       tot, nrows * nsl, lds, sum / reps * 1e3, best * 1e3, (double) tot * K * 8 / (best * 1e-3) / 1e12);
	}
	{   // against the serial sum on the host
		std::vector<double> got((size_t) nrows * ld);
		CK(hipMemcpy(got.data(), Xn, got.size() * 8, hipMemcpyDeviceToHost));
		size_t bad = 0;
		for (int i = 0; i < nrows; ++i)
			for (int k = 0; k < K; ++k) {
				double acc = hx[(size_t) i * ld + k];
				const size_t sl = k / mf::kSliceCols, c = k % mf::kSliceCols;
				for (int n = 0; n < len[i]; ++n) acc = acc + hs[((sl * scratch_entries + sbeg[i] + n) * mf::kSliceCols) + c];
				bad += memcmp(&acc, &got[(size_t) i * ld + k], 8) != 0;
			}
		printf("against the serial sum on the host: %zu of %d values differ\n", bad, nrows * K);
	}
	if (argc > 2) {   // stress: many launches beside a memory-bound kernel on a second stream, every result checked
		const int launches = atoi(argv[2]);
		std::vector<double> want((size_t) nrows * ld, 0.0), got((size_t) nrows * ld);
		for (int i = 0; i < nrows; ++i)
			for (int k = 0; k < K; ++k) {
				double acc = hx[(size_t) i * ld + k];
				const size_t sl = k / mf::kSliceCols, c = k % mf::kSliceCols;
				for (int n = 0; n < len[i]; ++n) acc = acc + hs[((sl * scratch_entries + sbeg[i] + n) * mf::kSliceCols) + c];
				want[(size_t) i * ld + k] = acc;
			}
		hipStream_t s2;
		CK(hipFuncSetAttribute((const void *) hammer, hipFuncAttributeMaxDynamicSharedMemorySize, 53248));
		CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
		CK(hipFuncSetAttribute((const void *) osum, hipFuncAttributeMaxDynamicSharedMemorySize, (int) mf::kOrderedSumLds));
		size_t bad_launches = 0, bad_values = 0;
		for (int l = 0; l < launches; ++l) {
			if (l % 2 == 0) hammer<<<4096, 64, 53248, s2>>>(flush, flush + (fbytes / 16), fbytes / 16, 40);
			CK(hipMemsetAsync(Xn, 0, (size_t) nrows * ld * 8, 0));
			osum<<<nrows * nsl, mf::kWave, mf::kOrderedSumLds, 0>>>(o);
			CK(hipMemcpy(got.data(), Xn, got.size() * 8, hipMemcpyDeviceToHost));
			size_t bad = 0;
			for (int i = 0; i < nrows; ++i)
				for (int k = 0; k < K; ++k)
					if (memcmp(&want[(size_t) i * ld + k], &got[(size_t) i * ld + k], 8) != 0) {
						if (bad_values + bad < 12) printf("  launch %d row %d (len %d) col %d: want %.17g got %.17g\n", l, i, len[i], k, want[(size_t) i * ld + k], got[(size_t) i * ld + k]);
						++bad;
					}
			bad_values += bad;
			bad_launches += bad != 0;
		}
		CK(hipDeviceSynchronize());
		printf("stress: %d launches, %zu with wrong values (%zu values)\n", launches, bad_launches, bad_values);
		return 0;
	}
	// where the time of one launch goes: clock stamps of every (row, slice) task (100 MHz wall clock)
	CK(hipFuncSetAttribute((const void *) osum, hipFuncAttributeMaxDynamicSharedMemorySize, mf::kOrderedSumLds));
	o.stamps = dst;
	CK(hipMemsetAsync(flush, 7, fbytes, 0));
	osum<<<nrows * nsl, mf::kWave, mf::kOrderedSumLds, 0>>>(o);
	CK(hipDeviceSynchronize());
	std::vector<unsigned long long> st((size_t) nrows * nsl * 4);
	CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
	unsigned long long t0 = ~0ull, t1 = 0;
	for (int i = 0; i < nrows * nsl; ++i) {
		t0 = std::min(t0, st[4 * (size_t) i]);
		t1 = std::max(t1, st[4 * (size_t) i + 2]);
	}
	printf("launch span %.1f us\n", (t1 - t0) * 0.01);
	printf("task  row  len  start_us  prologue_us  stream_us  ns_per_block  hw_id\n");
	for (int i = 0; i < nrows * nsl; i += (i < 40 ? 1 : 97)) {
		const unsigned long long *q = &st[4 * (size_t) i];
		const int li = i / nsl, nb = (len[li] + 15) / 16;
		printf("%5d %4d %5d %8.1f %8.1f %8.1f %8.0f   %08llx\n", i, li, len[li], (q[0] - t0) * 0.01, (q[1] - q[0]) * 0.01,
		       (q[2] - q[1]) * 0.01, (q[2] - q[1]) * 10.0 / nb, q[3]);
	}
	// waves resident over time (5-us bins)
	const int nb = (int) ((t1 - t0) / 500) + 1;
	std::vector<int> live(nb, 0);
	for (int i = 0; i < nrows * nsl; ++i)
		for (unsigned long long t = (st[4 * (size_t) i] - t0) / 500; t <= (st[4 * (size_t) i + 2] - t0) / 500; ++t) live[t]++;
	printf("tasks alive per 5-us bin:");
	for (int b = 0; b < nb; ++b) printf(" %d", live[b]);
	printf("\n");
	return 0;
}
