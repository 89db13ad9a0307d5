// Probe: ordered_sum_kernel alone on a synthetic set of extreme rows shaped like the item side of the power-law cfg3
// instance (126 rows, 5947 ... 727 entries, K=100) or its user side (214 rows of <= 2324 entries).  Tells what the
// launch costs when nothing else runs beside it, over the LDS request (waves per CU) and the grid order.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o osum_probe osum_probe.hip
#include "../../recommender-system_amd/csrc/mf_sweep.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
	do {                                                                           \
		hipError_t e_ = (x);                                                       \
		if (e_ != hipSuccess) {                                                    \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
			return 1;                                                              \
		}                                                                          \
	} while (0)

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
	{
		std::vector<double> h(sbytes / 8);
		for (size_t i = 0; i < h.size(); ++i) h[i] = (double) ((i * 2654435761u) & 1023) * 1e-3;
		CK(hipMemcpy(scratch, h.data(), sbytes, hipMemcpyHostToDevice));
	}
	CK(hipMemset(Xo, 0, (size_t) nrows * ld * 8));
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
	// a second buffer written between the timed launches, so the scratch is not served from L2 as a leftover
	double *flush;
	const size_t fbytes = 512u << 20;
	CK(hipMalloc(&flush, fbytes));
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	const size_t lds_list[] = {(size_t) mf::kRing * 1024, 40000, 53000, 80000, 160000};
	for (size_t lds : lds_list) {
		CK(hipFuncSetAttribute((const void *) mf::ordered_sum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		float best = 1e9f, sum = 0.f;
		const int reps = 10;
		for (int r = 0; r < reps + 2; ++r) {
			CK(hipMemsetAsync(flush, r, fbytes, 0));
			CK(hipEventRecord(a, 0));
			mf::ordered_sum_kernel<<<nrows * nsl, mf::kWave, lds, 0>>>(o);
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
	return 0;
}
