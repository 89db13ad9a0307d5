// mf_hip.hip -- C ABI (include/matfact_hip.h) of the MI355X backend: plan management, CSR/CSC build,
// kernel dispatch.  HIP only -- there is no CPU compute path in this library.
#include <cstring>

#include "../../include/matfact_hip.h"
#include "mf_kernels.hip.h"

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_hip_error;

#define MF_HIP(call)                                                                        \
	do {                                                                                    \
		hipError_t _e = (call);                                                             \
		if (_e != hipSuccess) {                                                             \
			g_last_hip_error = std::string(#call) + ": " + hipGetErrorString(_e);           \
			return _e == hipErrorOutOfMemory ? MF_ERR_NO_MEMORY : MF_ERR_HIP;               \
		}                                                                                   \
	} while (0)

using SweepFn = void (*)(mf::SweepArgs);

struct SweepVariant {
	SweepFn fn;
	int kt;         // compile-time K, 0 = runtime K
	int kpmax;      // 64-column groups held in registers (register-staged form)
	int dma;        // 1: LDS-DMA form
	int row_bytes;  // LDS tile row stride in bytes (DMA form)
	int xs_bytes;   // LDS bytes in front of the tile (DMA form)
	SweepFn coop;   // row-cooperative form for tiny sweeps (compile-time-K DMA variants only)
	SweepFn prod;   // products form for segments of extreme rows (all DMA variants)
};

template <int KT, int KP>
constexpr SweepVariant variant()
{
	return SweepVariant{mf::sweep_kernel<KT, KP>, KT, KP, 0, 0, 0, nullptr, nullptr};
}

template <int KT>
constexpr SweepVariant dma_variant()
{
	return SweepVariant{mf::sweep_dma_kernel<KT, mf::DmaGeom<KT>::kPasses>, KT, 0, 1, mf::DmaGeom<KT>::kStride,
	                    mf::DmaGeom<KT>::kXsBytes, mf::sweep_coop_kernel<KT>,
	                    mf::sweep_dma_kernel<KT, mf::DmaGeom<KT>::kPasses, true>};
}

// run-time even K <= 128 * NPASS through the LDS-DMA kernel (row_bytes / xs_bytes filled in per plan)
template <int NPASS>
constexpr SweepVariant dma_generic_variant()
{
	return SweepVariant{mf::sweep_dma_kernel<0, NPASS>, 0, NPASS, 1, 0, 0, nullptr, mf::sweep_dma_kernel<0, NPASS, true>};
}

// K-specialised instances for the K of the bundled samples and of the BASELINE configs, then generic ones.
const SweepVariant kSpecialised[] = {
    variant<10, 1>(), variant<20, 1>(), variant<30, 1>(), variant<50, 1>(),
    variant<100, 2>(), variant<128, 2>(), variant<256, 4>(),
};
// LDS-DMA form: the production kernel for these (even) K
const SweepVariant kDma[] = {
    dma_variant<10>(), dma_variant<20>(), dma_variant<30>(), dma_variant<50>(),
    dma_variant<100>(), dma_variant<128>(), dma_variant<256>(),
};
const SweepVariant kDmaGeneric[] = {
    dma_generic_variant<1>(), dma_generic_variant<2>(), dma_generic_variant<4>(), dma_generic_variant<8>(),
};
const SweepVariant kGeneric[] = {
    variant<0, 1>(), variant<0, 2>(), variant<0, 4>(), variant<0, 8>(),
    variant<0, 16>(), variant<0, 32>(), variant<0, 64>(),
};

constexpr size_t kLdsPerCu = 160 * 1024;

struct TimedLaunch {
	hipEvent_t t0, t1;
	int kind;   // 0 item sweep, 1 user sweep
};

}  // namespace

struct mf_plan {
	int device = 0;
	int users_total = 0, items = 0, K = 0;
	int u0 = 0, uc = 0;
	int64_t nnz = 0;
	double alpha = 0.0;
	int flags = 0;

	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;

	// CSR over the shard's users (idx = item id) and CSC over items (idx = LOCAL user id)
	int *csr_ptr = nullptr, *csr_idx = nullptr;
	double *csr_val = nullptr;
	int *csc_ptr = nullptr, *csc_idx = nullptr;
	double *csc_val = nullptr;

	double *Lbuf[2] = {nullptr, nullptr};
	double *Rbuf[2] = {nullptr, nullptr};
	bool r_external = false;
	int cur = 0;            // generation index of the current factors
	bool have_factors = false;
	int *best_dev = nullptr;
	// MFMA recommend scratch
	double *lnorm = nullptr;
	unsigned long long *rmax_bits = nullptr;
	int *ulist = nullptr, *ucount = nullptr;
	int64_t last_uncertain = -1;   // users re-scored by the exact pass in the last recommend (-1: exact form ran)

	SweepVariant sweep{};
	int nch = 0, stride = 0;
	size_t lds_bytes = 0;
	int nch_few = 0;            // chunk size when a sweep has too few rows to fill the chip (see choose_sweep)
	size_t lds_bytes_few = 0;
	int max_row_len[2] = {0, 0}; // longest column (item sweep) / longest user row (user sweep)
	// skew-aware split of a sweep with many rows: rows whose serial walk would dominate the launch go to the
	// row-cooperative kernel on a side stream, the others stay on the single-wave kernel
	int *long_rows[2] = {nullptr, nullptr}, *short_rows[2] = {nullptr, nullptr};
	int n_long[2] = {0, 0}, n_short[2] = {0, 0};
	// extreme rows of LARGE sweeps: 256-entry segments -> scaled rows in `scratch` -> ordered sum
	int n_seg[2] = {0, 0};
	int *seg_row[2] = {nullptr, nullptr}, *seg_beg[2] = {nullptr, nullptr}, *seg_end[2] = {nullptr, nullptr};
	long long *seg_out[2] = {nullptr, nullptr}, *lr_sbeg[2] = {nullptr, nullptr};
	int *lr_cnt[2] = {nullptr, nullptr};
	double *scratch = nullptr;
	size_t scratch_entries = 0;
	// tiny sweeps (a few us of data): ONE cooperative launch over all rows; a fork/join costs more than it saves
	int nch_coop = 0;
	size_t lds_bytes_coop = 0;
	bool coop_all[2] = {false, false};
	hipStream_t side_stream = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;

	bool timing = false;
	std::vector<TimedLaunch> timed;
	int64_t acc_launch[2] = {0, 0};
	double acc_ms[2] = {0.0, 0.0};
};

namespace {

int choose_sweep(mf_plan *p)
{
	const int K = p->K;
	p->sweep = SweepVariant{nullptr, 0, 0, 0, 0, 0, nullptr, nullptr};
	const char *impl = getenv("MF_SWEEP_IMPL");   // "dma" (default) | "reg": register-staged form only
	const bool allow_dma = !(impl && strcmp(impl, "reg") == 0);
	if (allow_dma)
		for (const auto &v : kDma)
			if (v.kt == K) p->sweep = v;
	if (!p->sweep.fn && allow_dma && (K & 1) == 0)
		for (const auto &v : kDmaGeneric)
			if (K <= 128 * v.kpmax && !p->sweep.fn) {
				p->sweep = v;
				p->sweep.row_bytes = 16 * ((K / 2) | 1);
				p->sweep.xs_bytes = ((K * 8 + 255) / 256) * 256;
			}
	if (!p->sweep.fn)
		for (const auto &v : kSpecialised)
			if (v.kt == K) p->sweep = v;
	if (!p->sweep.fn)
		for (const auto &v : kGeneric)
			if (K <= v.kpmax * mf::kWave) {
				p->sweep = v;
				break;
			}
	if (!p->sweep.fn) return MF_ERR_UNSUPPORTED;

	p->stride = K | 1;
	const size_t row_bytes = p->sweep.dma ? (size_t) p->sweep.row_bytes : (size_t) p->stride * sizeof(double);
	const size_t head = p->sweep.dma ? (size_t) p->sweep.xs_bytes : 0;
	auto fit = [&](size_t budget) {
		return budget > head ? (int) std::min<size_t>(64, (budget - head) / row_bytes) : 0;
	};
	// Chunk size = latency hiding vs fixed cost.  Each single-wave workgroup alternates "gather a chunk"
	// and "compute on it", so the bytes in flight per CU come from OTHER resident workgroups: small tiles
	// (~13 KB -> ~11 workgroups per CU) beat big ones (measured on cfg4, K=100: nch 64/32/16/8 ->
	// 37.1/29.3/24.1/25.5 ms per iteration); phase A costs K steps per chunk whatever its size, which is
	// what stops the trend below ~12 entries.
	// K=256: nch 8/12/16/24 -> 71/66/78/82 ms (12 rows = 6 workgroups per CU); K=30: nch 16..32 best.
	int nch = 16;
	if (head + (size_t) nch * row_bytes > kLdsPerCu / 6) nch = std::max(12, fit(kLdsPerCu / 6));
	nch = std::min(nch, fit(kLdsPerCu));
	if (const char *env = getenv("MF_SWEEP_NCH")) {
		const int v = atoi(env);
		if (v >= 1 && v <= 64 && head + (size_t) v * row_bytes <= kLdsPerCu) nch = v;
	}
	if (nch < 1) return MF_ERR_UNSUPPORTED;
	p->nch = nch;
	p->lds_bytes = head + (size_t) nch * row_bytes;
	// A sweep over FEW rows (ML100k: 943 x 1682) cannot fill 256 CUs whatever the chunk size; its time is the
	// longest row's serial chain of chunks, so use the largest chunk there (737 entries: 47 -> 12 chunks).
	int few = std::max(nch, std::min(64, fit(kLdsPerCu / 2)));
	if (getenv("MF_SWEEP_NCH")) few = nch;
	p->nch_few = few;
	p->lds_bytes_few = head + (size_t) few * row_bytes;
	MF_HIP(hipFuncSetAttribute((const void *) p->sweep.fn, hipFuncAttributeMaxDynamicSharedMemorySize,
	                           (int) std::max(p->lds_bytes, p->lds_bytes_few)));
	return MF_OK;
}

template <typename T>
int dev_alloc(T **out, size_t count)
{
	*out = nullptr;
	MF_HIP(hipMalloc((void **) out, std::max<size_t>(count, 1) * sizeof(T)));
	return MF_OK;
}

// stable counting sort of the entries by `key` into (ptr, idx, val)
void bucket(int64_t nnz, int nkeys, const int32_t *key, int32_t key_off, const int32_t *other,
            int32_t other_off, const double *val, std::vector<int> &ptr, std::vector<int> &idx,
            std::vector<double> &v)
{
	ptr.assign((size_t) nkeys + 1, 0);
	for (int64_t n = 0; n < nnz; ++n) ptr[(size_t) (key[n] - key_off) + 1]++;
	for (int k = 0; k < nkeys; ++k) ptr[(size_t) k + 1] += ptr[k];
	std::vector<int> fill(ptr.begin(), ptr.end() - 1);
	idx.resize((size_t) nnz);
	v.resize((size_t) nnz);
	for (int64_t n = 0; n < nnz; ++n) {
		const int pos = fill[(size_t) (key[n] - key_off)]++;
		idx[(size_t) pos] = other[n] - other_off;
		v[(size_t) pos] = val[n];
	}
}


// ---- device-side CSR / CSC build: the entries are uploaded once in file order; a STABLE radix
// sort of a permutation by row (CSR) or by column (CSC) keeps the file order inside every row and column,
// which is what makes the sweeps reproduce the serial summation order.
__global__ void __launch_bounds__(256) prep_keys_kernel(const int *__restrict__ row, const int *__restrict__ col,
                                                        int64_t nnz, int u0, int uc, int items,
                                                        unsigned *__restrict__ rkey, unsigned *__restrict__ perm,
                                                        int *__restrict__ flags)
{
	const int64_t n = (int64_t) blockIdx.x * 256 + threadIdx.x;
	if (n >= nnz) return;
	const int r = row[n] - u0, c = col[n];
	if (r < 0 || r >= uc || c < 0 || c >= items) atomicOr(&flags[0], 1);         // out of range
	if (n > 0 && row[n - 1] > row[n]) atomicOr(&flags[1], 1);                      // not row-sorted
	rkey[n] = (unsigned) r;
	perm[n] = (unsigned) n;
}

__global__ void __launch_bounds__(256) gather_kernel(const unsigned *__restrict__ perm, int64_t nnz,
                                                     const int *__restrict__ other, int other_off,
                                                     const double *__restrict__ val, int *__restrict__ idx_out,
                                                     double *__restrict__ val_out)
{
	const int64_t n = (int64_t) blockIdx.x * 256 + threadIdx.x;
	if (n >= nnz) return;
	const unsigned s = perm[n];
	idx_out[n] = other[s] - other_off;
	val_out[n] = val[s];
}

// ptr[k] = first position whose (sorted) key is >= k, k = 0..nkeys
__global__ void __launch_bounds__(256) ptr_kernel(const unsigned *__restrict__ sorted, int64_t nnz, int nkeys,
                                                  int *__restrict__ ptr)
{
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k > nkeys) return;
	int64_t lo = 0, hi = nnz;
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if (sorted[mid] < (unsigned) k) lo = mid + 1; else hi = mid;
	}
	ptr[k] = (int) lo;
}

__global__ void __launch_bounds__(256) copy_keys_kernel(const int *__restrict__ src, int64_t nnz,
                                                        unsigned *__restrict__ key, unsigned *__restrict__ perm)
{
	const int64_t n = (int64_t) blockIdx.x * 256 + threadIdx.x;
	if (n >= nnz) return;
	key[n] = (unsigned) src[n];
	perm[n] = (unsigned) n;
}

int bits_for(int nkeys)
{
	int b = 1;
	while (b < 32 && (1ll << b) < (long long) nkeys) ++b;
	return b;
}

struct DevTmp {   // frees its buffers on scope exit
	std::vector<void *> bufs;
	~DevTmp() { for (void *b : bufs) (void) hipFree(b); }
	template <typename T> int get(T **out, size_t count)
	{
		const int rc = dev_alloc(out, count);
		if (rc == MF_OK) bufs.push_back(*out);
		return rc;
	}
};

// Builds csr_* and csc_* of plan p from host SoA entries.  Returns MF_ERR_ARGUMENT for out-of-range indices.
int build_on_device(mf_plan *p, const mf_shard *s, std::vector<int> &csr_ptr_host, std::vector<int> &csc_ptr_host)
{
	const int64_t nnz = s->nnz;
	const size_t nz = (size_t) nnz;
	hipStream_t st = p->stream;
	MF_HIP(dev_alloc(&p->csr_ptr, (size_t) p->uc + 1) == MF_OK ? hipSuccess : hipErrorOutOfMemory);
	MF_HIP(dev_alloc(&p->csc_ptr, (size_t) p->items + 1) == MF_OK ? hipSuccess : hipErrorOutOfMemory);
	MF_HIP(dev_alloc(&p->csr_idx, nz + 64) == MF_OK ? hipSuccess : hipErrorOutOfMemory);
	MF_HIP(dev_alloc(&p->csr_val, nz + 64) == MF_OK ? hipSuccess : hipErrorOutOfMemory);
	MF_HIP(dev_alloc(&p->csc_idx, nz + 64) == MF_OK ? hipSuccess : hipErrorOutOfMemory);
	MF_HIP(dev_alloc(&p->csc_val, nz + 64) == MF_OK ? hipSuccess : hipErrorOutOfMemory);
	csr_ptr_host.assign((size_t) p->uc + 1, 0);
	csc_ptr_host.assign((size_t) p->items + 1, 0);
	if (nnz == 0) {
		MF_HIP(hipMemsetAsync(p->csr_ptr, 0, ((size_t) p->uc + 1) * sizeof(int), st));
		MF_HIP(hipMemsetAsync(p->csc_ptr, 0, ((size_t) p->items + 1) * sizeof(int), st));
		MF_HIP(hipStreamSynchronize(st));
		return MF_OK;
	}
	DevTmp tmp;
	int *d_row = nullptr, *d_col = nullptr, *d_flags = nullptr;
	unsigned *key_in = nullptr, *key_out = nullptr, *perm_in = nullptr, *perm_out = nullptr;
	int rc;
	if ((rc = tmp.get(&d_row, nz)) != MF_OK || (rc = tmp.get(&d_col, nz)) != MF_OK ||
	    (rc = tmp.get(&key_in, nz)) != MF_OK || (rc = tmp.get(&key_out, nz)) != MF_OK ||
	    (rc = tmp.get(&perm_in, nz)) != MF_OK || (rc = tmp.get(&perm_out, nz)) != MF_OK ||
	    (rc = tmp.get(&d_flags, 2)) != MF_OK)
		return rc;
	// the values land directly in csr_val when the input is row-sorted (the usual case); otherwise csc_val is
	// used as the staging copy of the file-order values and overwritten last
	double *d_val = p->csc_val;
	MF_HIP(hipMemcpyAsync(d_row, s->row, nz * sizeof(int), hipMemcpyHostToDevice, st));
	MF_HIP(hipMemcpyAsync(d_col, s->col, nz * sizeof(int), hipMemcpyHostToDevice, st));
	MF_HIP(hipMemcpyAsync(p->csr_val, s->val, nz * sizeof(double), hipMemcpyHostToDevice, st));
	MF_HIP(hipMemsetAsync(d_flags, 0, 2 * sizeof(int), st));
	const unsigned grid = (unsigned) ((nnz + 255) / 256);
	hipLaunchKernelGGL(prep_keys_kernel, dim3(grid), dim3(256), 0, st, d_row, d_col, nnz, p->u0, p->uc, p->items,
	                   key_in, perm_in, d_flags);
	int flags[2] = {0, 0};
	MF_HIP(hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, st));
	MF_HIP(hipStreamSynchronize(st));
	if (flags[0]) return MF_ERR_ARGUMENT;
	const bool row_sorted = flags[1] == 0;

	size_t temp_bytes = 0, need = 0;
	MF_HIP(rocprim::radix_sort_pairs(nullptr, need, key_in, key_out, perm_in, perm_out, nz, 0, bits_for(p->uc), st));
	temp_bytes = need;
	MF_HIP(rocprim::radix_sort_pairs(nullptr, need, key_in, key_out, perm_in, perm_out, nz, 0, bits_for(p->items), st));
	temp_bytes = std::max(temp_bytes, need);
	void *d_temp = nullptr;
	if ((rc = tmp.get((char **) &d_temp, temp_bytes)) != MF_OK) return rc;

	const double *vals_file_order = p->csr_val;   // file-order values currently live here
	if (row_sorted) {
		// CSR == file order: idx = col, val = val (already in place), ptr from the row keys
		MF_HIP(hipMemcpyAsync(p->csr_idx, d_col, nz * sizeof(int), hipMemcpyDeviceToDevice, st));
		hipLaunchKernelGGL(ptr_kernel, dim3((unsigned) ((p->uc + 256) / 256)), dim3(256), 0, st, key_in, nnz, p->uc,
		                   p->csr_ptr);
	} else {
		// keep a file-order copy of the values, then permute into csr_val
		MF_HIP(hipMemcpyAsync(d_val, p->csr_val, nz * sizeof(double), hipMemcpyDeviceToDevice, st));
		vals_file_order = d_val;
		MF_HIP(rocprim::radix_sort_pairs(d_temp, temp_bytes, key_in, key_out, perm_in, perm_out, nz, 0,
		                                 bits_for(p->uc), st));
		hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, st, perm_out, nnz, d_col, 0, vals_file_order,
		                   p->csr_idx, p->csr_val);
		hipLaunchKernelGGL(ptr_kernel, dim3((unsigned) ((p->uc + 256) / 256)), dim3(256), 0, st, key_out, nnz, p->uc,
		                   p->csr_ptr);
	}
	// CSC: stable sort of the file order by column
	hipLaunchKernelGGL(copy_keys_kernel, dim3(grid), dim3(256), 0, st, d_col, nnz, key_in, perm_in);
	MF_HIP(rocprim::radix_sort_pairs(d_temp, temp_bytes, key_in, key_out, perm_in, perm_out, nz, 0,
	                                 bits_for(p->items), st));
	if (row_sorted) {
		hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, st, perm_out, nnz, d_row, p->u0, vals_file_order,
		                   p->csc_idx, p->csc_val);
	} else {
		// vals_file_order aliases csc_val: gather into a temporary, then copy
		double *d_val2 = nullptr;
		if ((rc = tmp.get(&d_val2, nz)) != MF_OK) return rc;
		hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, st, perm_out, nnz, d_row, p->u0, vals_file_order,
		                   p->csc_idx, d_val2);
		MF_HIP(hipMemcpyAsync(p->csc_val, d_val2, nz * sizeof(double), hipMemcpyDeviceToDevice, st));
	}
	hipLaunchKernelGGL(ptr_kernel, dim3((unsigned) ((p->items + 256) / 256)), dim3(256), 0, st, key_out, nnz, p->items,
	                   p->csc_ptr);
	MF_HIP(hipGetLastError());
	MF_HIP(hipMemcpyAsync(csr_ptr_host.data(), p->csr_ptr, ((size_t) p->uc + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
	MF_HIP(hipMemcpyAsync(csc_ptr_host.data(), p->csc_ptr, ((size_t) p->items + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
	MF_HIP(hipStreamSynchronize(st));
	return MF_OK;
}

int launch_sweep(mf_plan *p, int kind, int seed)
{
	mf::SweepArgs a;
	a.K = p->K;
	a.nch = p->nch;
	a.stride = p->stride;
	a.seed = seed;
	a.c2 = p->alpha * 2;
	const int nxt = p->cur ^ 1;
	if (kind == 0) {   // item sweep: X = R, Y = L, CSC
		a.nrows = p->items;
		a.ptr = p->csc_ptr;
		a.idx = p->csc_idx;
		a.val = p->csc_val;
		a.X_old = p->Rbuf[p->cur];
		a.Y_old = p->Lbuf[p->cur];
		a.X_new = p->Rbuf[nxt];
	} else {           // user sweep: X = L, Y = R, CSR
		a.nrows = p->uc;
		a.ptr = p->csr_ptr;
		a.idx = p->csr_idx;
		a.val = p->csr_val;
		a.X_old = p->Lbuf[p->cur];
		a.Y_old = p->Rbuf[p->cur];
		a.X_new = p->Lbuf[nxt];
	}
	a.rowlist = nullptr;
	a.seg_row = a.seg_beg = a.seg_end = nullptr;
	a.seg_out = nullptr;
	a.scratch = nullptr;
	a.scratch_entries = 0;
	if (a.nrows <= 0) return MF_OK;
	const bool few_rows = a.nrows < 4096;
	const bool coop = p->coop_all[kind];
	if (few_rows) a.nch = coop ? p->nch_coop : p->nch_few;
	const size_t lds = coop ? p->lds_bytes_coop : (few_rows ? p->lds_bytes_few : p->lds_bytes);
	const SweepFn fn = coop ? p->sweep.coop : p->sweep.fn;
	const int block = coop ? mf::kCoopWaves * mf::kWave : mf::kWave;
	const int grid = std::min(a.nrows, 1 << 20);
	TimedLaunch t{};
	if (p->timing) {
		MF_HIP(hipEventCreate(&t.t0));
		MF_HIP(hipEventCreate(&t.t1));
		t.kind = kind;
		MF_HIP(hipEventRecord(t.t0, p->stream));
	}
	void *args[] = {&a};
	if (p->n_long[kind] > 0) {
		// extreme rows on the side stream, concurrently with the other rows on the main stream:
		//   products kernel over their 256-entry segments -> ordered sum per (row, 16-column slice)
		mf::SweepArgs b = a;
		b.nrows = p->n_seg[kind];
		b.rowlist = nullptr;
		b.nch = p->nch;
		b.seg_row = p->seg_row[kind];
		b.seg_beg = p->seg_beg[kind];
		b.seg_end = p->seg_end[kind];
		b.seg_out = p->seg_out[kind];
		b.scratch = p->scratch;
		b.scratch_entries = p->scratch_entries;
		void *bargs[] = {&b};
		mf::OrderedSumArgs o;
		o.nrows = p->n_long[kind];
		o.K = p->K;
		o.seed = seed;
		o.nslices = (p->K + 15) / 16;
		o.row = p->long_rows[kind];
		o.sbeg = p->lr_sbeg[kind];
		o.cnt = p->lr_cnt[kind];
		o.scratch = p->scratch;
		o.scratch_entries = p->scratch_entries;
		o.X_old = a.X_old;
		o.X_new = a.X_new;
		void *oargs[] = {&o};
		MF_HIP(hipEventRecord(p->ev_fork, p->stream));
		MF_HIP(hipStreamWaitEvent(p->side_stream, p->ev_fork, 0));
		MF_HIP(hipLaunchKernel((const void *) p->sweep.prod, dim3(b.nrows), dim3(mf::kWave), bargs, p->lds_bytes,
		                       p->side_stream));
		MF_HIP(hipLaunchKernel((const void *) mf::ordered_sum_kernel, dim3(o.nrows * o.nslices), dim3(mf::kWave), oargs,
		                       0, p->side_stream));
		MF_HIP(hipEventRecord(p->ev_join, p->side_stream));
		a.nrows = p->n_short[kind];
		a.rowlist = p->short_rows[kind];
		a.nch = p->nch;   // the extreme rows are gone: the occupancy-friendly chunk size is right again
		if (a.nrows > 0)
			MF_HIP(hipLaunchKernel((const void *) p->sweep.fn, dim3(std::min(a.nrows, 1 << 20)), dim3(mf::kWave), args,
			                       p->lds_bytes, p->stream));
		MF_HIP(hipStreamWaitEvent(p->stream, p->ev_join, 0));
	} else {
		MF_HIP(hipLaunchKernel((const void *) fn, dim3(grid), dim3(block), args, lds, p->stream));
	}
	if (p->timing) {
		MF_HIP(hipEventRecord(t.t1, p->stream));
		p->timed.push_back(t);
	}
	return MF_OK;
}

int drain_timing(mf_plan *p)
{
	for (auto &t : p->timed) {
		MF_HIP(hipEventSynchronize(t.t1));
		float ms = 0.f;
		MF_HIP(hipEventElapsedTime(&ms, t.t0, t.t1));
		p->acc_launch[t.kind]++;
		p->acc_ms[t.kind] += ms;
		(void) hipEventDestroy(t.t0);
		(void) hipEventDestroy(t.t1);
	}
	p->timed.clear();
	return MF_OK;
}

}  // namespace

extern "C" {

const char *mf_backend_strerror(int status)
{
	switch (status) {
	case MF_OK: return "ok";
	case MF_ERR_ARGUMENT: return "invalid argument";
	case MF_ERR_NO_DEVICE: return "no usable HIP device";
	case MF_ERR_HIP: return "HIP runtime error";
	case MF_ERR_NO_MEMORY: return "out of memory";
	case MF_ERR_UNSUPPORTED: return "unsupported shape";
	case MF_ERR_STATE: return "plan is not in a state that allows this call";
	default: return "unknown status";
	}
}

const char *mf_backend_last_hip_error(void) { return g_last_hip_error.c_str(); }

int mf_backend_abi_version(void) { return MATFACT_HIP_ABI_VERSION; }

int mf_backend_device_count(void)
{
	int n = 0;
	const hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) {
		g_last_hip_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
		return e == hipErrorNoDevice ? 0 : MF_ERR_NO_DEVICE;
	}
	return n;
}

int mf_plan_create(mf_plan **out, const mf_shard *s)
{
	if (!out) return MF_ERR_ARGUMENT;
	*out = nullptr;
	if (!s || s->users_total < 0 || s->items < 0 || s->features < 1 || s->nnz < 0 || s->user_begin < 0 ||
	    s->user_count < 0 || (int64_t) s->user_begin + s->user_count > s->users_total ||
	    s->nnz > INT32_MAX - 64 || (s->nnz > 0 && (!s->row || !s->col || !s->val)))
		return MF_ERR_ARGUMENT;
	const int ndev = mf_backend_device_count();
	if (ndev <= 0 || s->device < 0 || s->device >= ndev) return MF_ERR_NO_DEVICE;
	MF_HIP(hipSetDevice(s->device));

	mf_plan *p = new (std::nothrow) mf_plan();
	if (!p) return MF_ERR_NO_MEMORY;
	p->device = s->device;
	p->users_total = s->users_total;
	p->items = s->items;
	p->K = s->features;
	p->u0 = s->user_begin;
	p->uc = s->user_count;
	p->nnz = s->nnz;
	p->alpha = s->alpha;
	p->flags = s->flags;

	int rc = choose_sweep(p);
	auto fail = [&](int code) {
		mf_plan_destroy(p);
		return code;
	};
	if (rc != MF_OK) return fail(rc);

	if (hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(MF_ERR_HIP);
	p->stream = p->own_stream;

#define MF_TRY(x)                       \
	do {                                \
		int _rc = (x);                  \
		if (_rc != MF_OK) return fail(_rc); \
	} while (0)
#define MF_TRY_HIP(call)                                                              \
	do {                                                                              \
		hipError_t _e = (call);                                                       \
		if (_e != hipSuccess) {                                                       \
			g_last_hip_error = std::string(#call) + ": " + hipGetErrorString(_e);     \
			return fail(_e == hipErrorOutOfMemory ? MF_ERR_NO_MEMORY : MF_ERR_HIP);   \
		}                                                                             \
	} while (0)
	{
		const char *where = getenv("MF_BUILD");   // "device" (default) | "host": CSR/CSC bucketing on the CPU
		std::vector<int> rptr, cptr;
		if (where && strcmp(where, "host") == 0) {
			for (int64_t n = 0; n < s->nnz; ++n)
				if (s->row[n] < s->user_begin || s->row[n] >= s->user_begin + s->user_count || s->col[n] < 0 ||
				    s->col[n] >= s->items)
					return fail(MF_ERR_ARGUMENT);
			std::vector<int> idx;
			std::vector<double> val;
			const size_t nz = (size_t) s->nnz;
			try {
				bucket(s->nnz, p->uc, s->row, p->u0, s->col, 0, s->val, rptr, idx, val);
			} catch (const std::bad_alloc &) {
				return fail(MF_ERR_NO_MEMORY);
			}
			MF_TRY(dev_alloc(&p->csr_ptr, (size_t) p->uc + 1));
			MF_TRY(dev_alloc(&p->csr_idx, nz + 64));
			MF_TRY(dev_alloc(&p->csr_val, nz + 64));
			MF_TRY_HIP(hipMemcpy(p->csr_ptr, rptr.data(), ((size_t) p->uc + 1) * sizeof(int), hipMemcpyHostToDevice));
			if (nz) {
				MF_TRY_HIP(hipMemcpy(p->csr_idx, idx.data(), nz * sizeof(int), hipMemcpyHostToDevice));
				MF_TRY_HIP(hipMemcpy(p->csr_val, val.data(), nz * sizeof(double), hipMemcpyHostToDevice));
			}
			try {
				bucket(s->nnz, p->items, s->col, 0, s->row, p->u0, s->val, cptr, idx, val);
			} catch (const std::bad_alloc &) {
				return fail(MF_ERR_NO_MEMORY);
			}
			MF_TRY(dev_alloc(&p->csc_ptr, (size_t) p->items + 1));
			MF_TRY(dev_alloc(&p->csc_idx, nz + 64));
			MF_TRY(dev_alloc(&p->csc_val, nz + 64));
			MF_TRY_HIP(hipMemcpy(p->csc_ptr, cptr.data(), ((size_t) p->items + 1) * sizeof(int), hipMemcpyHostToDevice));
			if (nz) {
				MF_TRY_HIP(hipMemcpy(p->csc_idx, idx.data(), nz * sizeof(int), hipMemcpyHostToDevice));
				MF_TRY_HIP(hipMemcpy(p->csc_val, val.data(), nz * sizeof(double), hipMemcpyHostToDevice));
			}
		} else {
			MF_TRY(build_on_device(p, s, rptr, cptr));
		}
		for (int u = 0; u < p->uc; ++u) p->max_row_len[1] = std::max(p->max_row_len[1], rptr[(size_t) u + 1] - rptr[u]);
		for (int j = 0; j < p->items; ++j) p->max_row_len[0] = std::max(p->max_row_len[0], cptr[(size_t) j + 1] - cptr[j]);
		// ---- long / short row lists.  A row is "long" when its serial walk (~0.075 us per entry at 16-entry
		// chunks) would exceed roughly a quarter of the bandwidth time of the whole sweep (nnz * 8K bytes at
		// ~7 TB/s): len > 4e-6 * nnz * K, and never below 128 entries.  cfg4 has none; a power-law instance a few.
		const char *skew_env = getenv("MF_SWEEP_SKEW");   // "0" disables the split
		if (p->sweep.prod && !(skew_env && skew_env[0] == '0')) {
			const size_t per_entry = 2 * (size_t) mf::kCoopProducers * (size_t) p->sweep.row_bytes;
			const size_t head = (size_t) p->sweep.xs_bytes;
			int nl = (int) std::min<size_t>(32, (kLdsPerCu - 4096 - head) / per_entry);
			if (const char *env = getenv("MF_SWEEP_NCH")) {
				const int v = atoi(env);
				if (v >= 1 && v <= 64 && head + (size_t) v * per_entry <= kLdsPerCu) nl = v;
			}
			double thr = 4e-6 * (double) p->nnz * (double) p->K;
			if (const char *t = getenv("MF_SWEEP_LONG")) thr = atof(t);
			const int t_long = std::max(128, (int) std::min(thr, 2e9));
			// estimated bandwidth time of one sweep; below ~50 us the two-stream fork/join (tens of us on the 6000
			// launches of ML100k) costs more than the split saves: use one cooperative launch for all rows there
			const double est_us = (double) p->nnz * 8.0 * p->K / 6e12 * 1e6;
			const int nc = (int) std::min<size_t>(32, (48 * 1024) / per_entry);
			long long scratch_entries = 0;
			(void) nl;
			for (int kind = 0; kind < 2; ++kind) {
				const std::vector<int> &pt = kind == 0 ? cptr : rptr;
				const int nrows = kind == 0 ? p->items : p->uc;
				// ... and only rows well above the average count as long: when every row is equally long (the cfg4
				// twin: 1000 items x 1000 entries) there is no skew to fix and the single-wave kernel is the faster one
				const int t_kind = getenv("MF_SWEEP_LONG") ? t_long : std::max(t_long, (int) std::min<long long>(4 * (long long) (p->nnz / std::max(nrows, 1)), 2000000000ll));
				if (p->max_row_len[kind] < t_kind) continue;
				if (est_us < 50.0 && nrows < 4096 && !getenv("MF_SWEEP_LONG")) {
					if (p->sweep.coop && (nc >= 8 || getenv("MF_SWEEP_NCH"))) {
						p->coop_all[kind] = true;
						p->nch_coop = getenv("MF_SWEEP_NCH") ? nl : nc;
						p->lds_bytes_coop = head + (size_t) p->nch_coop * per_entry;
					}
					continue;
				}
				// the scratch buffer holds K doubles per entry of every extreme row: keep it under a quarter of the free
				// memory by raising the threshold (on Netflix-like data most entries sit in long columns)
				size_t free_b = 0, total_b = 0;
				(void) hipMemGetInfo(&free_b, &total_b);
				const size_t cap_entries = std::max<size_t>(free_b / 4 / ((size_t) ((p->K + 15) / 16) * 128), 1);
				int t_eff = t_kind;
				for (;;) {
					size_t ent = 0;
					for (int r = 0; r < nrows; ++r) {
						const int len = pt[(size_t) r + 1] - pt[r];
						if (len >= t_eff) ent += (size_t) len;
					}
					if (ent <= cap_entries || t_eff > (1 << 29)) break;
					t_eff *= 2;
				}
				if (p->max_row_len[kind] < t_eff) continue;
				std::vector<int> lg, sh;
				for (int r = 0; r < nrows; ++r) (pt[(size_t) r + 1] - pt[r] >= t_eff ? lg : sh).push_back(r);
				MF_TRY(dev_alloc(&p->long_rows[kind], lg.size()));
				MF_TRY(dev_alloc(&p->short_rows[kind], sh.size()));
				MF_TRY_HIP(hipMemcpy(p->long_rows[kind], lg.data(), lg.size() * sizeof(int), hipMemcpyHostToDevice));
				if (!sh.empty())
					MF_TRY_HIP(hipMemcpy(p->short_rows[kind], sh.data(), sh.size() * sizeof(int), hipMemcpyHostToDevice));
				p->n_long[kind] = (int) lg.size();
				p->n_short[kind] = (int) sh.size();
				// segments of 256 entries; scratch offsets in entry units, rows back to back
				constexpr int kSeg = 256;
				std::vector<int> srow, sbeg, send, lcnt;
				std::vector<long long> sout, lbeg;
				long long off = 0;
				for (int r : lg) {
					const int b = pt[r], e = pt[(size_t) r + 1];
					lbeg.push_back(off);
					lcnt.push_back(e - b);
					for (int c = b; c < e; c += kSeg) {
						srow.push_back(r);
						sbeg.push_back(c);
						send.push_back(std::min(e, c + kSeg));
						sout.push_back(off + (c - b));
					}
					off += e - b;
				}
				scratch_entries = std::max(scratch_entries, off);
				p->n_seg[kind] = (int) srow.size();
				MF_TRY(dev_alloc(&p->seg_row[kind], srow.size()));
				MF_TRY(dev_alloc(&p->seg_beg[kind], srow.size()));
				MF_TRY(dev_alloc(&p->seg_end[kind], srow.size()));
				MF_TRY(dev_alloc(&p->seg_out[kind], srow.size()));
				MF_TRY(dev_alloc(&p->lr_sbeg[kind], lg.size()));
				MF_TRY(dev_alloc(&p->lr_cnt[kind], lg.size()));
				MF_TRY_HIP(hipMemcpy(p->seg_row[kind], srow.data(), srow.size() * sizeof(int), hipMemcpyHostToDevice));
				MF_TRY_HIP(hipMemcpy(p->seg_beg[kind], sbeg.data(), srow.size() * sizeof(int), hipMemcpyHostToDevice));
				MF_TRY_HIP(hipMemcpy(p->seg_end[kind], send.data(), srow.size() * sizeof(int), hipMemcpyHostToDevice));
				MF_TRY_HIP(hipMemcpy(p->seg_out[kind], sout.data(), srow.size() * sizeof(long long), hipMemcpyHostToDevice));
				MF_TRY_HIP(hipMemcpy(p->lr_sbeg[kind], lbeg.data(), lg.size() * sizeof(long long), hipMemcpyHostToDevice));
				MF_TRY_HIP(hipMemcpy(p->lr_cnt[kind], lcnt.data(), lg.size() * sizeof(int), hipMemcpyHostToDevice));
			}
			if (p->coop_all[0] || p->coop_all[1])
				MF_TRY_HIP(hipFuncSetAttribute((const void *) p->sweep.coop, hipFuncAttributeMaxDynamicSharedMemorySize,
				                               (int) p->lds_bytes_coop));
			if (p->n_long[0] || p->n_long[1]) {
				MF_TRY_HIP(hipFuncSetAttribute((const void *) p->sweep.prod, hipFuncAttributeMaxDynamicSharedMemorySize,
				                               (int) p->lds_bytes));
				// [16-column slice][entry][16 doubles]; 8 entries of padding per slice: the last block of a row is read whole
				p->scratch_entries = (size_t) scratch_entries + 8;
				MF_TRY(dev_alloc(&p->scratch, p->scratch_entries * 16 * (size_t) ((p->K + 15) / 16)));
				MF_TRY_HIP(hipStreamCreateWithFlags(&p->side_stream, hipStreamNonBlocking));
				MF_TRY_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
				MF_TRY_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
			}
		}
	}

	const size_t nl = (size_t) p->uc * p->K, nr = (size_t) p->items * p->K;
	MF_TRY(dev_alloc(&p->Lbuf[0], nl));
	MF_TRY(dev_alloc(&p->Lbuf[1], nl));
	if (s->items_ext[0] && s->items_ext[1]) {
		p->r_external = true;
		p->Rbuf[0] = (double *) s->items_ext[0];
		p->Rbuf[1] = (double *) s->items_ext[1];
	} else {
		MF_TRY(dev_alloc(&p->Rbuf[0], nr));
		MF_TRY(dev_alloc(&p->Rbuf[1], nr));
	}
	MF_TRY(dev_alloc(&p->best_dev, (size_t) p->uc));
	MF_TRY(dev_alloc(&p->lnorm, (size_t) p->uc));
	MF_TRY(dev_alloc(&p->rmax_bits, 1));
	MF_TRY(dev_alloc(&p->ulist, (size_t) p->uc));
	MF_TRY(dev_alloc(&p->ucount, 1));
#undef MF_TRY
#undef MF_TRY_HIP
	*out = p;
	return MF_OK;
}

void mf_plan_destroy(mf_plan *p)
{
	if (!p) return;
	(void) hipSetDevice(p->device);
	if (p->stream) (void) hipStreamSynchronize(p->stream);
	for (auto &t : p->timed) {
		(void) hipEventDestroy(t.t0);
		(void) hipEventDestroy(t.t1);
	}
	(void) hipFree(p->csr_ptr);
	(void) hipFree(p->csr_idx);
	(void) hipFree(p->csr_val);
	(void) hipFree(p->csc_ptr);
	(void) hipFree(p->csc_idx);
	(void) hipFree(p->csc_val);
	(void) hipFree(p->Lbuf[0]);
	(void) hipFree(p->Lbuf[1]);
	if (!p->r_external) {
		(void) hipFree(p->Rbuf[0]);
		(void) hipFree(p->Rbuf[1]);
	}
	(void) hipFree(p->best_dev);
	for (int k = 0; k < 2; ++k) {
		(void) hipFree(p->long_rows[k]);
		(void) hipFree(p->short_rows[k]);
		(void) hipFree(p->seg_row[k]);
		(void) hipFree(p->seg_beg[k]);
		(void) hipFree(p->seg_end[k]);
		(void) hipFree(p->seg_out[k]);
		(void) hipFree(p->lr_sbeg[k]);
		(void) hipFree(p->lr_cnt[k]);
	}
	(void) hipFree(p->scratch);
	if (p->side_stream) (void) hipStreamDestroy(p->side_stream);
	if (p->ev_fork) (void) hipEventDestroy(p->ev_fork);
	if (p->ev_join) (void) hipEventDestroy(p->ev_join);
	(void) hipFree(p->lnorm);
	(void) hipFree(p->rmax_bits);
	(void) hipFree(p->ulist);
	(void) hipFree(p->ucount);
	if (p->own_stream) (void) hipStreamDestroy(p->own_stream);
	delete p;
}

int mf_plan_set_stream(mf_plan *p, void *hip_stream)
{
	if (!p) return MF_ERR_ARGUMENT;
	MF_HIP(hipSetDevice(p->device));
	MF_HIP(hipStreamSynchronize(p->stream));
	p->stream = hip_stream ? (hipStream_t) hip_stream : p->own_stream;
	return MF_OK;
}

int mf_plan_upload_factors(mf_plan *p, const double *L_block, const double *R)
{
	if (!p || (!L_block && p->uc > 0) || (!R && p->items > 0)) return MF_ERR_ARGUMENT;
	MF_HIP(hipSetDevice(p->device));
	const size_t nl = (size_t) p->uc * p->K * sizeof(double), nr = (size_t) p->items * p->K * sizeof(double);
	if (nl) MF_HIP(hipMemcpyAsync(p->Lbuf[p->cur], L_block, nl, hipMemcpyHostToDevice, p->stream));
	if (nr) MF_HIP(hipMemcpyAsync(p->Rbuf[p->cur], R, nr, hipMemcpyHostToDevice, p->stream));
	MF_HIP(hipStreamSynchronize(p->stream));
	p->have_factors = true;
	return MF_OK;
}

int mf_plan_download_factors(mf_plan *p, double *L_block, double *R)
{
	if (!p) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	const size_t nl = (size_t) p->uc * p->K * sizeof(double), nr = (size_t) p->items * p->K * sizeof(double);
	if (L_block && nl) MF_HIP(hipMemcpyAsync(L_block, p->Lbuf[p->cur], nl, hipMemcpyDeviceToHost, p->stream));
	if (R && nr) MF_HIP(hipMemcpyAsync(R, p->Rbuf[p->cur], nr, hipMemcpyDeviceToHost, p->stream));
	MF_HIP(hipStreamSynchronize(p->stream));
	return MF_OK;
}

int mf_plan_sweep_items(mf_plan *p, int seed_from_old)
{
	if (!p) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	return launch_sweep(p, 0, seed_from_old ? 1 : 0);
}

int mf_plan_sweep_users(mf_plan *p)
{
	if (!p) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	return launch_sweep(p, 1, 1);
}

void *mf_plan_items_next(mf_plan *p) { return p ? p->Rbuf[p->cur ^ 1] : nullptr; }
void *mf_plan_items_current(mf_plan *p) { return p ? p->Rbuf[p->cur] : nullptr; }

int mf_plan_flip(mf_plan *p)
{
	if (!p) return MF_ERR_ARGUMENT;
	p->cur ^= 1;
	return MF_OK;
}

static int iterate_eager(mf_plan *p, int iters)
{
	for (int it = 0; it < iters; ++it) {
		int rc = launch_sweep(p, 0, 1);
		if (rc != MF_OK) return rc;
		rc = launch_sweep(p, 1, 1);
		if (rc != MF_OK) return rc;
		p->cur ^= 1;
	}
	return MF_OK;
}

int mf_plan_iterate(mf_plan *p, int iters)
{
	if (!p || iters < 0) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	// Launch-bound regime (inst1: 100000 iterations of a 13-entry instance, ~4 us per launch): capture an even
	// number of iterations -- so the ping-pong parity returns to where it started -- into a HIP graph and replay it.
	// Only for small sweeps; a large sweep is not launch-bound and a graph would pin its arguments for nothing.
	const char *genv = getenv("MF_GRAPH");   // "0" disables
	const bool small = (double) p->nnz * p->K < 2e6 && !p->timing && p->n_long[0] == 0 && p->n_long[1] == 0;
	constexpr int kGraphIters = 32;
	if (small && iters >= 4 * kGraphIters && !(genv && genv[0] == '0')) {
		hipGraph_t graph = nullptr;
		hipGraphExec_t exec = nullptr;
		const int cur0 = p->cur;
		hipError_t e = hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal);
		int rc = MF_OK;
		if (e == hipSuccess) {
			rc = iterate_eager(p, kGraphIters);
			e = hipStreamEndCapture(p->stream, &graph);
		}
		if (e == hipSuccess && rc == MF_OK) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
		if (e == hipSuccess && rc == MF_OK) {
			p->cur = cur0;   // the capture only recorded; kGraphIters is even, so every replay starts from cur0
			const int replays = iters / kGraphIters;
			for (int g = 0; g < replays && e == hipSuccess; ++g) e = hipGraphLaunch(exec, p->stream);
			iters -= replays * kGraphIters;
		}
		if (exec) (void) hipGraphExecDestroy(exec);
		if (graph) (void) hipGraphDestroy(graph);
		if (rc != MF_OK) return rc;
		if (e != hipSuccess) {
			g_last_hip_error = std::string("hip graph path: ") + hipGetErrorString(e);
			return MF_ERR_HIP;
		}
	}
	return iterate_eager(p, iters);
}

int mf_plan_recommend(mf_plan *p, int32_t *best)
{
	if (!p || (!best && p->uc > 0)) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	if (p->uc == 0) return MF_OK;
	const char *impl = getenv("MF_RECOMMEND_IMPL");   // "mfma" (default) | "exact"
	const bool use_mfma = !(impl && strcmp(impl, "exact") == 0);
	mf::RecArgs ex;
	ex.users = p->uc;
	ex.items = p->items;
	ex.K = p->K;
	ex.L = p->Lbuf[p->cur];
	ex.R = p->Rbuf[p->cur];
	ex.csr_ptr = p->csr_ptr;
	ex.csr_idx = p->csr_idx;
	ex.best = p->best_dev;
	ex.ulist = nullptr;
	if (!use_mfma) {
		const int grid = (p->uc + mf::kRT - 1) / mf::kRT;
		hipLaunchKernelGGL(mf::recommend_kernel, dim3(grid), dim3(256), 0, p->stream, ex);
		MF_HIP(hipGetLastError());
		p->last_uncertain = -1;
	} else {
		// pass 1: scores on the FP64 matrix cores + certification margin; pass 2: exact re-scoring of the rest
		MF_HIP(hipMemsetAsync(p->rmax_bits, 0, sizeof(unsigned long long), p->stream));
		MF_HIP(hipMemsetAsync(p->ucount, 0, sizeof(int), p->stream));
		hipLaunchKernelGGL(mf::row_norm_kernel, dim3((p->uc + 63) / 64), dim3(64), 0, p->stream, ex.L, p->uc,
		                   p->K, p->lnorm, (unsigned long long *) nullptr);
		if (p->items > 0)
			hipLaunchKernelGGL(mf::row_norm_kernel, dim3((p->items + 63) / 64), dim3(64), 0, p->stream, ex.R,
			                   p->items, p->K, (double *) nullptr, p->rmax_bits);
		mf::RecMfmaArgs m;
		m.users = p->uc;
		m.items = p->items;
		m.K = p->K;
		m.L = ex.L;
		m.R = ex.R;
		m.csr_ptr = p->csr_ptr;
		m.csr_idx = p->csr_idx;
		m.lnorm = p->lnorm;
		m.rnorm_max_bits = p->rmax_bits;
		m.thr_scale = 8.0 * (double) (p->K + 8) * 1.1102230246251565e-16;
		m.best = p->best_dev;
		m.ulist = p->ulist;
		m.ucount = p->ucount;
		if ((p->K & 1) == 0)
			hipLaunchKernelGGL(mf::recommend_mfma_kernel<true>, dim3((p->uc + mf::kMU - 1) / mf::kMU),
			                   dim3(mf::kMThreads), 0, p->stream, m);
		else
			hipLaunchKernelGGL(mf::recommend_mfma_kernel<false>, dim3((p->uc + mf::kMU - 1) / mf::kMU),
			                   dim3(mf::kMThreads), 0, p->stream, m);
		MF_HIP(hipGetLastError());
		int cnt = 0;
		MF_HIP(hipMemcpyAsync(&cnt, p->ucount, sizeof(int), hipMemcpyDeviceToHost, p->stream));
		MF_HIP(hipStreamSynchronize(p->stream));
		p->last_uncertain = cnt;
		if (cnt > 0) {
			ex.users = cnt;
			ex.ulist = p->ulist;
			hipLaunchKernelGGL(mf::recommend_kernel, dim3((cnt + mf::kRT - 1) / mf::kRT), dim3(256), 0, p->stream,
			                   ex);
			MF_HIP(hipGetLastError());
		}
	}
	MF_HIP(hipMemcpyAsync(best, p->best_dev, (size_t) p->uc * sizeof(int), hipMemcpyDeviceToHost, p->stream));
	MF_HIP(hipStreamSynchronize(p->stream));
	return MF_OK;
}

int mf_plan_recommend_info(mf_plan *p, int64_t *exact_pass_users)
{
	if (!p || !exact_pass_users) return MF_ERR_ARGUMENT;
	*exact_pass_users = p->last_uncertain;
	return MF_OK;
}

int mf_plan_predict(mf_plan *p, double *B)
{
	if (!p || !B) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	const size_t n = (size_t) p->uc * (size_t) p->items;
	if (n > ((size_t) 1 << 26)) return MF_ERR_UNSUPPORTED;
	if (n == 0) return MF_OK;
	MF_HIP(hipSetDevice(p->device));
	double *dB = nullptr;
	MF_HIP(hipMalloc((void **) &dB, n * sizeof(double)));
	hipLaunchKernelGGL(mf::predict_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, p->stream,
	                   p->Lbuf[p->cur], p->Rbuf[p->cur], p->uc, p->items, p->K, dB);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(B, dB, n * sizeof(double), hipMemcpyDeviceToHost, p->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
	(void) hipFree(dB);
	if (e != hipSuccess) {
		g_last_hip_error = std::string("mf_plan_predict: ") + hipGetErrorString(e);
		return MF_ERR_HIP;
	}
	return MF_OK;
}

int mf_plan_synchronize(mf_plan *p)
{
	if (!p) return MF_ERR_ARGUMENT;
	MF_HIP(hipSetDevice(p->device));
	MF_HIP(hipStreamSynchronize(p->stream));
	return MF_OK;
}

int mf_plan_timing(mf_plan *p, int enable)
{
	if (!p) return MF_ERR_ARGUMENT;
	p->timing = enable != 0;
	return MF_OK;
}

int mf_plan_timing_read(mf_plan *p, int64_t *item_launches, double *item_ms, int64_t *user_launches,
                        double *user_ms)
{
	if (!p) return MF_ERR_ARGUMENT;
	MF_HIP(hipSetDevice(p->device));
	const int rc = drain_timing(p);
	if (rc != MF_OK) return rc;
	if (item_launches) *item_launches = p->acc_launch[0];
	if (item_ms) *item_ms = p->acc_ms[0];
	if (user_launches) *user_launches = p->acc_launch[1];
	if (user_ms) *user_ms = p->acc_ms[1];
	p->acc_launch[0] = p->acc_launch[1] = 0;
	p->acc_ms[0] = p->acc_ms[1] = 0.0;
	return MF_OK;
}

int mf_plan_describe(mf_plan *p, char *buf, int buflen)
{
	if (!p || !buf || buflen <= 0) return MF_ERR_ARGUMENT;
	if (p->sweep.dma)
		snprintf(buf, (size_t) buflen,
		         "sweep_dma_kernel<KT=%d,NPASS=%d> K=%d nch=%d row_bytes=%d lds=%zu long_rows=%d/%d coop_nch=%d",
		         p->sweep.kt, p->sweep.kt ? (p->K / 2 + 63) / 64 : p->sweep.kpmax, p->K, p->nch, p->sweep.row_bytes,
		         p->lds_bytes, p->n_long[0] + (p->coop_all[0] ? p->items : 0), p->n_long[1] + (p->coop_all[1] ? p->uc : 0),
		         p->coop_all[0] || p->coop_all[1] ? p->nch_coop : 0);
	else
		snprintf(buf, (size_t) buflen, "sweep_kernel<KT=%d,KPMAX=%d> K=%d nch=%d stride=%d lds=%zu",
		         p->sweep.kt, p->sweep.kpmax, p->K, p->nch, p->stride, p->lds_bytes);
	return MF_OK;
}

/* ---------------------------------------------------------------------------------------- LEVEL 1 */

static int make_single_plan(const mf_problem *pr, int device, mf_plan **out, std::vector<int32_t> &row,
                            std::vector<int32_t> &col, std::vector<double> &val)
{
	if (!pr || pr->users < 0 || pr->items < 0 || pr->features < 1 || pr->nnz < 0 || pr->iters < 0 ||
	    (pr->nnz > 0 && !pr->entries))
		return MF_ERR_ARGUMENT;
	try {
		row.resize((size_t) pr->nnz);
		col.resize((size_t) pr->nnz);
		val.resize((size_t) pr->nnz);
	} catch (const std::bad_alloc &) {
		return MF_ERR_NO_MEMORY;
	}
	for (int64_t n = 0; n < pr->nnz; ++n) {
		row[(size_t) n] = pr->entries[n].row;
		col[(size_t) n] = pr->entries[n].col;
		val[(size_t) n] = pr->entries[n].value;
	}
	mf_shard s;
	memset(&s, 0, sizeof s);
	s.users_total = pr->users;
	s.items = pr->items;
	s.features = pr->features;
	s.user_begin = 0;
	s.user_count = pr->users;
	s.nnz = pr->nnz;
	s.row = row.data();
	s.col = col.data();
	s.val = val.data();
	s.alpha = pr->alpha;
	s.device = device;
	return mf_plan_create(out, &s);
}

int mf_backend_run(const mf_problem *pr, double *L, double *R, int32_t *best, int device)
{
	if (!pr || !L || !R) return MF_ERR_ARGUMENT;   // L and R carry the initial factors in
	mf_plan *p = nullptr;
	std::vector<int32_t> row, col;
	std::vector<double> val;
	int rc = make_single_plan(pr, device, &p, row, col, val);
	if (rc != MF_OK) return rc;
	rc = mf_plan_upload_factors(p, L, R);
	if (rc == MF_OK) rc = mf_plan_iterate(p, pr->iters);
	if (rc == MF_OK && best) rc = mf_plan_recommend(p, best);
	if (rc == MF_OK) rc = mf_plan_download_factors(p, L, R);
	mf_plan_destroy(p);
	return rc;
}

// One sharded factorisation: the factor with `nrows_a` rows ("A": users, or items when transposed) is cut into
// ndev contiguous blocks and kept private; the other factor ("B") is replicated and summed after every sweep.
// key_a / key_b are the entries' indices into A and B in file order.
static int run_shards(int ndev, const int *devices, int nrows_a, int nrows_b, int K, int64_t nnz,
                      const int32_t *key_a, const int32_t *key_b, const double *val, double alpha, int iters,
                      double *A, double *B, std::vector<int> &begin)
{
	// ---- blocks of A balanced by entry count (cut at row boundaries)
	std::vector<int64_t> cnt((size_t) nrows_a + 1, 0);
	for (int64_t n = 0; n < nnz; ++n) cnt[(size_t) key_a[n] + 1]++;
	for (int u = 0; u < nrows_a; ++u) cnt[(size_t) u + 1] += cnt[u];
	begin.assign((size_t) ndev + 1, 0);
	{
		int u = 0;
		for (int g = 1; g < ndev; ++g) {
			const int64_t target = cnt[nrows_a] * g / ndev;
			while (u < nrows_a && cnt[u] < target) ++u;
			begin[g] = u;
		}
		begin[ndev] = nrows_a;
	}
	std::vector<mf_plan *> plan((size_t) ndev, nullptr);
	std::vector<hipEvent_t> ev_items((size_t) ndev, nullptr), ev_red((size_t) ndev, nullptr);
	int rc = MF_OK;
	// ---- one resident plan per shard (entries filtered in file order)
	for (int g = 0; g < ndev && rc == MF_OK; ++g) {
		std::vector<int32_t> row, col;
		std::vector<double> v;
		for (int64_t n = 0; n < nnz; ++n)
			if (key_a[n] >= begin[g] && key_a[n] < begin[g + 1]) {
				row.push_back(key_a[n]);
				col.push_back(key_b[n]);
				v.push_back(val[n]);
			}
		mf_shard s;
		memset(&s, 0, sizeof s);
		s.users_total = nrows_a;
		s.items = nrows_b;
		s.features = K;
		s.user_begin = begin[g];
		s.user_count = begin[g + 1] - begin[g];
		s.nnz = (int64_t) row.size();
		s.row = row.data();
		s.col = col.data();
		s.val = v.data();
		s.alpha = alpha;
		s.device = devices[g];
		rc = mf_plan_create(&plan[g], &s);
		if (rc == MF_OK) rc = mf_plan_upload_factors(plan[g], A + (size_t) begin[g] * K, B);
		if (rc == MF_OK) {
			if (hipEventCreateWithFlags(&ev_items[g], hipEventDisableTiming) != hipSuccess ||
			    hipEventCreateWithFlags(&ev_red[g], hipEventDisableTiming) != hipSuccess)
				rc = MF_ERR_HIP;
		}
	}
	// ---- iterations: B sweep (shard 0 seeds from the old factor, matFact-mpi.c:187) -> A sweep -> wait for every
	//      shard's B sweep -> reduce my slice over all buffers -> wait for every reduce -> flip
	const size_t nb = (size_t) nrows_b * K;
	for (int it = 0; it < iters && rc == MF_OK; ++it) {
		for (int g = 0; g < ndev && rc == MF_OK; ++g) {
			rc = mf_plan_sweep_items(plan[g], g == 0);
			if (rc == MF_OK && hipEventRecord(ev_items[g], plan[g]->stream) != hipSuccess) rc = MF_ERR_HIP;
			if (rc == MF_OK) rc = mf_plan_sweep_users(plan[g]);
		}
		for (int g = 0; g < ndev && rc == MF_OK; ++g) {
			(void) hipSetDevice(plan[g]->device);
			for (int h = 0; h < ndev; ++h)
				if (h != g && hipStreamWaitEvent(plan[g]->stream, ev_items[h], 0) != hipSuccess) rc = MF_ERR_HIP;
			mf::PeerReduceArgs a;
			a.nshards = ndev;
			for (int h = 0; h < ndev; ++h) a.buf[h] = plan[h]->Rbuf[plan[h]->cur ^ 1];
			a.begin = ((nb / 2) * g / ndev) * 2;
			a.end = g == ndev - 1 ? nb : ((nb / 2) * (g + 1) / ndev) * 2;
			if (a.end > a.begin && rc == MF_OK) {
				const size_t pairs = (a.end - a.begin + 1) / 2;
				const unsigned grid = (unsigned) std::min<size_t>((pairs + 255) / 256, 2048);
				hipLaunchKernelGGL(mf::peer_allreduce_kernel, dim3(grid), dim3(256), 0, plan[g]->stream, a);
				if (hipGetLastError() != hipSuccess) rc = MF_ERR_HIP;
			}
			if (rc == MF_OK && hipEventRecord(ev_red[g], plan[g]->stream) != hipSuccess) rc = MF_ERR_HIP;
		}
		for (int g = 0; g < ndev && rc == MF_OK; ++g) {
			(void) hipSetDevice(plan[g]->device);
			for (int h = 0; h < ndev; ++h)
				if (h != g && hipStreamWaitEvent(plan[g]->stream, ev_red[h], 0) != hipSuccess) rc = MF_ERR_HIP;
			mf_plan_flip(plan[g]);
		}
	}
	for (int g = 0; g < ndev && rc == MF_OK; ++g) rc = mf_plan_synchronize(plan[g]);
	for (int g = 0; g < ndev && rc == MF_OK; ++g)
		rc = mf_plan_download_factors(plan[g], A + (size_t) begin[g] * K, g == 0 ? B : nullptr);
	for (int g = 0; g < ndev; ++g) {
		if (plan[g]) (void) hipSetDevice(plan[g]->device);
		if (ev_items[g]) (void) hipEventDestroy(ev_items[g]);
		if (ev_red[g]) (void) hipEventDestroy(ev_red[g]);
		mf_plan_destroy(plan[g]);
	}
	return rc;
}

int mf_backend_run_multi(const mf_problem *pr, double *L, double *R, int32_t *best, const int *devices, int ndev)
{
	if (!pr || !L || !R || !devices || ndev < 1 || ndev > mf::kMaxShards || pr->users < 0 || pr->items < 0 ||
	    pr->features < 1 || pr->nnz < 0 || pr->iters < 0 || (pr->nnz > 0 && !pr->entries))
		return MF_ERR_ARGUMENT;
	if (ndev == 1) return mf_backend_run(pr, L, R, best, devices[0]);
	const int total = mf_backend_device_count();
	if (total <= 0) return MF_ERR_NO_DEVICE;
	for (int g = 0; g < ndev; ++g)
		if (devices[g] < 0 || devices[g] >= total) return MF_ERR_NO_DEVICE;
	const int U = pr->users, I = pr->items, K = pr->features;
	std::vector<int32_t> row((size_t) pr->nnz), col((size_t) pr->nnz);
	std::vector<double> val((size_t) pr->nnz);
	for (int64_t n = 0; n < pr->nnz; ++n) {
		const mf_entry &e = pr->entries[n];
		if (e.row < 0 || e.row >= U || e.col < 0 || e.col >= I) return MF_ERR_ARGUMENT;
		row[(size_t) n] = e.row;
		col[(size_t) n] = e.col;
		val[(size_t) n] = e.value;
	}
	// ---- peer access between distinct devices
	for (int g = 0; g < ndev; ++g)
		for (int h = 0; h < ndev; ++h)
			if (devices[g] != devices[h]) {
				int can = 0;
				MF_HIP(hipDeviceCanAccessPeer(&can, devices[g], devices[h]));
				if (!can) return MF_ERR_UNSUPPORTED;
				MF_HIP(hipSetDevice(devices[g]));
				const hipError_t e = hipDeviceEnablePeerAccess(devices[h], 0);
				if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) MF_HIP(e);
				(void) hipGetLastError();
			}
	// ---- which factor is cut?  The one with MORE rows stays private, the smaller one is replicated and summed:
	// users >= items -> cut the users (8x1 grid); items > users -> cut the items (1x8 grid) -- the aspect-ratio
	// rule of create_balanced_grid (mpiutil.c:54-88) and of matFact-omp's reduce_L (matFact-omp.c:44).  Cutting the
	// items is the same computation with the roles of (row, L) and (col, R) exchanged; file order is untouched, so
	// every per-row and per-column summation order is too.
	std::vector<int> begin;
	int rc;
	if (I > U)
		rc = run_shards(ndev, devices, I, U, K, pr->nnz, col.data(), row.data(), val.data(), pr->alpha, pr->iters, R,
		                L, begin);
	else
		rc = run_shards(ndev, devices, U, I, K, pr->nnz, row.data(), col.data(), val.data(), pr->alpha, pr->iters, L,
		                R, begin);
	if (rc != MF_OK || !best) return rc;
	// ---- recommendations: always by user blocks against the full R (no collective, matFact-mpi.c:82-96 fused form)
	for (int g = 0; g < ndev && rc == MF_OK; ++g) {
		const int b0 = (int) ((int64_t) U * g / ndev), b1 = (int) ((int64_t) U * (g + 1) / ndev);
		if (b1 == b0) continue;
		std::vector<int32_t> r2, c2;
		std::vector<double> v2;
		for (int64_t n = 0; n < pr->nnz; ++n)
			if (row[(size_t) n] >= b0 && row[(size_t) n] < b1) {
				r2.push_back(row[(size_t) n]);
				c2.push_back(col[(size_t) n]);
				v2.push_back(val[(size_t) n]);
			}
		mf_shard s;
		memset(&s, 0, sizeof s);
		s.users_total = U;
		s.items = I;
		s.features = K;
		s.user_begin = b0;
		s.user_count = b1 - b0;
		s.nnz = (int64_t) r2.size();
		s.row = r2.data();
		s.col = c2.data();
		s.val = v2.data();
		s.alpha = pr->alpha;
		s.device = devices[g];
		mf_plan *plan = nullptr;
		rc = mf_plan_create(&plan, &s);
		if (rc == MF_OK) rc = mf_plan_upload_factors(plan, L + (size_t) b0 * K, R);
		if (rc == MF_OK) rc = mf_plan_recommend(plan, best + b0);
		mf_plan_destroy(plan);
	}
	return rc;
}

int mf_backend_factorize(const mf_problem *pr, double *L, double *R, int device)
{
	return mf_backend_run(pr, L, R, nullptr, device);
}

int mf_backend_recommend(const mf_problem *pr, const double *L, const double *R, int32_t *best, int device)
{
	if (!pr || !L || !R || (!best && pr->users > 0)) return MF_ERR_ARGUMENT;
	mf_plan *p = nullptr;
	std::vector<int32_t> row, col;
	std::vector<double> val;
	int rc = make_single_plan(pr, device, &p, row, col, val);
	if (rc != MF_OK) return rc;
	rc = mf_plan_upload_factors(p, L, R);
	if (rc == MF_OK) rc = mf_plan_recommend(p, best);
	mf_plan_destroy(p);
	return rc;
}

}  // extern "C"
