// mf_build.hip.h -- CSR/CSC construction (device-side stable radix sort, host fallback) and the row schedule.
#pragma once

namespace {

// Host table -> device, ORDERED WITH THE PLAN'S STREAM.  The plan's stream is created hipStreamNonBlocking: nothing
// orders it with the null stream that a plain hipMemcpy / hipMemset uses, so every set-up transfer goes through the
// plan's own stream and is complete (the host vector may go out of scope) when this returns.  (Round 3: three flaky
// mismatches in the GPU suite, never reproducible alone, all on plans whose set-up mixed the two streams.)
inline hipError_t h2d(mf_plan *p, void *dst, const void *src, size_t bytes)
{
	if (bytes == 0) return hipSuccess;
	const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, p->stream);
	return e != hipSuccess ? e : hipStreamSynchronize(p->stream);
}


// inside helpers that return a status directly
#define MF_TRY(x)                       \
	do {                                \
		int _rc = (x);                  \
		if (_rc != MF_OK) return _rc;   \
	} while (0)
#define MF_TRY_HIP(call) MF_HIP(call)

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
            std::vector<double> &v, std::vector<int> *pos_out = nullptr)
{
	if (pos_out) pos_out->resize((size_t) nnz);
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
		if (pos_out) (*pos_out)[(size_t) n] = pos;
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
	if (n > 0 && row[n - 1] == row[n] && col[n - 1] > col[n]) atomicOr(&flags[2], 1);   // columns not ascending in a row
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

// inverse of a permutation: inv[perm[q]] = q
__global__ void __launch_bounds__(256) invert_perm_kernel(const unsigned *__restrict__ perm, int64_t nnz,
                                                          int *__restrict__ inv)
{
	const int64_t q = (int64_t) blockIdx.x * 256 + threadIdx.x;
	if (q < nnz) inv[perm[q]] = (int) q;
}

// map[file2csr ? file2csr[perm_csc[q]] : perm_csc[q]] = q: CSR position -> CSC position of the same entry
__global__ void __launch_bounds__(256) csr2csc_kernel(const unsigned *__restrict__ perm_csc, int64_t nnz,
                                                      const int *__restrict__ file2csr, int *__restrict__ map)
{
	const int64_t q = (int64_t) blockIdx.x * 256 + threadIdx.x;
	if (q >= nnz) return;
	const unsigned f = perm_csc[q];
	map[file2csr ? file2csr[f] : (int) f] = (int) q;
}

__global__ void __launch_bounds__(256) fill_records_kernel(const int *__restrict__ idx, int64_t nnz,
                                                           mf::StreamRec *__restrict__ rec)
{
	const int64_t n = (int64_t) blockIdx.x * 256 + threadIdx.x;
	if (n < nnz) rec[n].idx = idx[n];
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
// the reference's array of structs -> the three arrays the build works on
__global__ void __launch_bounds__(256) split_entries_kernel(const mf_entry *__restrict__ e, int64_t nnz,
                                                            int *__restrict__ row, int *__restrict__ col,
                                                            double *__restrict__ val)
{
	const int64_t n = (int64_t) blockIdx.x * 256 + threadIdx.x;
	if (n >= nnz) return;
	const mf_entry x = e[n];
	row[n] = x.row;
	col[n] = x.col;
	val[n] = x.value;
}

int build_on_device(mf_plan *p, const mf_shard *s, const mf_entry *aos, bool swap, std::vector<int> &csr_ptr_host,
                    std::vector<int> &csc_ptr_host)
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
	int *d_row = nullptr, *d_col = nullptr, *d_flags = nullptr, *file2csr = nullptr;
	unsigned *key_in = nullptr, *key_out = nullptr, *perm_in = nullptr, *perm_out = nullptr;
	int rc;
	if ((rc = tmp.get(&d_row, nz)) != MF_OK || (rc = tmp.get(&d_col, nz)) != MF_OK ||
	    (rc = tmp.get(&key_in, nz)) != MF_OK || (rc = tmp.get(&key_out, nz)) != MF_OK ||
	    (rc = tmp.get(&perm_in, nz)) != MF_OK || (rc = tmp.get(&perm_out, nz)) != MF_OK ||
	    (rc = tmp.get(&d_flags, 3)) != MF_OK)
		return rc;
	// the values land directly in csr_val when the input is row-sorted (the usual case); otherwise csc_val is
	// used as the staging copy of the file-order values and overwritten last
	double *d_val = p->csc_val;
	const unsigned grid = (unsigned) ((nnz + 255) / 256);
	if (aos) {
		// one upload of the 16-byte structs, split on the device (no host pass over the entries)
		mf_entry *d_aos = nullptr;
		if ((rc = tmp.get(&d_aos, nz)) != MF_OK) return rc;
		MF_HIP(hipMemcpyAsync(d_aos, aos, nz * sizeof(mf_entry), hipMemcpyHostToDevice, st));
		hipLaunchKernelGGL(split_entries_kernel, dim3(grid), dim3(256), 0, st, d_aos, nnz, swap ? d_col : d_row,
		                   swap ? d_row : d_col, p->csr_val);
	} else {
		MF_HIP(hipMemcpyAsync(d_row, s->row, nz * sizeof(int), hipMemcpyHostToDevice, st));
		MF_HIP(hipMemcpyAsync(d_col, s->col, nz * sizeof(int), hipMemcpyHostToDevice, st));
		MF_HIP(hipMemcpyAsync(p->csr_val, s->val, nz * sizeof(double), hipMemcpyHostToDevice, st));
	}
	MF_HIP(hipMemsetAsync(d_flags, 0, 3 * sizeof(int), st));
	hipLaunchKernelGGL(prep_keys_kernel, dim3(grid), dim3(256), 0, st, d_row, d_col, nnz, p->u0, p->uc, p->items,
	                   key_in, perm_in, d_flags);
	int flags[3] = {0, 0, 0};
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
		if (p->want_map) {
			if ((rc = tmp.get(&file2csr, nz)) != MF_OK) return rc;
			hipLaunchKernelGGL(invert_perm_kernel, dim3(grid), dim3(256), 0, st, perm_out, nnz, file2csr);
		}
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
	if (!row_sorted || flags[2]) {
		// The recommendation masks rated items by walking a user's item ids in ascending order (print_output's cursor,
		// matFact.c:13-23, relies on (row, col)-sorted input).  The file is not: give the mask its own copy of the ids,
		// ascending inside every row -- the column-sorted sequence, stably re-sorted by row.  The sweeps keep file order.
		unsigned *mk_in = nullptr, *mk_out = nullptr, *mv_out = nullptr;
		if ((rc = tmp.get(&mk_in, nz)) != MF_OK || (rc = tmp.get(&mk_out, nz)) != MF_OK || (rc = tmp.get(&mv_out, nz)) != MF_OK)
			return rc;
		MF_HIP(dev_alloc(&p->mask_idx, nz + 64) == MF_OK ? hipSuccess : hipErrorOutOfMemory);
		hipLaunchKernelGGL(copy_keys_kernel, dim3(grid), dim3(256), 0, st, p->csc_idx, nnz, mk_in, perm_in);
		size_t need2 = 0;
		MF_HIP(rocprim::radix_sort_pairs(nullptr, need2, mk_in, mk_out, key_out, mv_out, nz, 0, bits_for(p->uc), st));
		void *d_temp2 = d_temp;
		if (need2 > temp_bytes && (rc = tmp.get((char **) &d_temp2, need2)) != MF_OK) return rc;
		MF_HIP(rocprim::radix_sort_pairs(d_temp2, need2, mk_in, mk_out, key_out, mv_out, nz, 0, bits_for(p->uc), st));
		MF_HIP(hipMemcpyAsync(p->mask_idx, mv_out, nz * sizeof(int), hipMemcpyDeviceToDevice, st));
	}
	if (p->want_map) {
		MF_HIP(dev_alloc(&p->csr2csc, nz + 64) == MF_OK ? hipSuccess : hipErrorOutOfMemory);
		hipLaunchKernelGGL(csr2csc_kernel, dim3(grid), dim3(256), 0, st, perm_out, nnz, file2csr, p->csr2csc);
	}
	MF_HIP(hipGetLastError());
	MF_HIP(hipMemcpyAsync(csr_ptr_host.data(), p->csr_ptr, ((size_t) p->uc + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
	MF_HIP(hipMemcpyAsync(csc_ptr_host.data(), p->csc_ptr, ((size_t) p->items + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
	MF_HIP(hipStreamSynchronize(st));
	return MF_OK;
}


// CSR over the shard's users and CSC over the items, on the device (default) or bucketed on the host
// (MF_BUILD=host, kept for A/B tests); rptr / cptr return the two row-pointer arrays for the schedule decisions.
int build_sparse(mf_plan *p, const mf_shard *s_in, const mf_entry *aos, bool swap, std::vector<int> &rptr,
                 std::vector<int> &cptr)
{
	if (p->cfg.build_host) {   // MF_BUILD=host
		// host fallback of the BUILD only (tests): works on the three arrays
		mf_shard sh = *s_in;
		std::vector<int32_t> hrow, hcol;
		std::vector<double> hval;
		if (aos) {
			try {
				hrow.resize((size_t) sh.nnz);
				hcol.resize((size_t) sh.nnz);
				hval.resize((size_t) sh.nnz);
			} catch (const std::bad_alloc &) {
				return MF_ERR_NO_MEMORY;
			}
			for (int64_t n = 0; n < sh.nnz; ++n) {
				hrow[(size_t) n] = swap ? aos[n].col : aos[n].row;
				hcol[(size_t) n] = swap ? aos[n].row : aos[n].col;
				hval[(size_t) n] = aos[n].value;
			}
			sh.row = hrow.data();
			sh.col = hcol.data();
			sh.val = hval.data();
		}
		const mf_shard *s = &sh;
		for (int64_t n = 0; n < s->nnz; ++n)
			if (s->row[n] < s->user_begin || s->row[n] >= s->user_begin + s->user_count || s->col[n] < 0 ||
			    s->col[n] >= s->items)
				return MF_ERR_ARGUMENT;
		std::vector<int> idx, pos_r, pos_c;
		std::vector<double> val;
		const size_t nz = (size_t) s->nnz;
		try {
			bucket(s->nnz, p->uc, s->row, p->u0, s->col, 0, s->val, rptr, idx, val, p->want_map ? &pos_r : nullptr);
		} catch (const std::bad_alloc &) {
			return MF_ERR_NO_MEMORY;
		}
		MF_TRY(dev_alloc(&p->csr_ptr, (size_t) p->uc + 1));
		MF_TRY(dev_alloc(&p->csr_idx, nz + 64));
		MF_TRY(dev_alloc(&p->csr_val, nz + 64));
		MF_TRY_HIP(h2d(p, p->csr_ptr, rptr.data(), ((size_t) p->uc + 1) * sizeof(int)));
		{
			// mask ids ascending inside every row (see build_on_device) when the file order is not
			bool ascending = true;
			for (int u = 0; u < p->uc && ascending; ++u)
				ascending = std::is_sorted(idx.begin() + rptr[(size_t) u], idx.begin() + rptr[(size_t) u + 1]);
			if (!ascending) {
				std::vector<int> mk(idx);
				for (int u = 0; u < p->uc; ++u) std::sort(mk.begin() + rptr[(size_t) u], mk.begin() + rptr[(size_t) u + 1]);
				MF_TRY(dev_alloc(&p->mask_idx, nz + 64));
				MF_TRY_HIP(h2d(p, p->mask_idx, mk.data(), nz * sizeof(int)));
			}
		}
		if (nz) {
			MF_TRY_HIP(h2d(p, p->csr_idx, idx.data(), nz * sizeof(int)));
			MF_TRY_HIP(h2d(p, p->csr_val, val.data(), nz * sizeof(double)));
		}
		try {
			bucket(s->nnz, p->items, s->col, 0, s->row, p->u0, s->val, cptr, idx, val, p->want_map ? &pos_c : nullptr);
		} catch (const std::bad_alloc &) {
			return MF_ERR_NO_MEMORY;
		}
		MF_TRY(dev_alloc(&p->csc_ptr, (size_t) p->items + 1));
		MF_TRY(dev_alloc(&p->csc_idx, nz + 64));
		MF_TRY(dev_alloc(&p->csc_val, nz + 64));
		MF_TRY_HIP(h2d(p, p->csc_ptr, cptr.data(), ((size_t) p->items + 1) * sizeof(int)));
		if (nz) {
			MF_TRY_HIP(h2d(p, p->csc_idx, idx.data(), nz * sizeof(int)));
			MF_TRY_HIP(h2d(p, p->csc_val, val.data(), nz * sizeof(double)));
		}
		if (p->want_map) {
			std::vector<int> map(nz + 1);
			for (size_t n = 0; n < nz; ++n) map[(size_t) pos_r[n]] = pos_c[n];
			MF_TRY(dev_alloc(&p->csr2csc, nz + 64));
			if (nz) MF_TRY_HIP(h2d(p, p->csr2csc, map.data(), nz * sizeof(int)));
		}
	} else {
		MF_TRY(build_on_device(p, s_in, aos, swap, rptr, cptr));
	}
	return MF_OK;
}

// Does the single-wave launch of this sweep take the wave-pair form (mf_sweep.hip.h: loader + compute wave per row)?  Where
// the kernel exists (64 <= K <= 128, compile-time K) and the side is
//   made of long rows: 512 entries per row or more on average, skewed or not -- a wave spends its life inside rows, where
//          the pair overlaps the gather of chunk c+1 with the arithmetic of chunk c (cfg4's 1e5 items of 1000 entries:
//          11.77 vs 12.19 ms, three alternating runs on one box).  On a skewed side of that kind (the Netflix shape's 17770
//          items of 3770 entries, cfg4-Zipf's) the extreme-row threshold moves up 2.7x with the pairs (plan_row_schedule),
//          the scratch round trip shrinks and the side stream no longer eats into the other sweep: Netflix shape 19.7 ->
//          17.1 ms, cfg4-Zipf 35.5 -> 31.2 (tools/r3_pair_nflx.sh; at the single-wave threshold the pairs LOSE there, 20.0
//          vs 19.7); or
//   small and skewed: at most 65536 rows and ~2 ms of bytes, the longest row at least four times the mean -- such a launch
//          ENDS on its long rows, which a pair walks 2.4x faster than one wave (cfg3 power-law).
// Short equally long rows stay on the single-wave form (cfg3 uniform, 253 / 166 entries per row: 0.207 vs 0.222 ms), and so
// does a large side of short rows (users of the Netflix shape 8.2 vs 6.8 ms, of cfg4 12.9 vs 11.4).  MF_SWEEP_PAIR=0|1 overrides.
static bool pair_long_rows(const mf_plan *p, int kind)
{
	const int nrows = kind == 0 ? p->items : p->uc;
	return nrows >= 512 && p->nnz / nrows >= 512;
}
bool pair_wanted(const mf_plan *p, int kind)
{
	if (!p->sweep.pair || p->cfg.sweep_pair == 0) return false;
	if (p->cfg.sweep_pair == 1) return true;
	if (p->cfg.sweep_pair_kind[kind] >= 0) return p->cfg.sweep_pair_kind[kind] == 1;   // MF_SWEEP_PAIR_I / _U (experiments build)
	const int nrows = kind == 0 ? p->items : p->uc;
	if (nrows < 512 || p->nnz <= 0) return false;
	if (pair_long_rows(p, kind)) return true;
	if ((long long) p->max_row_len[kind] < 4 * std::max<long long>(p->nnz / nrows, 1)) return false;
	return nrows <= 65536 && (double) p->nnz * 8.0 * p->K / 6e12 * 1e6 <= 2000.0;
}

// Schedule of the two sweeps from the row lengths: which rows count as long, whether a tiny sweep runs as ONE
// cooperative launch, and the segment tables + scratch buffer of the extreme-row path.
int plan_row_schedule(mf_plan *p, const std::vector<int> &rptr, const std::vector<int> &cptr)
{
	for (int u = 0; u < p->uc; ++u) p->max_row_len[1] = std::max(p->max_row_len[1], rptr[(size_t) u + 1] - rptr[u]);
	for (int j = 0; j < p->items; ++j) p->max_row_len[0] = std::max(p->max_row_len[0], cptr[(size_t) j + 1] - cptr[j]);
	// ---- long / short row lists.  A row is "long" when its serial walk would exceed a good part of the bandwidth time
	// of the whole sweep (nnz * 8K bytes at ~7 TB/s): len > 4e-6 (6e-6) * nnz * K, and never below 128 entries.  cfg4
	// has none; a power-law instance a few.
	const mf_config &cfg = p->cfg;
	if (p->sweep.prod && cfg.skew) {   // MF_SWEEP_SKEW=0 disables the split
		const size_t per_entry = 2 * (size_t) mf::kCoopProducers * (size_t) p->sweep.row_bytes;
		const size_t head = (size_t) p->sweep.xs_bytes;
		int nl = (int) std::min<size_t>(32, (kLdsPerCu - 4096 - head) / per_entry);
		if (const int v = cfg.sweep_nch; v >= 1 && head + (size_t) v * per_entry <= kLdsPerCu) nl = v;
		// (6e-6 where the accumulate form with the pipelined phases exists: its lone wave walks 0.13 us per entry at K=100
		// instead of 0.17, so fewer rows need the scratch round trip -- Netflix shape 21.05 -> 19.68 ms at 40000 instead of
		// 26800 entries, cfg4-Zipf 34.5 -> 33.9; round 2's 4e-6 otherwise)
		double thr = (p->sweep.pf ? 6e-6 : 4e-6) * (double) p->nnz * (double) p->K;
		if (cfg.sweep_long_set) thr = cfg.sweep_long;
		const int t_long = std::max(128, (int) std::min(thr, 2e9));
		// estimated bandwidth time of one sweep; below ~50 us the two-stream fork/join (tens of us on the 6000
		// launches of ML100k) costs more than the split saves: use one cooperative launch for all rows there
		const double est_us = (double) p->nnz * 8.0 * p->K / 6e12 * 1e6;
		const int nc = (int) std::min<size_t>(32, (48 * 1024) / per_entry);
		long long scratch_entries = 0;
		for (int kind = 0; kind < 2; ++kind) {
			const std::vector<int> &pt = kind == 0 ? cptr : rptr;
			const int nrows = kind == 0 ? p->items : p->uc;
			// ... and only rows well above the average count as long: when every row is equally long (the cfg4
			// twin: 1000 items x 1000 entries) there is no skew to fix and the single-wave kernel is the faster one
			// (a side of long rows walked by wave pairs -- 0.055 us per entry instead of 0.13 --: 16e-6 nnz K.  Netflix shape,
			// pairs on the item side: 18.5 / 17.9 / 17.1 / 18.6 ms at 60 / 80 / 100 / 160 thousand entries, the rule gives 107 000;
			// cfg4-Zipf 32.6 / 31.2 / 31.2 / 39.9 at 120 / 160 / 220 / 400 thousand, the rule gives 160 000)
			const bool pairs_long = pair_wanted(p, kind) && pair_long_rows(p, kind) && p->cfg.sweep_pair != 1;
			const int t_side = pairs_long ? std::max(128, (int) std::min(16e-6 * (double) p->nnz * (double) p->K, 2e9)) : t_long;
			const int t_kind = cfg.sweep_long_kind[kind] > 0 ? cfg.sweep_long_kind[kind] : cfg.sweep_long_set ? t_long : std::max(t_side, (int) std::min<long long>(4 * (long long) (p->nnz / std::max(nrows, 1)), 2000000000ll));
			if (p->max_row_len[kind] < t_kind) continue;
			// With the wave-pair form a long row is walked at ~0.055 us per entry at K=100 (a lone single wave: 0.13): when
			// even the longest row's walk fits the sweep's bandwidth time the split buys nothing and costs the scratch
			// round trip and a fork/join (cfg3 power-law users, longest row 2324: 0.160 -> 0.139 ms without the split).
			const bool forced_thr = cfg.sweep_long_set || cfg.sweep_long_kind[kind] > 0;
			if (pair_wanted(p, kind) && !forced_thr && est_us >= 50.0 &&
			    (double) p->max_row_len[kind] * 0.055 * p->K / 100.0 <= 1.3 * est_us)
				continue;
			if (est_us < 50.0 && nrows < 4096 && !cfg.sweep_long_set && cfg.sweep_long_kind[kind] <= 0) {
				if (p->sweep.coop && (nc >= 8 || cfg.sweep_nch)) {
					p->coop_all[kind] = true;
					p->nch_coop = cfg.sweep_nch ? nl : nc;
					p->lds_bytes_coop = head + (size_t) p->nch_coop * per_entry;
				}
				continue;
			}
			// the scratch buffer holds K doubles per entry of every extreme row: keep it under a quarter of the free
			// memory by raising the threshold (on Netflix-like data most entries sit in long columns)
			size_t free_b = 0, total_b = 0;
			(void) hipMemGetInfo(&free_b, &total_b);
			const size_t nsl = (size_t) ((p->K + mf::kSliceCols - 1) / mf::kSliceCols);
			const size_t cap_entries = std::max<size_t>(free_b / 4 / (nsl * mf::kSliceCols * 8), 1);
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
			std::vector<int> lg, sh, md;
			// mid-length rows: at least t_mid entries (default: a quarter of the extreme threshold, never below 4x the mean)
			const int t_mid = !p->sweep.db || cfg.sweep_mid == 0 ? t_eff
			                  : cfg.sweep_mid > 0            ? std::min(cfg.sweep_mid, t_eff)
			                                                 : std::min(t_eff, std::max(t_eff / 4, (int) std::min<long long>(4 * (long long) (p->nnz / std::max(nrows, 1)), 1 << 30)));
			for (int r = 0; r < nrows; ++r) {
				const int len = pt[(size_t) r + 1] - pt[r];
				(len >= t_eff ? lg : len >= t_mid ? md : sh).push_back(r);
			}
			// longest first: workgroups are dispatched in list order as slots free up, so the long walks start
			// at once and the short rows fill in behind them (longest-processing-time-first scheduling)
			auto by_len = [&](int x, int y) { return pt[(size_t) x + 1] - pt[x] > pt[(size_t) y + 1] - pt[y]; };
			if (!cfg.nosort) {
				std::stable_sort(sh.begin(), sh.end(), by_len);
				std::stable_sort(md.begin(), md.end(), by_len);
				std::stable_sort(lg.begin(), lg.end(), by_len);
			}
			if (!md.empty()) {
				MF_TRY(dev_alloc(&p->mid_rows[kind], md.size()));
				MF_TRY_HIP(h2d(p, p->mid_rows[kind], md.data(), md.size() * sizeof(int)));
				p->n_mid[kind] = (int) md.size();
			}
			MF_TRY(dev_alloc(&p->long_rows[kind], lg.size()));
			MF_TRY(dev_alloc(&p->short_rows[kind], sh.size()));
			MF_TRY_HIP(h2d(p, p->long_rows[kind], lg.data(), lg.size() * sizeof(int)));
			if (!sh.empty())
				MF_TRY_HIP(h2d(p, p->short_rows[kind], sh.data(), sh.size() * sizeof(int)));
			p->n_long[kind] = (int) lg.size();
			p->long_len[kind] = t_eff;
			p->n_short[kind] = (int) sh.size();
			// segments of 256 entries; scratch offsets in entry units, rows back to back
			// entries per segment of the products launch: one wave walks a segment chunk by chunk (~2.5 us per 16 entries
			// of exposed latency), so short segments finish sooner and there are more of them to overlap
			// (cfg3 power-law: 256 -> 64 entries 0.452 -> 0.421 ms per iteration; Netflix-shaped 21.6 -> 21.4 ms)
			const int kSeg = cfg.sweep_seg;
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
			MF_TRY_HIP(h2d(p, p->seg_row[kind], srow.data(), srow.size() * sizeof(int)));
			MF_TRY_HIP(h2d(p, p->seg_beg[kind], sbeg.data(), srow.size() * sizeof(int)));
			MF_TRY_HIP(h2d(p, p->seg_end[kind], send.data(), srow.size() * sizeof(int)));
			MF_TRY_HIP(h2d(p, p->seg_out[kind], sout.data(), srow.size() * sizeof(long long)));
			MF_TRY_HIP(h2d(p, p->lr_sbeg[kind], lbeg.data(), lg.size() * sizeof(long long)));
			MF_TRY_HIP(h2d(p, p->lr_cnt[kind], lcnt.data(), lg.size() * sizeof(int)));
		}
		if (p->coop_all[0] || p->coop_all[1])
			MF_TRY_HIP(raise_lds_limit((const void *) p->sweep.coop, p->lds_bytes_coop));
		if (p->n_long[0] || p->n_long[1]) {
			if (cfg.rest_coop && p->sweep.coop) {
				p->rest_coop = true;
				p->nch_coop = cfg.sweep_nch ? nl : nc;
				p->lds_bytes_coop = head + (size_t) p->nch_coop * per_entry;
				MF_TRY_HIP(raise_lds_limit((const void *) p->sweep.coop, p->lds_bytes_coop));
			}
			p->nch_prod = p->nch;
			if (const int v = cfg.sweep_pnch; v >= 1 && v <= 64 && (size_t) p->sweep.xs_bytes + (size_t) v * p->sweep.row_bytes <= kLdsPerCu) p->nch_prod = v;
			p->lds_bytes_prod = (size_t) p->sweep.xs_bytes + (size_t) p->nch_prod * p->sweep.row_bytes;
			p->lds_bytes_osum = mf::kOrderedSumLds;
			if (cfg.os_lds) p->lds_bytes_osum = std::min<size_t>(kLdsPerCu, std::max<size_t>(p->lds_bytes_osum, cfg.os_lds));
			MF_TRY_HIP(raise_lds_limit((const void *) p->sweep.prod, p->lds_bytes_prod));
			MF_TRY_HIP(raise_lds_limit(cfg.os_dpp ? (const void *) mf::ordered_sum_kernel<true> : (const void *) mf::ordered_sum_kernel<false>, p->lds_bytes_osum));
			// [slice][entry][kSliceCols doubles]; one block of padding per slice: the last block of a row is read whole
			p->scratch_entries = (size_t) scratch_entries + mf::kBlockEntries;
			MF_TRY(dev_alloc(&p->scratch, p->scratch_entries * mf::kSliceCols *
			                                  (size_t) ((p->K + mf::kSliceCols - 1) / mf::kSliceCols)));
			int prio_lo = 0, prio_hi = 0;
			MF_TRY_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
			// beside the single-wave form: high priority -- the ordered sums are few, latency-bound waves that must get their
			// slots ahead of the thousands of workgroups of the sweep they run under.  Beside the wave-pair form: LOW
			// priority -- there the launch ends on the pairs of the long rows, which must be dispatched at once, and the
			// side path has the whole other sweep to hide under (cfg3 power-law 0.304 -> 0.268 ms).  MF_SIDE_PRIO=0|1.
			const bool side_low = cfg.side_prio >= 0 ? cfg.side_prio == 0 : (pair_wanted(p, 0) || pair_wanted(p, 1));
			MF_TRY_HIP(hipStreamCreateWithPriority(&p->side_stream, hipStreamNonBlocking, side_low ? prio_lo : prio_hi));
			MF_TRY_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
			MF_TRY_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
			if (p->n_mid[0] || p->n_mid[1]) {
				// two tiles of nch_mid rows: as many as fit a third of a CU's LDS (K=100: 32 rows, 53 KB, three per CU)
				const size_t rb = (size_t) p->sweep.row_bytes, hd = (size_t) p->sweep.xs_bytes;
				int nm = cfg.mid_nch > 0 ? cfg.mid_nch : 32;
				while (nm > 4 && hd + 2 * (size_t) nm * rb > kLdsPerCu / (cfg.mid_nch > 0 ? 1 : 3)) --nm;
				p->nch_mid = std::min(nm, 64);
				p->lds_bytes_mid = hd + 2 * (size_t) p->nch_mid * rb;
				p->mid_coop = cfg.mid_coop && p->sweep.coop;
				if (p->mid_coop) {   // 2 buffers x 7 producers x nch rows
					int nc2 = cfg.mid_nch > 0 ? cfg.mid_nch : 13;
					while (nc2 > 1 && hd + (size_t) nc2 * per_entry > kLdsPerCu - 2048) --nc2;
					p->nch_mid = nc2;
					p->lds_bytes_mid = hd + (size_t) nc2 * per_entry;
					MF_TRY_HIP(raise_lds_limit((const void *) p->sweep.coop, p->lds_bytes_mid));
				} else
				MF_TRY_HIP(raise_lds_limit((const void *) p->sweep.db, p->lds_bytes_mid));
				MF_TRY_HIP(hipStreamCreateWithPriority(&p->mid_stream, hipStreamNonBlocking, prio_hi));
				MF_TRY_HIP(hipEventCreateWithFlags(&p->ev_mid_join, hipEventDisableTiming));
			}
		}
	}
	// ---- wave priority for the long rows of the single-wave launch (what the launch ends on): when the launch is skewed
	// (its longest row at least four times its mean), rows of at least twice the mean run at raised priority
	for (int kind = 0; kind < 2; ++kind) {
		const std::vector<int> &pt = kind == 0 ? cptr : rptr;
		const int nrows = kind == 0 ? p->items : p->uc;
		long long ent = 0, rows = 0;
		int longest = 0;
		for (int r = 0; r < nrows; ++r) {
			const int len = pt[(size_t) r + 1] - pt[r];
			if (p->n_long[kind] > 0 && len >= p->long_len[kind]) continue;   // on the extreme-row path
			ent += len;
			++rows;
			longest = std::max(longest, len);
		}
		const long long mean = rows ? ent / rows : 0;
		p->prio_len[kind] = 0;
		if (p->cfg.sweep_prio > 0)
			p->prio_len[kind] = p->cfg.sweep_prio;
		else if (p->cfg.sweep_prio < 0 && rows > 256 && longest >= 4 * std::max<long long>(mean, 1))
			p->prio_len[kind] = (int) std::max<long long>(64, 2 * mean);
	}
	// ---- double-buffered single-wave form for the WHOLE single-wave launch: measured slower than the single-buffered form
	// whenever the launch has more rows than double-tile workgroups fit the chip (cfg3 uniform 0.222 -> 0.429 ms: the
	// second tile halves the resident workgroups and the CU's gather rate is shared by fewer requests in flight), so it
	// is off unless forced (MF_SWEEP_DB=1) or the launch is below MF_SWEEP_DB_ROWS rows (experiments build).  Its use is
	// the mid-length rows' launch above.
	for (int kind = 0; kind < 2; ++kind) {
		const int nrows = kind == 0 ? p->items : p->uc;
		const int launch_rows = p->n_long[kind] > 0 ? p->n_short[kind] : nrows;
		const int limit = p->cfg.db_rows;
		p->use_db[kind] = p->sweep.db && !p->coop_all[kind] && !p->rest_coop && launch_rows > 0 &&
		                  (p->cfg.sweep_db == 1 || (p->cfg.sweep_db < 0 && launch_rows <= limit));
		p->use_pair[kind] = !p->coop_all[kind] && !p->rest_coop && !p->use_db[kind] && launch_rows > 0 && pair_wanted(p, kind);
		// Trios (loader / phase-A / phase-B waves, mf_sweep.hip.h): experiments build only, MF_SWEEP_TRIO=1 (every side that
		// runs pairs) or MF_SWEEP_TRIO_U=1 (the user side alone).  No rule chooses them: on the one side they were meant for --
		// cfg3 power-law users, not split, longest row 2324 entries = 128 us of walk against 112 us of bytes -- the trio alone
		// is slower than the pair (0.1447 vs 0.1378 ms with the items on pairs; the 0.121 of the all-trio run was the item
		// side's ordered sums no longer overlapping the user sweep), and every throughput-bound side loses to the third tile.
		p->use_trio[kind] = p->use_pair[kind] && p->sweep.trio && (p->cfg.sweep_trio == 1 || (kind == 1 && p->cfg.sweep_trio == 2));
	}
	// ---- a sweep of a few thousand rows is a handful of rounds of workgroups: in index order its tail is whatever
	// long rows happen to start last.  Longest first (workgroups are dispatched in list order) the tail is made of the
	// shortest rows.  cfg3 uniform (3952 / 6040 rows of 50..311 entries): see DESIGN 5.1.  Large sweeps keep the index
	// order (the tail is a negligible part of them and neighbouring rows share lines of the entry arrays) except for rows
	// several times longer than the average, which lead the list.
	if (!p->cfg.nosort)
		for (int kind = 0; kind < 2; ++kind) {
			const std::vector<int> &pt = kind == 0 ? cptr : rptr;
			const int nrows = kind == 0 ? p->items : p->uc;
			if (p->n_long[kind] > 0 || p->coop_all[kind] || nrows < 512) continue;
			auto len = [&](int r) { return pt[(size_t) r + 1] - pt[r]; };
			auto longer = [&](int x, int y) { return len(x) > len(y); };
			std::vector<int> order;
			order.reserve((size_t) nrows);
			if (nrows <= (1 << 15)) {   // a dozen rounds of workgroups at most: the tail matters, the order of the row reads does not
				for (int r = 0; r < nrows; ++r) order.push_back(r);
				std::stable_sort(order.begin(), order.end(), longer);
			} else {
				// a large sweep with a few very long rows (power-law users): only those move to the front
				const long long mean = p->nnz / nrows;
				if ((long long) p->max_row_len[kind] < 8 * std::max<long long>(mean, 1)) continue;
				std::vector<int> head;
				for (int r = 0; r < nrows; ++r) (len(r) >= 4 * mean ? head : order).push_back(r);
				std::stable_sort(head.begin(), head.end(), longer);
				order.insert(order.begin(), head.begin(), head.end());
			}
			MF_TRY(dev_alloc(&p->short_rows[kind], order.size()));
			MF_TRY_HIP(h2d(p, p->short_rows[kind], order.data(), order.size() * sizeof(int)));
			p->lpt[kind] = true;
		}
	return MF_OK;
}

// Tables of the errors + streams iteration (mf_stream.hip.h): the CSR rows cut into segments of at most es_nch
// entries (one wave each in the errors launch) and the workgroup table of the streams launch (mf_resident.hip.h).
int plan_es_schedule(mf_plan *p, const std::vector<int> &rptr, const std::vector<int> &cptr)
{
	p->es_mode = false;
	if (!p->want_map || !p->csr2csc) return MF_OK;
	const size_t row_bytes = (size_t) p->sweep.row_bytes, head = (size_t) p->sweep.xs_bytes;
	int nch = (int) std::min<size_t>(64, (kLdsPerCu / 3 - head) / row_bytes);
	if (const int v = p->cfg.es_nch; v >= 1 && v <= 64 && head + (size_t) v * row_bytes <= kLdsPerCu) nch = v;
	if (nch < 1) return MF_OK;
	p->es_nch = nch;
	p->es_lds_errors = head + (size_t) nch * row_bytes;
	std::vector<int> srow, sbeg, send;
	for (int u = 0; u < p->uc; ++u)
		for (int c = rptr[(size_t) u]; c < rptr[(size_t) u + 1]; c += nch) {
			srow.push_back(u);
			sbeg.push_back(c);
			send.push_back(std::min(rptr[(size_t) u + 1], c + nch));
		}
	int ncu = 256;
	{
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
	}
	p->es_nseg = (int) srow.size();
	if (p->es_nseg == 0 || p->res_sw <= 0) return MF_OK;
	MF_TRY(dev_alloc(&p->es_seg_row, srow.size()));
	MF_TRY(dev_alloc(&p->es_seg_beg, srow.size()));
	MF_TRY(dev_alloc(&p->es_seg_end, srow.size()));
	MF_TRY(dev_alloc(&p->rec_csr, (size_t) p->nnz + 64));
	MF_TRY(dev_alloc(&p->rec_csc, (size_t) p->nnz + 64));
	MF_TRY_HIP(h2d(p, p->es_seg_row, srow.data(), srow.size() * sizeof(int)));
	MF_TRY_HIP(h2d(p, p->es_seg_beg, sbeg.data(), srow.size() * sizeof(int)));
	MF_TRY_HIP(h2d(p, p->es_seg_end, send.data(), srow.size() * sizeof(int)));
	// records = {idx (fixed), pad, err (rewritten every iteration)}; the 64 entries of slack behind the last one are
	// read (never used) by the streams launch's 64-wide chunk loads
	// on the plan's own stream: it is a non-blocking stream, NOT ordered with the null stream a plain hipMemset runs on
	MF_TRY_HIP(hipMemsetAsync(p->rec_csr, 0, ((size_t) p->nnz + 64) * sizeof(mf::StreamRec), p->stream));
	MF_TRY_HIP(hipMemsetAsync(p->rec_csc, 0, ((size_t) p->nnz + 64) * sizeof(mf::StreamRec), p->stream));
	{
		const unsigned grid = (unsigned) ((p->nnz + 255) / 256);
		hipLaunchKernelGGL(fill_records_kernel, dim3(grid), dim3(256), 0, p->stream, p->csr_idx, p->nnz, p->rec_csr);
		hipLaunchKernelGGL(fill_records_kernel, dim3(grid), dim3(256), 0, p->stream, p->csc_idx, p->nnz, p->rec_csc);
		MF_TRY_HIP(hipGetLastError());
		MF_TRY_HIP(hipStreamSynchronize(p->stream));
	}
	MF_TRY_HIP(raise_lds_limit((const void *) p->sweep.errs, p->es_lds_errors));
	// ---- LDS-resident streams (mf_resident.hip.h): ~one workgroup per CU; every (side, slice) gets `per` workgroups
	// of eight waves, and the side's rows are cut into per * 8 runs of consecutive rows balanced by cost
	{
		const int sw = p->res_sw, nsl = (p->K + sw - 1) / sw;
		const int per = std::max(1, ncu / (2 * nsl));
		// waves of a workgroup that own rows (the others only help with the slice copy): MF_ES_ACTIVE
		const int aw = p->cfg.es_active >= 1 && p->cfg.es_active <= mf::kResidentWaves ? p->cfg.es_active : mf::kResidentWaves;
		std::vector<mf::SliceWg> wgs;
		for (int side = 0; side < 2; ++side) {
			const std::vector<int> &pt = side == 0 ? cptr : rptr;
			const int nrows = side == 0 ? p->items : p->uc;
			// run boundaries: greedy on cost = entries + 16 per row; a run never splits a row and never holds more than
			// kResidentRows rows (its row pointers live in one register), so a side of many short rows gets more
			// workgroups than one per CU and slice
			std::vector<int> cut(1, 0);
			{
				const int target_runs = per * aw;
				const double row_cost = p->cfg.es_row_cost > 0 ? (double) p->cfg.es_row_cost : 16.0;   // entries a row end is worth
				const double total_cost = (double) pt[(size_t) nrows] + row_cost * nrows;
				double acc_cost = 0, done = 0;
				int in_run = 0;
				for (int r = 0; r < nrows; ++r) {
					acc_cost += (pt[(size_t) r + 1] - pt[(size_t) r]) + row_cost;
					++in_run;
					const int left = target_runs - (int) cut.size();
					const bool share = left > 0 && acc_cost >= (total_cost - done) / (left + 1);
					if (r + 1 < nrows && (share || in_run == mf::kResidentRows)) {
						cut.push_back(r + 1);
						done += acc_cost;
						acc_cost = 0;
						in_run = 0;
					}
				}
				cut.push_back(nrows);
				while (((int) cut.size() - 1) % aw != 0) cut.push_back(nrows);
			}
			const int nwg_side = ((int) cut.size() - 1) / aw;
			for (int sl = 0; sl < nsl; ++sl)
				for (int w = 0; w < nwg_side; ++w) {
					mf::SliceWg g;
					g.side = side;
					g.slice = sl;
					for (int i = 0; i <= mf::kResidentWaves; ++i) {
						g.row_beg[i] = cut[(size_t) (w * aw + std::min(i, aw))];
						g.ent_beg[i] = pt[(size_t) g.row_beg[i]];
					}
					if (g.row_beg[mf::kResidentWaves] > g.row_beg[0]) wgs.push_back(g);
				}
		}
		p->res_nwg = (int) wgs.size();
		p->res_lds = (size_t) std::max(p->uc, p->items) * sw * 8 + mf::kResidentWaves * mf::kResidentWaveLds;
		if (p->res_nwg > 0) {
			MF_TRY(dev_alloc(&p->res_wg, wgs.size()));
			MF_TRY_HIP(h2d(p, p->res_wg, wgs.data(), wgs.size() * sizeof(mf::SliceWg)));
			const void *fn = sw == 8   ? (const void *) mf::stream_resident_kernel<8>
			                 : sw == 4 ? (const void *) mf::stream_resident_kernel<4>
			                           : (const void *) mf::stream_resident_kernel<2>;
			MF_TRY_HIP(raise_lds_limit(fn, p->res_lds));
		}
	}
	p->es_mode = true;
	return MF_OK;
}

#undef MF_TRY
#undef MF_TRY_HIP

}  // namespace
