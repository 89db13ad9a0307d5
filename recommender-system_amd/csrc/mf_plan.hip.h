// mf_plan.hip.h -- error macro, kernel variant tables and the resident plan (struct mf_plan).
#pragma once

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

// hipFuncAttributeMaxDynamicSharedMemorySize is per FUNCTION, not per plan: two live plans that share a kernel
// instance (the run-time-K forms) but need different tile sizes must never lower each other's limit.
inline hipError_t raise_lds_limit(const void *fn, size_t bytes)
{
	static std::mutex mu;
	static std::map<std::pair<int, const void *>, size_t> limit;   // the attribute is kept per device
	std::lock_guard<std::mutex> lock(mu);
	int dev = 0;
	(void) hipGetDevice(&dev);
	size_t &cur = limit[std::make_pair(dev, fn)];
	if (bytes <= cur) return hipSuccess;
	const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
	if (e == hipSuccess) cur = bytes;
	return e;
}

struct SweepVariant {
	SweepFn fn;
	int kt;         // compile-time K, 0 = runtime K
	int kpmax;      // 64-column groups held in registers (register-staged form)
	int dma;        // 1: LDS-DMA form
	int row_bytes;  // LDS tile row stride in bytes (DMA form)
	int xs_bytes;   // LDS bytes in front of the tile (DMA form)
	SweepFn coop;   // row-cooperative form for tiny sweeps (compile-time-K DMA variants only)
	SweepFn prod;   // products form for segments of extreme rows (all DMA variants)
	SweepFn errs;   // errors form: e_n per entry of a segment (all DMA variants; mf_stream.hip.h)
	SweepFn db;     // intra-wave double-buffered form for launches of few rows (all DMA variants)
	SweepFn pf;     // accumulate form with the LDS reads of phases A / B kept in flight (compile-time-K DMA variants)
	SweepFn pair;   // wave-pair form (loader + compute) for launches that end on long rows (64 <= K <= 128, compile-time K)
	SweepFn pair2;  // ... with two loader waves
	SweepFn trio = nullptr;   // ... with the compute wave split into a phase-A and a phase-B wave (three waves per row)
};

template <int KT, int KP>
constexpr SweepVariant variant()
{
	return SweepVariant{mf::sweep_kernel<KT, KP>, KT, KP, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
}

template <int KT, int NL>
constexpr SweepFn pair_fn()
{
	if constexpr (mf::DmaGeom<KT>::kPasses == 1 && (mf::DmaGeom<KT>::kPieces | 1) > 32)
		return mf::sweep_pair_kernel<KT, NL>;
	else
		return nullptr;
}
template <int KT>
constexpr SweepFn trio_fn()
{
	if constexpr (mf::DmaGeom<KT>::kPasses == 1 && (mf::DmaGeom<KT>::kPieces | 1) > 32)
		return mf::sweep_trio_kernel<KT>;
	else
		return nullptr;
}

template <int KT>
constexpr SweepVariant dma_variant()
{
	return SweepVariant{mf::sweep_dma_kernel<KT, mf::DmaGeom<KT>::kPasses>, KT, 0, 1, mf::DmaGeom<KT>::kStride,
	                    mf::DmaGeom<KT>::kXsBytes, mf::sweep_coop_kernel<KT>,
	                    mf::sweep_dma_kernel<KT, mf::DmaGeom<KT>::kPasses, mf::kSweepProducts>,
	                    mf::sweep_dma_kernel<KT, mf::DmaGeom<KT>::kPasses, mf::kSweepErrors>,
	                    mf::sweep_db_kernel<KT, mf::DmaGeom<KT>::kPasses>,
	                    mf::sweep_dma_kernel<KT, mf::DmaGeom<KT>::kPasses, mf::kSweepAccumulate, 8>, pair_fn<KT, 1>(),
#ifdef MF_EXPERIMENTS
	                    pair_fn<KT, 2>(),   // two loader waves: measured within noise of one (the compute wave is the bound)
	                    trio_fn<KT>()       // loader / phase-A / phase-B waves: measured slower wherever it was tried
#else
	                    nullptr, nullptr
#endif
	};
}

// run-time even K <= 128 * NPASS through the LDS-DMA kernel (row_bytes / xs_bytes filled in per plan)
template <int NPASS>
constexpr SweepVariant dma_generic_variant()
{
	return SweepVariant{mf::sweep_dma_kernel<0, NPASS>, 0, NPASS, 1, 0, 0, nullptr,
	                    mf::sweep_dma_kernel<0, NPASS, mf::kSweepProducts>, mf::sweep_dma_kernel<0, NPASS, mf::kSweepErrors>,
	                    mf::sweep_db_kernel<0, NPASS>, nullptr, nullptr, nullptr};
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
	int kind;            // 0 item sweep / errors launch, 1 user sweep / streams launch
	bool shared_start;   // t0 is the previous record's t1 (not owned)
};

}  // namespace

struct mf_plan {
	mf_config cfg;   // the environment switches, read once at creation (mf_config.hip.h)
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
	int *mask_idx = nullptr;   // item ids ascending inside every user's row, only when the file order is not (recommend mask)
	// errors + streams iteration (mf_stream.hip.h): CSR position -> CSC position, the {idx, e_n} records in both orders, the segment table
	// of the errors launch and the task list (both factors' rows, longest first) of the streams launch
	bool want_map = false, es_mode = false;
	int *csr2csc = nullptr;
	mf::StreamRec *rec_csr = nullptr, *rec_csc = nullptr;
	int es_nseg = 0, es_nch = 0;
	size_t es_lds_errors = 0;
	int *es_seg_row = nullptr, *es_seg_beg = nullptr, *es_seg_end = nullptr;
	// streams launch with a column slice of Y resident in LDS (mf_resident.hip.h): small factor matrices only
	int res_sw = 0, res_nwg = 0;
	size_t res_lds = 0;
	mf::SliceWg *res_wg = nullptr;

	double *Lbuf[2] = {nullptr, nullptr};
	double *Rbuf[2] = {nullptr, nullptr};
	int ldl = 0, ldr = 0;   // row pitch of the L and R buffers in doubles (K, or K padded to whole 128-byte lines)
	bool r_external = false;
	bool l_external = false;
	bool join_pending = false;          // ordered sums of the last item sweep still run on the side stream
	mf_candidate *cand_dev = nullptr;   // recommend_scored output, allocated on first use
	mf_candidate *cand_pack = nullptr;  // the listed users' records in list order (recommend_scored_users)
	mf_filter *filt_dev = nullptr;      // recommend_filter output, allocated on first use
	mf_filter *part_dev = nullptr;      // per-split reports of a small recommendation (nsplit x users), allocated on first use
	int part_cap = 0;
	bool rec_half_used = false;   // the last MFMA pass ran as 64-user workgroups, two per CU (recommend_mfma2_kernel)
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
	int nch_db = 0;             // chunk size and LDS request (two tiles) of the double-buffered form
	size_t lds_bytes_db = 0;
	bool use_db[2] = {false, false};   // the sweep's single-wave launch takes the double-buffered form (plan_row_schedule)
	bool use_pair[2] = {false, false}; // ... or the wave-pair form
	int nch_pair = 0;
	int pair_waves = 2;   // waves per row of the pair form in use: 2 (loader + compute), 3 with two loaders
	bool use_trio[2] = {false, false};   // ... that side's pairs as loader / phase-A / phase-B trios (three tiles)
	int nch_trio = 0;
	size_t lds_bytes_trio = 0;
	int pair_loaders = 1;   // loader waves of the wave-pair form
	size_t lds_bytes_pair = 0;
	// mid-length rows of a skewed sweep (below the extreme threshold, far above the mean): their own launch of the
	// double-buffered form with a LARGE chunk on a second side stream -- a wave that keeps 48 rows in flight gets a
	// matching share of its CU's gather rate instead of 1/11 of it, so the walk of the longest remaining row no longer
	// sets the sweep's time
	int *mid_rows[2] = {nullptr, nullptr};
	int n_mid[2] = {0, 0};
	int nch_mid = 0;
	size_t lds_bytes_mid = 0;
	bool mid_coop = false;   // the mid-length rows go through the row-cooperative kernel (8 waves per row) instead
	hipStream_t mid_stream = nullptr;
	hipEvent_t ev_mid_join = nullptr;
	int max_row_len[2] = {0, 0}; // longest column (item sweep) / longest user row (user sweep)
	int prio_len[2] = {0, 0};    // rows at least this long run at raised wave priority in the single-wave launch (0: none)
	// skew-aware split of a sweep with many rows: rows whose serial walk would dominate the launch go to the
	// row-cooperative kernel on a side stream, the others stay on the single-wave kernel
	int *long_rows[2] = {nullptr, nullptr}, *short_rows[2] = {nullptr, nullptr};
	bool lpt[2] = {false, false};   // short_rows[kind] = ALL rows, longest first: the order of a sweep without extreme rows
	int n_long[2] = {0, 0}, n_short[2] = {0, 0};
	int long_len[2] = {0, 0};   // a row at least this long is on the extreme-row path (when n_long > 0)
	// extreme rows of LARGE sweeps: 256-entry segments -> scaled rows in `scratch` -> ordered sum
	int n_seg[2] = {0, 0};
	int *seg_row[2] = {nullptr, nullptr}, *seg_beg[2] = {nullptr, nullptr}, *seg_end[2] = {nullptr, nullptr};
	long long *seg_out[2] = {nullptr, nullptr}, *lr_sbeg[2] = {nullptr, nullptr};
	int *lr_cnt[2] = {nullptr, nullptr};
	double *scratch = nullptr;
	size_t scratch_entries = 0;
	// tiny sweeps (a few us of data): ONE cooperative launch over all rows; a fork/join costs more than it saves
	int nch_coop = 0;
	bool rest_coop = false;     // rows beside the extreme ones go through the cooperative kernel (experiment)
	int nch_prod = 0;           // chunk size of the products launch (extreme rows)
	size_t lds_bytes_prod = 0;
	size_t lds_bytes_osum = 0;  // LDS request of ordered_sum_kernel (ring + padding that bounds the waves per CU)
	size_t lds_bytes_coop = 0;
	bool coop_all[2] = {false, false};
	hipStream_t side_stream = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;

	bool timing = false;
	std::vector<TimedLaunch> timed;
	int64_t acc_launch[2] = {0, 0};
	double acc_ms[2] = {0.0, 0.0};
};

