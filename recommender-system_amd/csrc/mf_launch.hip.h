// mf_launch.hip.h -- choice of the sweep kernel variant / chunk size and the launch of one sweep.
#pragma once
#include <climits>

namespace {

// Slice width (columns) of the LDS-resident streams launch, 0 when a slice of the larger factor does not fit the
// LDS of a CU beside the waves' buffers or K would need more than eight slices (every slice re-reads the records).
int resident_slice_width(const mf_config &cfg, int K, int yrows_max)
{
	if (K & 1) return 0;
	for (int sw : {cfg.es_sw ? cfg.es_sw : 8, 8, 4, 2})   // MF_ES_SW: slice width to try first (A/B)
		if ((sw == 8 || sw == 4 || sw == 2) && (K + sw - 1) / sw <= 8 && (size_t) yrows_max * sw * 8 + mf::kResidentWaves * mf::kResidentWaveLds + 1024 <= kLdsPerCu) return sw;
	return 0;
}

// Row pitch (doubles) of a factor buffer the plan owns: 8K bytes rounded up to whole 128-byte lines when that saves
// at least a tenth of the lines a gathered row touches on average (rows start wherever 8K * r falls: a row of B bytes
// touches B/128 + 1 - gcd(B, 128)/128 lines).  LDS-DMA forms only; MF_ROW_PITCH=0 keeps K (A/B).
int row_pitch(const mf_config &cfg, int K, bool dma)
{
	if (!dma || !cfg.row_pitch) return K;
	const int bytes = 8 * K;
	if (bytes % 128 == 0) return K;
	int g = 128, b = bytes;
	while (b) {
		const int t = g % b;
		g = b;
		b = t;
	}
	const double plain = bytes / 128.0 + 1.0 - g / 128.0, padded = (bytes + 127) / 128;
	return plain >= 1.1 * padded ? ((bytes + 127) / 128) * 16 : K;
}

// does this K run on an LDS-DMA form of the sweep (even K up to 1024, unless MF_SWEEP_IMPL=reg)?  Mirrors choose_sweep.
bool sweep_is_dma(const mf_config &cfg, int K)
{
	return !cfg.sweep_reg && (K & 1) == 0 && K >= 2 && K <= 128 * 8;
}

int choose_sweep(mf_plan *p)
{
	const int K = p->K;
	p->sweep = SweepVariant{nullptr, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	const bool allow_dma = !p->cfg.sweep_reg;   // MF_SWEEP_IMPL=dma (default) | reg: register-staged form only
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
	if (!p->cfg.sweep_pf) p->sweep.pf = nullptr;   // MF_SWEEP_PF=0 (experiments build): round 2's accumulate form everywhere

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
	if (const int v = p->cfg.sweep_nch; v >= 1 && head + (size_t) v * row_bytes <= kLdsPerCu) nch = v;
	if (nch < 1) return MF_ERR_UNSUPPORTED;
	p->nch = nch;
	p->lds_bytes = head + (size_t) nch * row_bytes;
	// A sweep over FEW rows (ML100k: 943 x 1682) cannot fill 256 CUs whatever the chunk size; its time is the
	// longest row's serial chain of chunks, so use the largest chunk there (737 entries: 47 -> 12 chunks).
	int few = std::max(nch, std::min(64, fit(kLdsPerCu / 2)));
	if (p->cfg.sweep_nch) few = nch;
	p->nch_few = few;
	p->lds_bytes_few = head + (size_t) few * row_bytes;
	MF_HIP(raise_lds_limit((const void *) p->sweep.fn, (size_t) (std::max(p->lds_bytes, p->lds_bytes_few))));
	if (p->sweep.pf) MF_HIP(raise_lds_limit((const void *) p->sweep.pf, (size_t) (std::max(p->lds_bytes, p->lds_bytes_few))));
	if (p->sweep.pair) {   // wave-pair form: two tiles
		// 32-entry chunks: the loader's ~90 cycles per gathered row are what a pair is bound by, the K steps of phase A are
		// paid per chunk -- a lone 5993-entry row: 0.526 ms at 16, 0.332 at 32; cfg3 power-law 0.311 / 0.268 / 0.314 at 24 / 32 / 40
		int npr = p->cfg.pair_nch > 0 ? p->cfg.pair_nch : 32;
		if (p->cfg.sweep_nch) npr = p->cfg.sweep_nch;
		while (npr > 1 && head + 2 * (size_t) npr * row_bytes > kLdsPerCu / 2) --npr;
		p->nch_pair = npr;
		p->lds_bytes_pair = head + 2 * (size_t) npr * row_bytes;
		p->pair_loaders = p->cfg.pair_loaders == 2 && p->sweep.pair2 ? 2 : 1;
		if (p->pair_loaders == 2) p->sweep.pair = p->sweep.pair2;
		p->pair_waves = p->pair_loaders + 1;
		MF_HIP(raise_lds_limit((const void *) p->sweep.pair, p->lds_bytes_pair));
		if (p->sweep.trio) {   // loader / phase-A / phase-B waves: three tiles + the errors of two chunks
			int ntr = p->cfg.pair_nch > 0 ? p->cfg.pair_nch : 32;
			if (p->cfg.sweep_nch) ntr = p->cfg.sweep_nch;
			while (ntr > 1 && head + 1024 + 3 * (size_t) ntr * row_bytes > kLdsPerCu / 2) --ntr;
			p->nch_trio = ntr;
			p->lds_bytes_trio = head + 1024 + 3 * (size_t) ntr * row_bytes;
			MF_HIP(raise_lds_limit((const void *) p->sweep.trio, p->lds_bytes_trio));
		}
	}
	// double-buffered form (few rows per CU: the wave hides its own gather): two tiles of nch_db rows
	if (p->sweep.db) {
		int ndb = p->cfg.db_nch > 0 ? p->cfg.db_nch : 16;
		if (p->cfg.sweep_nch) ndb = p->cfg.sweep_nch;
		while (ndb > 1 && head + 2 * (size_t) ndb * row_bytes > kLdsPerCu / 2) --ndb;
		p->nch_db = ndb;
		p->lds_bytes_db = head + 2 * (size_t) ndb * row_bytes;
		MF_HIP(raise_lds_limit((const void *) p->sweep.db, p->lds_bytes_db));
	}
	// ---- errors + streams iteration (mf_stream.hip.h) for instances whose factors stay in L2 / Infinity Cache: the
	// two sweeps are then bound by the latency of one wave walking a row chunk by chunk, not by bandwidth.  It costs a
	// third gather of every entry's row, so it is only chosen while the factors are cache-resident; MF_ITER_MODE=es |
	// sweeps overrides, MF_ES_MAX_MB moves the limit.
	p->want_map = false;
	p->res_sw = resident_slice_width(p->cfg, K, std::max(p->uc, p->items));
	if (p->sweep.errs && p->nnz > 0) {
		// Used where the streams launch can keep a slice of Y resident in LDS (mf_resident.hip.h; instML100k 84 -> 39 us
		// per iteration); MF_ITER_MODE=sweeps keeps the two sweeps, =es asks for it explicitly (same condition).
		// Not below a few thousand entries: there two graph-replayed single-wave sweeps are quicker than a launch that
		// first copies a slice of Y into every CU's LDS (inst30-40, 170 entries: 17 vs 20 us per iteration).
		const bool forced = p->cfg.iter_mode == mf_config::kIterEs;
		p->want_map = p->res_sw > 0 && p->cfg.iter_mode != mf_config::kIterSweeps && (forced || p->nnz >= 4096);
	}
	return MF_OK;
}


// defer_join: leave the ordered sums of the extreme rows running on the side stream when the call returns
// (p->join_pending); the caller joins before anything reads the new generation.
int launch_sweep(mf_plan *p, int kind, int seed, bool defer_join = false)
{
	mf::SweepArgs a;
	a.K = p->K;
	a.nch = p->nch;
	a.stride = p->stride;
	a.seed = seed;
	a.prio_len = p->prio_len[kind];
	a.c2 = p->alpha * 2;
	const int nxt = p->cur ^ 1;
	a.ldx = kind == 0 ? p->ldr : p->ldl;
	a.ldy = kind == 0 ? p->ldl : p->ldr;
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
	a.rowlist = p->lpt[kind] ? p->short_rows[kind] : nullptr;
	a.seg_row = a.seg_beg = a.seg_end = nullptr;
	a.seg_out = nullptr;
	a.scratch = nullptr;
	a.scratch_entries = 0;
	if (a.nrows <= 0) return MF_OK;
	// "few rows": the launch cannot fill the chip whatever the chunk size, its time is the longest row's serial chain
	// of chunks -> the largest chunk.  Only below ~2048 rows: at 3952 rows (the cfg3 item sweep) the large chunk's LDS
	// footprint cost more occupancy than it saved (item sweep 0.189 -> 0.123 ms with the ordinary chunk).
	const bool few_rows = a.nrows < p->cfg.sweep_few;
	const bool coop = p->coop_all[kind];
	const bool db = !coop && p->use_db[kind];
	const bool pair = !coop && !db && p->use_pair[kind];
	const bool trio = pair && p->use_trio[kind] && p->sweep.trio;
	if (few_rows) a.nch = coop ? p->nch_coop : p->nch_few;
	if (db) a.nch = p->nch_db;
	if (pair) a.nch = trio ? p->nch_trio : p->nch_pair;
	const size_t lds = coop ? p->lds_bytes_coop : db ? p->lds_bytes_db : trio ? p->lds_bytes_trio : pair ? p->lds_bytes_pair : (few_rows ? p->lds_bytes_few : p->lds_bytes);
	// Accumulate form of the single-wave launch.  Up to kPfRows rows: the form whose phases keep their LDS reads in flight
	// and whose gather issue is lean -- what a wave walking a long row alone is bound by (cfg3 uniform 0.224 -> 0.201 ms,
	// power-law 0.367 -> 0.350, a lone 5993-entry row 1.02 -> 0.78 ms).  Larger launches of one-pass rows (K <= 128) are
	// never bound by one wave and keep round 2's form (cfg4 user sweep, 1e6 rows: 11.45 vs 11.60 ms).  Two-pass rows
	// (K = 256) take it at every size: their phase A is a 256-deep chain per chunk that the six resident workgroups of a
	// CU do not hide, and keeping its LDS reads in flight shortens it (cfg5: 346.6 -> 327.9 ms, 0.745 -> 0.787).
	const int kPfRows = p->cfg.pf_rows > 0 ? p->cfg.pf_rows : (p->K > 128 ? INT_MAX : 262144);
	const SweepFn single = p->sweep.pf && p->n_short[kind] <= kPfRows && a.nrows <= kPfRows ? p->sweep.pf : p->sweep.fn;
	const SweepFn fn = coop ? p->sweep.coop : db ? p->sweep.db : trio ? p->sweep.trio : pair ? p->sweep.pair : single;
	const int block = coop ? mf::kCoopWaves * mf::kWave : trio ? 3 * mf::kWave : pair ? p->pair_waves * mf::kWave : mf::kWave;
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
		// extreme rows: products kernel over their 256-entry segments -> ordered sum per (row, column slice),
		// beside the sweep of the other rows (schedule below)
		mf::SweepArgs b = a;
		b.nrows = p->n_seg[kind];
		b.prio_len = 0;
		b.rowlist = nullptr;
		b.nch = p->nch_prod;
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
		o.ldx = a.ldx;
		o.seed = seed;
		o.nslices = (p->K + mf::kSliceCols - 1) / mf::kSliceCols;
		o.row = p->long_rows[kind];
		o.sbeg = p->lr_sbeg[kind];
		o.cnt = p->lr_cnt[kind];
		o.scratch = p->scratch;
		o.scratch_entries = p->scratch_entries;
		o.X_old = a.X_old;
		o.X_new = a.X_new;
		o.stamps = nullptr;
		o.max_cnt = p->max_row_len[kind];
		void *oargs[] = {&o};
		// Schedule (MF_SWEEP_SUM_ORDER): "after" (default) -- products kernel and ordered sums on the side stream
		// while the remaining rows run on the main stream; "under" -- products first on the main stream, then the
		// ordered sums on the side stream under the sweep of the remaining rows.
		const bool under = p->cfg.sum_under;
		hipStream_t prod_stream = under ? p->stream : p->side_stream;
		if (!under) {
			MF_HIP(hipEventRecord(p->ev_fork, p->stream));
			MF_HIP(hipStreamWaitEvent(p->side_stream, p->ev_fork, 0));
		}
		MF_HIP(hipLaunchKernel((const void *) p->sweep.prod, dim3(b.nrows), dim3(mf::kWave), bargs, p->lds_bytes_prod,
		                       prod_stream));
		if (under) {
			MF_HIP(hipEventRecord(p->ev_fork, p->stream));
			MF_HIP(hipStreamWaitEvent(p->side_stream, p->ev_fork, 0));
		}
		MF_HIP(hipLaunchKernel(p->cfg.os_dpp ? (const void *) mf::ordered_sum_kernel<true> : (const void *) mf::ordered_sum_kernel<false>,
		                       dim3(o.nrows * o.nslices), dim3(mf::kWave), oargs,
		                       p->lds_bytes_osum, p->side_stream));
		MF_HIP(hipEventRecord(p->ev_join, p->side_stream));
		if (p->n_mid[kind] > 0) {
			// mid-length rows: double-buffered form with the large chunk, on its own stream beside everything else
			mf::SweepArgs m = a;
			m.nrows = p->n_mid[kind];
			m.rowlist = p->mid_rows[kind];
			m.nch = p->nch_mid;
			void *margs[] = {&m};
			MF_HIP(hipStreamWaitEvent(p->mid_stream, p->ev_fork, 0));
			if (p->mid_coop)
				MF_HIP(hipLaunchKernel((const void *) p->sweep.coop, dim3(m.nrows), dim3(mf::kCoopWaves * mf::kWave), margs, p->lds_bytes_mid, p->mid_stream));
			else
			MF_HIP(hipLaunchKernel((const void *) p->sweep.db, dim3(m.nrows), dim3(mf::kWave), margs, p->lds_bytes_mid, p->mid_stream));
			MF_HIP(hipEventRecord(p->ev_mid_join, p->mid_stream));
		}
		a.nrows = p->n_short[kind];
		a.rowlist = p->short_rows[kind];
		a.nch = p->nch;   // the extreme rows are gone: the occupancy-friendly chunk size is right again
		if (a.nrows > 0 && p->rest_coop) {
			a.nch = p->nch_coop;
			MF_HIP(hipLaunchKernel((const void *) p->sweep.coop, dim3(std::min(a.nrows, 1 << 20)),
			                       dim3(mf::kCoopWaves * mf::kWave), args, p->lds_bytes_coop, p->stream));
		} else if (a.nrows > 0 && pair) {
			a.nch = trio ? p->nch_trio : p->nch_pair;
			MF_HIP(hipLaunchKernel((const void *) (trio ? p->sweep.trio : p->sweep.pair), dim3(std::min(a.nrows, 1 << 20)),
			                       dim3((trio ? 3 : p->pair_waves) * mf::kWave), args, trio ? p->lds_bytes_trio : p->lds_bytes_pair, p->stream));
		} else if (a.nrows > 0 && db) {
			a.nch = p->nch_db;
			MF_HIP(hipLaunchKernel((const void *) p->sweep.db, dim3(std::min(a.nrows, 1 << 20)), dim3(mf::kWave), args,
			                       p->lds_bytes_db, p->stream));
		} else if (a.nrows > 0)
			MF_HIP(hipLaunchKernel((const void *) single, dim3(std::min(a.nrows, 1 << 20)), dim3(mf::kWave), args,
			                       p->lds_bytes, p->stream));
		if (p->n_mid[kind] > 0) MF_HIP(hipStreamWaitEvent(p->stream, p->ev_mid_join, 0));
		if (defer_join)
			p->join_pending = true;
		else
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

// One iteration in the errors + streams form: errors launch over the CSR segments (e_n in CSR and CSC order), then
// ONE streams launch that adds up both factors' rows (timed as kind 0 / kind 1 like the two sweeps).
int launch_es_iteration(mf_plan *p)
{
	const int nxt = p->cur ^ 1;
	mf::SweepArgs a;
	memset(&a, 0, sizeof a);
	a.nrows = p->es_nseg;
	a.K = p->K;
	a.nch = p->es_nch;
	a.stride = p->stride;
	a.ldx = p->ldl;
	a.ldy = p->ldr;
	a.seed = 1;
	a.c2 = p->alpha * 2;
	a.ptr = p->csr_ptr;
	a.idx = p->csr_idx;
	a.val = p->csr_val;
	a.X_old = p->Lbuf[p->cur];
	a.Y_old = p->Rbuf[p->cur];
	a.X_new = nullptr;
	a.seg_row = p->es_seg_row;
	a.seg_beg = p->es_seg_beg;
	a.seg_end = p->es_seg_end;
	a.err_a = reinterpret_cast<double *>(p->rec_csr);
	a.err_b = reinterpret_cast<double *>(p->rec_csc);
	a.map = p->csr2csc;
	mf::SliceArgs ra;
	ra.K = p->K;
	ra.wg = p->res_wg;
	ra.ptr[0] = p->csc_ptr;
	ra.ptr[1] = p->csr_ptr;
	ra.side[0] = mf::StreamSide{p->rec_csc, p->Rbuf[p->cur], p->Lbuf[p->cur], p->Rbuf[nxt]};
	ra.side[1] = mf::StreamSide{p->rec_csr, p->Lbuf[p->cur], p->Rbuf[p->cur], p->Lbuf[nxt]};
	ra.yrows[0] = p->uc;
	ra.yrows[1] = p->items;
	ra.ldx[0] = p->ldr;   // side 0: X = R, Y = L
	ra.ldy[0] = p->ldl;
	ra.ldx[1] = p->ldl;
	ra.ldy[1] = p->ldr;
	TimedLaunch t0{}, t1{};
	if (p->timing) {
		MF_HIP(hipEventCreate(&t0.t0));
		MF_HIP(hipEventCreate(&t0.t1));
		MF_HIP(hipEventCreate(&t1.t1));
		t0.kind = 0;
		t1.kind = 1;
		MF_HIP(hipEventRecord(t0.t0, p->stream));
	}
	void *eargs[] = {&a};
	MF_HIP(hipLaunchKernel((const void *) p->sweep.errs, dim3(p->es_nseg), dim3(mf::kWave), eargs, p->es_lds_errors,
	                       p->stream));
	if (p->timing) MF_HIP(hipEventRecord(t0.t1, p->stream));
	{
		void *rargs[] = {&ra};
		const void *fn = p->res_sw == 8   ? (const void *) mf::stream_resident_kernel<8>
		                 : p->res_sw == 4 ? (const void *) mf::stream_resident_kernel<4>
		                                  : (const void *) mf::stream_resident_kernel<2>;
		MF_HIP(hipLaunchKernel(fn, dim3(p->res_nwg), dim3(mf::kResidentThreads), rargs, p->res_lds, p->stream));
	}
	if (p->timing) {
		MF_HIP(hipEventRecord(t1.t1, p->stream));
		t1.t0 = t0.t1;
		t1.shared_start = true;
		p->timed.push_back(t0);
		p->timed.push_back(t1);
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
		if (!t.shared_start) (void) hipEventDestroy(t.t0);
	}
	for (auto &t : p->timed) (void) hipEventDestroy(t.t1);
	p->timed.clear();
	return MF_OK;
}

}  // namespace
