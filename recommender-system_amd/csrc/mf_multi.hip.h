// mf_multi.hip.h -- several shards in one process (mf_backend_run_multi); included inside extern "C".

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

