// mf_multi.hip.h -- several shards in one process (mf_backend_run_multi).
//
// The decomposition of matFact-mpi.c:155-214 with the grid create_balanced_grid (mpiutil.c:54-88) picks when one side
// is much longer: the factor with more rows ("A": users, or items when items > users) is cut into ndev contiguous
// blocks balanced by entry count and stays private; the other factor ("B") is replicated and summed after every B
// sweep -- either by the hand-written peer-to-peer reduce over xGMI (peer_allreduce_kernel: a direct reduce-scatter +
// all-gather, 1/N of the buffer over each link) or by RCCL (ncclAllReduce(ncclDouble, ncclSum), the collective
// matFact-mpi.c:207-208 asks for); MF_MULTI_REDUCE=peer|rccl chooses, peer is the default.
//
// Set-up is O(nnz), not O(ndev * nnz): ONE counting pass gives every row's entry count and tells whether the entries
// are sorted by A's key; sorted input (the reference's files are (row, col)-sorted) makes every shard a contiguous
// slice of the caller's array, handed to the plan as it is (16-byte structs, split on the device); otherwise one
// stable scatter pass buckets the entries by owner.  Plans are built, and recommendations run, from one host thread
// per shard, so the devices work concurrently.

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the library itself is dlopen()ed on first use
#include <chrono>
#include <thread>

namespace {

// ---- RCCL, loaded on first use: librccl.so is ~570 MB of code objects that a single-GPU run never needs
struct Rccl {
	typedef decltype(&ncclCommInitAll) CommInitAll_t;
	typedef decltype(&ncclCommDestroy) CommDestroy_t;
	typedef decltype(&ncclAllReduce) AllReduce_t;
	typedef decltype(&ncclGroupStart) Group_t;
	typedef decltype(&ncclGetErrorString) ErrStr_t;
	void *handle = nullptr;
	CommInitAll_t CommInitAll = nullptr;
	CommDestroy_t CommDestroy = nullptr;
	AllReduce_t AllReduce = nullptr;
	Group_t GroupStart = nullptr, GroupEnd = nullptr;
	ErrStr_t GetErrorString = nullptr;
	bool load()
	{
		if (handle) return true;
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (handle) break;
		}
		if (!handle) return false;
		CommInitAll = (CommInitAll_t) dlsym(handle, "ncclCommInitAll");
		CommDestroy = (CommDestroy_t) dlsym(handle, "ncclCommDestroy");
		AllReduce = (AllReduce_t) dlsym(handle, "ncclAllReduce");
		GroupStart = (Group_t) dlsym(handle, "ncclGroupStart");
		GroupEnd = (Group_t) dlsym(handle, "ncclGroupEnd");
		GetErrorString = (ErrStr_t) dlsym(handle, "ncclGetErrorString");
		return CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd && GetErrorString;
	}
};
Rccl g_rccl;

struct MultiTiming {
	double setup_s = 0, iterate_s = 0, recommend_s = 0;
	int shards = 0, reducer = 0, sliced = 0;
};
MultiTiming g_multi_timing;

double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Entries of every shard as slices of ONE array of 16-byte structs.  owner blocks are [begin[g], begin[g+1]) over
// the key `by_col ? col : row`.  Sorted by that key: the slices are the caller's own array (no copy); otherwise a
// stable scatter into `store`.  cnt[] (prefix sums of the per-key counts, nkeys + 1) comes from the counting pass.
struct ShardSlices {
	const mf_entry *base = nullptr;
	std::vector<int64_t> off;      // ndev + 1 offsets into base
	std::vector<mf_entry> store;   // only when a scatter was needed
};

int slice_entries(const mf_entry *entries, int64_t nnz, bool by_col, bool sorted, const std::vector<int64_t> &cnt,
                  const std::vector<int> &begin, ShardSlices &out)
{
	const int ndev = (int) begin.size() - 1;
	out.off.assign((size_t) ndev + 1, 0);
	for (int g = 0; g <= ndev; ++g) out.off[(size_t) g] = cnt[(size_t) begin[(size_t) g]];
	if (sorted) {
		out.base = entries;
		return MF_OK;
	}
	try {
		out.store.resize((size_t) nnz);
	} catch (const std::bad_alloc &) {
		return MF_ERR_NO_MEMORY;
	}
	const int nkeys = (int) cnt.size() - 1;
	std::vector<signed char> owner((size_t) std::max(nkeys, 1));
	for (int g = 0; g < ndev; ++g)
		for (int k = begin[(size_t) g]; k < begin[(size_t) g + 1]; ++k) owner[(size_t) k] = (signed char) g;
	std::vector<int64_t> fill(out.off.begin(), out.off.end() - 1);
	for (int64_t n = 0; n < nnz; ++n) {   // stable: file order inside every shard is what keeps the summation order
		const int key = by_col ? entries[n].col : entries[n].row;
		out.store[(size_t) fill[(size_t) owner[(size_t) key]]++] = entries[n];
	}
	out.base = out.store.data();
	return MF_OK;
}

// per-key counts (prefix-summed) + sortedness by that key + range check of both indices: the one pass over the input
int count_entries(const mf_problem *pr, bool by_col, std::vector<int64_t> &cnt, bool &sorted)
{
	const int nkeys = by_col ? pr->items : pr->users;
	try {
		cnt.assign((size_t) nkeys + 1, 0);
	} catch (const std::bad_alloc &) {
		return MF_ERR_NO_MEMORY;
	}
	sorted = true;
	int prev = -1;
	for (int64_t n = 0; n < pr->nnz; ++n) {
		const mf_entry &e = pr->entries[n];
		if (e.row < 0 || e.row >= pr->users || e.col < 0 || e.col >= pr->items) return MF_ERR_ARGUMENT;
		const int key = by_col ? e.col : e.row;
		cnt[(size_t) key + 1]++;
		sorted = sorted && key >= prev;
		prev = key;
	}
	for (int k = 0; k < nkeys; ++k) cnt[(size_t) k + 1] += cnt[(size_t) k];
	return MF_OK;
}

// blocks of keys balanced by entry count, cut at key boundaries (reference: BLOCK_LOW balances rows, mpiutil.h:8)
void balance_blocks(const std::vector<int64_t> &cnt, int ndev, std::vector<int> &begin)
{
	const int nkeys = (int) cnt.size() - 1;
	begin.assign((size_t) ndev + 1, 0);
	int u = 0;
	for (int g = 1; g < ndev; ++g) {
		const int64_t target = cnt[(size_t) nkeys] * g / ndev;
		while (u < nkeys && cnt[(size_t) u] < target) ++u;
		begin[(size_t) g] = u;
	}
	begin[(size_t) ndev] = nkeys;
}

// runs fn(g) for g = 0..n-1 on one host thread each (every HIP call inside sets its own device) and returns the
// first non-OK status; the thread-local HIP error text of a failing worker is carried back to the caller's thread
template <typename F>
int for_each_shard(int n, F fn)
{
	std::vector<int> rc((size_t) n, MF_OK);
	std::vector<std::string> err((size_t) n);
	std::vector<std::thread> th;
	for (int g = 1; g < n; ++g)
		th.emplace_back([&, g] {
			rc[(size_t) g] = fn(g);
			err[(size_t) g] = g_last_hip_error;
		});
	rc[0] = fn(0);
	err[0] = g_last_hip_error;
	for (auto &t : th) t.join();
	for (int g = 0; g < n; ++g)
		if (rc[(size_t) g] != MF_OK) {
			g_last_hip_error = err[(size_t) g];
			return rc[(size_t) g];
		}
	return MF_OK;
}

struct ShardSet {
	std::vector<mf_plan *> plan;
	std::vector<int> device;
	std::vector<hipStream_t> red_stream;
	std::vector<hipEvent_t> ev_items, ev_red;
	std::vector<ncclComm_t> comm;
	~ShardSet()
	{
		for (size_t g = 0; g < plan.size(); ++g) {
			if (g < device.size()) (void) hipSetDevice(device[g]);
			if (g < comm.size() && comm[g] && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(comm[g]);
			if (g < ev_items.size() && ev_items[g]) (void) hipEventDestroy(ev_items[g]);
			if (g < ev_red.size() && ev_red[g]) (void) hipEventDestroy(ev_red[g]);
			if (plan[g]) (void) hipStreamSynchronize(plan[g]->stream);
			if (g < red_stream.size() && red_stream[g]) (void) hipStreamDestroy(red_stream[g]);
			mf_plan_destroy(plan[g]);
		}
	}
};

// One sharded factorisation over plans that already hold their entries and initial factors.  Per iteration and
// shard g:   B sweep (shard 0 seeds from the old factor, the others from zero: matFact-mpi.c:187)  -> ev_items[g]
//            A sweep on the SAME stream (needs no communication)            || reduce of B_next on red_stream[g]
//            flip once every reduce has finished.
// The reduce runs on its own high-priority stream, so it overlaps the A sweep (the MPI variant's
// MPI_Iallreduce ... MPI_Waitall, matFact-mpi.c:207-209).
int iterate_shards(ShardSet &ss, int iters, size_t nb, bool use_rccl)
{
	const int ndev = (int) ss.plan.size();
	for (int it = 0; it < iters; ++it) {
		for (int g = 0; g < ndev; ++g) {
			mf_plan *p = ss.plan[(size_t) g];
			int rc = mf_plan_sweep_items(p, g == 0);
			if (rc != MF_OK) return rc;
			MF_HIP(hipEventRecord(ss.ev_items[(size_t) g], p->stream));
			rc = mf_plan_sweep_users(p);
			if (rc != MF_OK) return rc;
		}
		if (use_rccl) {
			// one group call: a single thread drives every device's rank of the communicator
			for (int g = 0; g < ndev; ++g) {
				MF_HIP(hipSetDevice(ss.plan[(size_t) g]->device));
				MF_HIP(hipStreamWaitEvent(ss.red_stream[(size_t) g], ss.ev_items[(size_t) g], 0));
			}
			ncclResult_t nrc = g_rccl.GroupStart();
			for (int g = 0; g < ndev && nrc == ncclSuccess; ++g) {
				mf_plan *p = ss.plan[(size_t) g];
				double *buf = p->Rbuf[p->cur ^ 1];
				nrc = g_rccl.AllReduce(buf, buf, nb, ncclDouble, ncclSum, ss.comm[(size_t) g], ss.red_stream[(size_t) g]);
			}
			const ncclResult_t erc = g_rccl.GroupEnd();
			if (nrc == ncclSuccess) nrc = erc;
			if (nrc != ncclSuccess) {
				g_last_hip_error = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(nrc);
				return MF_ERR_HIP;
			}
		} else {
			for (int g = 0; g < ndev; ++g) {
				MF_HIP(hipSetDevice(ss.plan[(size_t) g]->device));
				for (int h = 0; h < ndev; ++h)
					MF_HIP(hipStreamWaitEvent(ss.red_stream[(size_t) g], ss.ev_items[(size_t) h], 0));
				mf::PeerReduceArgs a;
				a.nshards = ndev;
				for (int h = 0; h < ndev; ++h) a.buf[h] = ss.plan[(size_t) h]->Rbuf[ss.plan[(size_t) h]->cur ^ 1];
				a.begin = ((nb / 2) * (size_t) g / (size_t) ndev) * 2;
				a.end = g == ndev - 1 ? nb : ((nb / 2) * (size_t) (g + 1) / (size_t) ndev) * 2;
				if (a.end > a.begin) {
					const size_t pairs = (a.end - a.begin + 1) / 2;
					const unsigned grid = (unsigned) std::min<size_t>((pairs + 255) / 256, 2048);
					hipLaunchKernelGGL(mf::peer_allreduce_kernel, dim3(grid), dim3(256), 0, ss.red_stream[(size_t) g], a);
					MF_HIP(hipGetLastError());
				}
			}
		}
		for (int g = 0; g < ndev; ++g) {
			MF_HIP(hipSetDevice(ss.plan[(size_t) g]->device));
			MF_HIP(hipEventRecord(ss.ev_red[(size_t) g], ss.red_stream[(size_t) g]));
		}
		// the peer reduce of device h writes its slice into EVERY shard's buffer: wait for all of them; RCCL's
		// all-reduce on my stream completes only when my buffer is final: my own event is enough
		for (int g = 0; g < ndev; ++g) {
			mf_plan *p = ss.plan[(size_t) g];
			MF_HIP(hipSetDevice(p->device));
			for (int h = 0; h < ndev; ++h)
				if (!use_rccl || h == g) MF_HIP(hipStreamWaitEvent(p->stream, ss.ev_red[(size_t) h], 0));
			mf_plan_flip(p);
		}
		// (the next iteration's reduce on red_stream[g] waits for every shard's next B sweep, which its main stream
		// enqueues behind the joins above: no reduce can run ahead of an unfinished one)
	}
	for (int g = 0; g < ndev; ++g) {
		const int rc = mf_plan_synchronize(ss.plan[(size_t) g]);
		if (rc != MF_OK) return rc;
	}
	return MF_OK;
}

}  // namespace

extern "C" {

int mf_backend_multi_last_timing(double *setup_s, double *iterate_s, double *recommend_s, int *info)
{
	if (setup_s) *setup_s = g_multi_timing.setup_s;
	if (iterate_s) *iterate_s = g_multi_timing.iterate_s;
	if (recommend_s) *recommend_s = g_multi_timing.recommend_s;
	if (info) {
		info[0] = g_multi_timing.shards;
		info[1] = g_multi_timing.reducer;   // 0 peer kernel, 1 RCCL
		info[2] = g_multi_timing.sliced;    // 1: shards were slices of the caller's array (no bucketing pass)
	}
	return MF_OK;
}

int mf_backend_run_multi(const mf_problem *pr, double *L, double *R, int32_t *best, const int *devices, int ndev)
{
	if (!pr || !L || !R || !devices || ndev < 1 || ndev > mf::kMaxShards || pr->users < 0 || pr->items < 0 ||
	    pr->features < 1 || pr->nnz < 0 || pr->iters < 0 || (pr->nnz > 0 && !pr->entries))
		return MF_ERR_ARGUMENT;
	const char *force = getenv("MF_MULTI_FORCE");   // "1": take the sharded path even for one shard (tests)
	if (ndev == 1 && !(force && force[0] == '1')) return mf_backend_run(pr, L, R, best, devices[0]);
	const int total = mf_backend_device_count();
	if (total <= 0) return MF_ERR_NO_DEVICE;
	for (int g = 0; g < ndev; ++g)
		if (devices[g] < 0 || devices[g] >= total) return MF_ERR_NO_DEVICE;
	const char *red = getenv("MF_MULTI_REDUCE");    // "peer" (default) | "rccl"
	const bool use_rccl = red && strcmp(red, "rccl") == 0;
	const int U = pr->users, I = pr->items, K = pr->features;
	const double t_start = now_s();
	g_multi_timing = MultiTiming();
	g_multi_timing.shards = ndev;
	g_multi_timing.reducer = use_rccl ? 1 : 0;
	bool distinct = true;
	for (int g = 0; g < ndev; ++g)
		for (int h = 0; h < g; ++h) distinct = distinct && devices[g] != devices[h];
	if (use_rccl && !distinct) return MF_ERR_UNSUPPORTED;   // RCCL wants one rank per device
	if (use_rccl && !g_rccl.load()) {
		g_last_hip_error = "librccl.so could not be loaded";
		return MF_ERR_UNSUPPORTED;
	}
	// ---- peer access between distinct devices (the peer kernel dereferences every shard's buffer)
	if (!use_rccl)
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
	// items is the same computation with the roles of (row, L) and (col, R) exchanged (the plan reads the structs
	// with row and col swapped); file order is untouched, so every per-row and per-column summation order is too.
	const bool cut_items = I > U;
	const int nrows_a = cut_items ? I : U, nrows_b = cut_items ? U : I;
	double *A = cut_items ? R : L, *B = cut_items ? L : R;
	std::vector<int64_t> cnt;
	std::vector<int> begin;
	bool sorted = true;
	int rc = count_entries(pr, cut_items, cnt, sorted);
	if (rc != MF_OK) return rc;
	balance_blocks(cnt, ndev, begin);
	ShardSlices sl;
	rc = slice_entries(pr->entries, pr->nnz, cut_items, sorted, cnt, begin, sl);
	if (rc != MF_OK) return rc;
	g_multi_timing.sliced = sorted ? 1 : 0;

	ShardSet ss;
	ss.plan.assign((size_t) ndev, nullptr);
	ss.device.assign(devices, devices + ndev);
	ss.red_stream.assign((size_t) ndev, nullptr);
	ss.ev_items.assign((size_t) ndev, nullptr);
	ss.ev_red.assign((size_t) ndev, nullptr);
	ss.comm.assign((size_t) ndev, nullptr);
	rc = for_each_shard(ndev, [&](int g) -> int {
		mf_shard s;
		memset(&s, 0, sizeof s);
		s.users_total = nrows_a;
		s.items = nrows_b;
		s.features = K;
		s.user_begin = begin[(size_t) g];
		s.user_count = begin[(size_t) g + 1] - begin[(size_t) g];
		s.nnz = sl.off[(size_t) g + 1] - sl.off[(size_t) g];
		s.alpha = pr->alpha;
		s.device = devices[g];
		int r = plan_create_impl(&ss.plan[(size_t) g], &s, sl.base + sl.off[(size_t) g], cut_items);
		if (r != MF_OK) return r;
		r = mf_plan_upload_factors(ss.plan[(size_t) g], A + (size_t) begin[(size_t) g] * K, B);
		if (r != MF_OK) return r;
		int prio_lo = 0, prio_hi = 0;
		MF_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
		MF_HIP(hipStreamCreateWithPriority(&ss.red_stream[(size_t) g], hipStreamNonBlocking, prio_hi));
		MF_HIP(hipEventCreateWithFlags(&ss.ev_items[(size_t) g], hipEventDisableTiming));
		MF_HIP(hipEventCreateWithFlags(&ss.ev_red[(size_t) g], hipEventDisableTiming));
		return MF_OK;
	});
	if (rc != MF_OK) return rc;
	if (use_rccl) {
		const ncclResult_t nrc = g_rccl.CommInitAll(ss.comm.data(), ndev, devices);
		if (nrc != ncclSuccess) {
			g_last_hip_error = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nrc);
			return MF_ERR_HIP;
		}
	}
	const double t_setup = now_s();
	g_multi_timing.setup_s = t_setup - t_start;

	// the replicated buffers are summed whole, padding included (every plan pads the same way: zeros)
	rc = iterate_shards(ss, pr->iters, (size_t) nrows_b * (size_t) ss.plan[0]->ldr, use_rccl);
	if (rc != MF_OK) return rc;
	const double t_iter = now_s();
	g_multi_timing.iterate_s = t_iter - t_setup;

	// ---- factors back: every shard its block of A, shard 0 the replicated B
	rc = for_each_shard(ndev, [&](int g) -> int {
		return mf_plan_download_factors(ss.plan[(size_t) g], A + (size_t) begin[(size_t) g] * K, g == 0 ? B : nullptr);
	});
	if (rc != MF_OK || !best) return rc;

	// ---- recommendations: by user blocks against the full R, no collective (the fused form of matFact-mpi.c:82-96)
	if (!cut_items) {
		// the factorisation plans ARE user blocks holding the reduced R and the block's CSR: recommend on them, all
		// devices at once
		rc = for_each_shard(ndev, [&](int g) -> int {
			return mf_plan_recommend(ss.plan[(size_t) g], best + begin[(size_t) g]);
		});
	} else {
		// items were cut: the plans hold item blocks.  Free them, then one user-block plan per device (slices of the
		// caller's array again when it is row-sorted)
		for (auto &p : ss.plan) {
			mf_plan_destroy(p);
			p = nullptr;
		}
		std::vector<int64_t> ucnt;
		std::vector<int> ubegin;
		bool usorted = true;
		rc = count_entries(pr, false, ucnt, usorted);
		if (rc != MF_OK) return rc;
		balance_blocks(ucnt, ndev, ubegin);
		ShardSlices usl;
		rc = slice_entries(pr->entries, pr->nnz, false, usorted, ucnt, ubegin, usl);
		if (rc != MF_OK) return rc;
		rc = for_each_shard(ndev, [&](int g) -> int {
			const int b0 = ubegin[(size_t) g], b1 = ubegin[(size_t) g + 1];
			if (b1 == b0) return MF_OK;
			mf_shard s;
			memset(&s, 0, sizeof s);
			s.users_total = U;
			s.items = I;
			s.features = K;
			s.user_begin = b0;
			s.user_count = b1 - b0;
			s.nnz = usl.off[(size_t) g + 1] - usl.off[(size_t) g];
			s.alpha = pr->alpha;
			s.device = devices[g];
			mf_plan *plan = nullptr;
			int r = plan_create_impl(&plan, &s, usl.base + usl.off[(size_t) g], false);
			if (r == MF_OK) r = mf_plan_upload_factors(plan, L + (size_t) b0 * K, R);
			if (r == MF_OK) r = mf_plan_recommend(plan, best + b0);
			mf_plan_destroy(plan);
			return r;
		});
	}
	g_multi_timing.recommend_s = now_s() - t_iter;
	return rc;
}

}  // extern "C"
