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
#include <atomic>
#include <chrono>
#include <condition_variable>
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
	double setup_s = 0, iterate_s = 0, recommend_s = 0, enqueue_s = 0;
	int shards = 0, reducer = 0, sliced = 0, threads = 0;
	long long entry_passes = 0;   // passes of the host over all nnz entries during set-up (count, scatter)
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
	g_multi_timing.entry_passes++;   // the stable scatter reads every entry once more
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
	g_multi_timing.entry_passes++;
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

// A reusable barrier between the per-shard host threads (C++17: no std::barrier).
struct HostBarrier {
	std::mutex mu;
	std::condition_variable cv;
	int n, waiting = 0;
	unsigned long long generation = 0;
	explicit HostBarrier(int count) : n(count) {}
	void arrive_and_wait()
	{
		std::unique_lock<std::mutex> lk(mu);
		const unsigned long long gen = generation;
		if (++waiting == n) {
			waiting = 0;
			++generation;
			cv.notify_all();
		} else
			cv.wait(lk, [&] { return generation != gen; });
	}
};

// What one shard enqueues per iteration.  Phase 1: B sweep (shard 0 seeds from the old factor, the others from zero:
// matFact-mpi.c:187), ev_items[g], A sweep on the same stream (needs no communication).  Phase 2, once EVERY shard has
// recorded its ev_items: the reduce of B_next on red_stream[g] (its own high-priority stream, so it overlaps the A
// sweep -- the MPI_Iallreduce ... MPI_Waitall of matFact-mpi.c:207-209), ev_red[g].  Phase 3, once every ev_red is
// recorded: the plan's stream waits for the reduces that wrote into its buffer, then flips.
//
// hipStreamWaitEvent captures the event's LATEST record at the time of the call, so a wait on another shard's event must
// be enqueued after that shard's thread has recorded it for this iteration: the phases are separated by host barriers
// when the shards are driven by one thread each (threads = true), and by program order when one thread drives them all.
struct ShardLoop {
	ShardSet &ss;
	size_t nb;
	bool use_rccl;
	int ndev;
	int phase1(int g)
	{
		mf_plan *p = ss.plan[(size_t) g];
		int rc = mf_plan_sweep_items(p, g == 0);   // sets the device
		if (rc != MF_OK) return rc;
		MF_HIP(hipEventRecord(ss.ev_items[(size_t) g], p->stream));
		return mf_plan_sweep_users(p);
	}
	int phase2_peer(int g)
	{
		MF_HIP(hipSetDevice(ss.plan[(size_t) g]->device));
		// device g sums slice g of every shard's buffer: it needs every shard's B sweep
		for (int h = 0; h < ndev; ++h) MF_HIP(hipStreamWaitEvent(ss.red_stream[(size_t) g], ss.ev_items[(size_t) h], 0));
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
		MF_HIP(hipEventRecord(ss.ev_red[(size_t) g], ss.red_stream[(size_t) g]));
		return MF_OK;
	}
	int phase2_rccl_wait(int g)
	{
		MF_HIP(hipSetDevice(ss.plan[(size_t) g]->device));
		MF_HIP(hipStreamWaitEvent(ss.red_stream[(size_t) g], ss.ev_items[(size_t) g], 0));   // RCCL orders the ranks itself
		return MF_OK;
	}
	int phase2_rccl_reduce(int g)
	{
		mf_plan *p = ss.plan[(size_t) g];
		double *buf = p->Rbuf[p->cur ^ 1];
		const ncclResult_t nrc = g_rccl.AllReduce(buf, buf, nb, ncclDouble, ncclSum, ss.comm[(size_t) g], ss.red_stream[(size_t) g]);
		if (nrc != ncclSuccess) {
			g_last_hip_error = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(nrc);
			return MF_ERR_HIP;
		}
		return MF_OK;
	}
	int phase2_rccl_record(int g)
	{
		MF_HIP(hipSetDevice(ss.plan[(size_t) g]->device));
		MF_HIP(hipEventRecord(ss.ev_red[(size_t) g], ss.red_stream[(size_t) g]));
		return MF_OK;
	}
	int phase3(int g)
	{
		mf_plan *p = ss.plan[(size_t) g];
		MF_HIP(hipSetDevice(p->device));
		// the peer reduce of device h writes its slice into EVERY shard's buffer: wait for all of them; RCCL's
		// all-reduce on my stream completes only when my buffer is final: my own event is enough
		for (int h = 0; h < ndev; ++h)
			if (!use_rccl || h == g) MF_HIP(hipStreamWaitEvent(p->stream, ss.ev_red[(size_t) h], 0));
		return mf_plan_flip(p);
		// (the next iteration's reduce on red_stream[g] waits for every shard's next B sweep, which its main stream
		// enqueues behind the joins above: no reduce can run ahead of an unfinished one)
	}
};

// One sharded factorisation over plans that already hold their entries and initial factors.  threads = true (default):
// one host thread per shard enqueues that shard's work, so the ~2N+6 runtime calls a shard needs per iteration (N event
// waits before its reduce, N before its flip) are issued concurrently instead of N(2N+6) in a row from one thread -- at
// N = 8 about 180 calls per iteration against a 3 ms iteration of the cfg4 shard.  enqueue_s returns the time the
// slowest thread spent enqueueing (everything but the final synchronize).
int iterate_shards(ShardSet &ss, int iters, size_t nb, bool use_rccl, bool threads, double *enqueue_s)
{
	const int ndev = (int) ss.plan.size();
	ShardLoop loop{ss, nb, use_rccl, ndev};
	if (enqueue_s) *enqueue_s = 0.0;
	if (!threads || ndev == 1) {
		const double t0 = now_s();
		for (int it = 0; it < iters; ++it) {
			for (int g = 0; g < ndev; ++g) {
				const int rc = loop.phase1(g);
				if (rc != MF_OK) return rc;
			}
			if (use_rccl) {
				// one group call: a single thread drives every device's rank of the communicator
				for (int g = 0; g < ndev; ++g) {
					const int rc = loop.phase2_rccl_wait(g);
					if (rc != MF_OK) return rc;
				}
				ncclResult_t nrc = g_rccl.GroupStart();
				int rc = MF_OK;
				for (int g = 0; g < ndev && nrc == ncclSuccess && rc == MF_OK; ++g) rc = loop.phase2_rccl_reduce(g);
				const ncclResult_t erc = g_rccl.GroupEnd();
				if (rc != MF_OK) return rc;
				if (nrc == ncclSuccess) nrc = erc;
				if (nrc != ncclSuccess) {
					g_last_hip_error = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(nrc);
					return MF_ERR_HIP;
				}
				for (int g = 0; g < ndev; ++g) {
					rc = loop.phase2_rccl_record(g);
					if (rc != MF_OK) return rc;
				}
			} else
				for (int g = 0; g < ndev; ++g) {
					const int rc = loop.phase2_peer(g);
					if (rc != MF_OK) return rc;
				}
			for (int g = 0; g < ndev; ++g) {
				const int rc = loop.phase3(g);
				if (rc != MF_OK) return rc;
			}
		}
		if (enqueue_s) *enqueue_s = now_s() - t0;
	} else {
		HostBarrier bar(ndev);
		std::vector<int> failed((size_t) ndev, MF_OK);
		std::vector<double> spent((size_t) ndev, 0.0);
		std::atomic<int> any_failed{0};
		// a thread that fails keeps arriving at the barriers (skipping its work) so nobody waits for it forever
		const int rc = for_each_shard(ndev, [&](int g) -> int {
			const double t0 = now_s();
			int mine = MF_OK;
			auto run = [&](int r) {
				if (r != MF_OK && mine == MF_OK) {
					mine = r;
					any_failed.store(1);
				}
			};
			for (int it = 0; it < iters; ++it) {
				if (!any_failed.load()) run(loop.phase1(g));
				bar.arrive_and_wait();   // every ev_items of this iteration is recorded
				if (!any_failed.load()) {
					if (use_rccl) {      // one thread per rank, no group: each rank's call may block until its peers arrive
						run(loop.phase2_rccl_wait(g));
						if (mine == MF_OK) run(loop.phase2_rccl_reduce(g));
						if (mine == MF_OK) run(loop.phase2_rccl_record(g));
					} else
						run(loop.phase2_peer(g));
				}
				bar.arrive_and_wait();   // every ev_red of this iteration is recorded
				if (!any_failed.load()) run(loop.phase3(g));
				// (no barrier here: the next wait on one of my events comes after the next iteration's first barrier, which
				// I reach only after re-recording it)
			}
			spent[(size_t) g] = now_s() - t0;
			failed[(size_t) g] = mine;
			return mine;
		});
		if (rc != MF_OK) return rc;
		if (enqueue_s) *enqueue_s = *std::max_element(spent.begin(), spent.end());
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

int mf_backend_multi_last_counters(double *enqueue_s, int64_t *entry_passes, int *host_threads)
{
	if (enqueue_s) *enqueue_s = g_multi_timing.enqueue_s;
	if (entry_passes) *entry_passes = g_multi_timing.entry_passes;
	if (host_threads) *host_threads = g_multi_timing.threads;
	return MF_OK;
}

int mf_backend_run_multi(const mf_problem *pr, double *L, double *R, int32_t *best, const int *devices, int ndev)
{
	if (!pr || !L || !R || !devices || ndev < 1 || ndev > mf::kMaxShards || pr->users < 0 || pr->items < 0 ||
	    pr->features < 1 || pr->nnz < 0 || pr->iters < 0 || (pr->nnz > 0 && !pr->entries))
		return MF_ERR_ARGUMENT;
	const mf_config cfg = mf_config::from_env();
	if (ndev == 1 && !cfg.multi_force) return   /* MF_MULTI_FORCE=1: the sharded path even for one shard (tests) */ mf_backend_run(pr, L, R, best, devices[0]);
	const int total = mf_backend_device_count();
	if (total <= 0) return MF_ERR_NO_DEVICE;
	for (int g = 0; g < ndev; ++g)
		if (devices[g] < 0 || devices[g] >= total) return MF_ERR_NO_DEVICE;
	const bool use_rccl = cfg.multi_rccl;   // MF_MULTI_REDUCE=peer (default) | rccl
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
	g_multi_timing.threads = cfg.multi_threads && ndev > 1 ? ndev : 1;
	rc = iterate_shards(ss, pr->iters, (size_t) nrows_b * (size_t) ss.plan[0]->ldr, use_rccl, cfg.multi_threads, &g_multi_timing.enqueue_s);
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
