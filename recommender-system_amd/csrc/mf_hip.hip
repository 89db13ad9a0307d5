// mf_hip.hip -- C ABI (include/matfact_hip.h) of the MI355X backend: plan management, CSR/CSC build,
// kernel dispatch.  HIP only -- there is no CPU compute path in this library.
#include <cstring>
#include <map>
#include <mutex>

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

#include "mf_config.hip.h"
#include "mf_plan.hip.h"
#include "mf_launch.hip.h"
#include "mf_build.hip.h"

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

}   // extern "C"

// `aos` (optional): the entries as the reference's array of (row, col, value) structs; they are then uploaded as they
// are and split into the three arrays on the device (the level-1 entry points: no host-side copy of 1e8 entries).
// `swap`: read the structs with row and col exchanged (the item-cut form of mf_backend_run_multi).
static int plan_create_impl(mf_plan **out, const mf_shard *s, const mf_entry *aos, bool swap = false)
{
	if (!out) return MF_ERR_ARGUMENT;
	*out = nullptr;
	if (!s || s->users_total < 0 || s->items < 0 || s->features < 1 || s->nnz < 0 || s->user_begin < 0 ||
	    s->user_count < 0 || (int64_t) s->user_begin + s->user_count > s->users_total ||
	    s->nnz > INT32_MAX - 64 || (s->nnz > 0 && !aos && (!s->row || !s->col || !s->val)))
		return MF_ERR_ARGUMENT;
	// caller-owned R buffers: both or neither, and 16-B aligned (the gather moves 16-byte pieces of rows)
	if ((s->items_ext[0] == nullptr) != (s->items_ext[1] == nullptr) || ((uintptr_t) s->items_ext[0] & 15) ||
	    ((uintptr_t) s->items_ext[1] & 15) || (s->items_ext[0] && s->items_ext[0] == s->items_ext[1]))
		return MF_ERR_ARGUMENT;
	if ((s->users_ext[0] == nullptr) != (s->users_ext[1] == nullptr) || ((uintptr_t) s->users_ext[0] & 15) ||
	    ((uintptr_t) s->users_ext[1] & 15) || (s->users_ext[0] && s->users_ext[0] == s->users_ext[1]))
		return MF_ERR_ARGUMENT;
	// declared pitch of caller-owned buffers: even (16-byte aligned rows) and at least K
	if (s->items_pitch < 0 || s->users_pitch < 0 || (s->items_pitch && (s->items_pitch < s->features || (s->items_pitch & 1))) ||
	    (s->users_pitch && (s->users_pitch < s->features || (s->users_pitch & 1))))
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
	p->cfg = mf_config::from_env();

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
		std::vector<int> rptr, cptr;
		MF_TRY(build_sparse(p, s, aos, swap, rptr, cptr));
		MF_TRY(plan_row_schedule(p, rptr, cptr));
		MF_TRY(plan_es_schedule(p, rptr, cptr));
	}

	// Row pitch of the factor buffers the plan owns: rows of 8K bytes are gathered in whole 128-byte lines, so when 8K
	// is not a multiple of 128 a row costs a line more than its bytes wherever it happens to start (80-byte rows:
	// 1.5 lines on average instead of 1; 240-byte rows: 2.75 instead of 2).  The plan pads its own rows to whole
	// lines where that saves at least a tenth of the lines; caller-owned buffers keep the caller's pitch K.
	p->ldl = p->ldr = p->K;
	{
		const int own = row_pitch(p->cfg, p->K, p->sweep.dma != 0);
		p->ldl = s->users_ext[0] && s->users_ext[1] ? (s->users_pitch ? s->users_pitch : p->K) : own;
		p->ldr = s->items_ext[0] && s->items_ext[1] ? (s->items_pitch ? s->items_pitch : p->K) : own;
	}
	const size_t nl = (size_t) p->uc * p->ldl, nr = (size_t) p->items * p->ldr;
	if (s->users_ext[0] && s->users_ext[1]) {
		p->l_external = true;
		p->Lbuf[0] = (double *) s->users_ext[0];
		p->Lbuf[1] = (double *) s->users_ext[1];
	} else {
		MF_TRY(dev_alloc(&p->Lbuf[0], nl));
		MF_TRY(dev_alloc(&p->Lbuf[1], nl));
		if (p->ldl != p->K) {   // the padding is never read by a kernel, but it is summed by the multi-GPU reducers
			MF_TRY_HIP(hipMemsetAsync(p->Lbuf[0], 0, std::max<size_t>(nl, 1) * sizeof(double), p->stream));
			MF_TRY_HIP(hipMemsetAsync(p->Lbuf[1], 0, std::max<size_t>(nl, 1) * sizeof(double), p->stream));
		}
	}
	if (s->items_ext[0] && s->items_ext[1]) {
		p->r_external = true;
		p->Rbuf[0] = (double *) s->items_ext[0];
		p->Rbuf[1] = (double *) s->items_ext[1];
	} else {
		MF_TRY(dev_alloc(&p->Rbuf[0], nr));
		MF_TRY(dev_alloc(&p->Rbuf[1], nr));
		if (p->ldr != p->K) {
			MF_TRY_HIP(hipMemsetAsync(p->Rbuf[0], 0, std::max<size_t>(nr, 1) * sizeof(double), p->stream));
			MF_TRY_HIP(hipMemsetAsync(p->Rbuf[1], 0, std::max<size_t>(nr, 1) * sizeof(double), p->stream));
		}
	}
	MF_TRY_HIP(hipStreamSynchronize(p->stream));   // the plan is complete when the call returns
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

extern "C" {

int mf_plan_create(mf_plan **out, const mf_shard *s) { return plan_create_impl(out, s, nullptr); }

int mf_backend_row_pitch(int features)
{
	if (features < 1) return MF_ERR_ARGUMENT;
	const mf_config cfg = mf_config::from_env();
	return row_pitch(cfg, features, sweep_is_dma(cfg, features));   // rows are padded for the LDS-DMA forms only
}

int mf_plan_row_pitch(mf_plan *p, int32_t *users_pitch, int32_t *items_pitch)
{
	if (!p) return MF_ERR_ARGUMENT;
	if (users_pitch) *users_pitch = p->ldl;
	if (items_pitch) *items_pitch = p->ldr;
	return MF_OK;
}

void mf_plan_destroy(mf_plan *p)
{
	if (!p) return;
	(void) hipSetDevice(p->device);
	if (p->stream) (void) hipStreamSynchronize(p->stream);
	for (auto &t : p->timed) {
		if (!t.shared_start) (void) hipEventDestroy(t.t0);
		(void) hipEventDestroy(t.t1);
	}
	(void) hipFree(p->csr2csc);
	(void) hipFree(p->mask_idx);
	(void) hipFree(p->rec_csr);
	(void) hipFree(p->rec_csc);
	(void) hipFree(p->es_seg_row);
	(void) hipFree(p->es_seg_beg);
	(void) hipFree(p->es_seg_end);
	(void) hipFree(p->res_wg);
	(void) hipFree(p->csr_ptr);
	(void) hipFree(p->csr_idx);
	(void) hipFree(p->csr_val);
	(void) hipFree(p->csc_ptr);
	(void) hipFree(p->csc_idx);
	(void) hipFree(p->csc_val);
	if (!p->l_external) {
		(void) hipFree(p->Lbuf[0]);
		(void) hipFree(p->Lbuf[1]);
	}
	(void) hipFree(p->cand_dev);
	(void) hipFree(p->cand_pack);
	(void) hipFree(p->filt_dev);
	(void) hipFree(p->part_dev);
	if (!p->r_external) {
		(void) hipFree(p->Rbuf[0]);
		(void) hipFree(p->Rbuf[1]);
	}
	(void) hipFree(p->best_dev);
	for (int k = 0; k < 2; ++k) {
		(void) hipFree(p->long_rows[k]);
		(void) hipFree(p->mid_rows[k]);
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
	if (p->mid_stream) (void) hipStreamDestroy(p->mid_stream);
	if (p->ev_mid_join) (void) hipEventDestroy(p->ev_mid_join);
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
	// the caller's rows are K doubles apart, the device rows ldl / ldr
	const size_t w = (size_t) p->K * sizeof(double);
	if (p->uc)
		MF_HIP(hipMemcpy2DAsync(p->Lbuf[p->cur], (size_t) p->ldl * sizeof(double), L_block, w, w, (size_t) p->uc,
		                        hipMemcpyHostToDevice, p->stream));
	if (p->items)
		MF_HIP(hipMemcpy2DAsync(p->Rbuf[p->cur], (size_t) p->ldr * sizeof(double), R, w, w, (size_t) p->items,
		                        hipMemcpyHostToDevice, p->stream));
	MF_HIP(hipStreamSynchronize(p->stream));
	p->have_factors = true;
	return MF_OK;
}

int mf_plan_download_factors(mf_plan *p, double *L_block, double *R)
{
	if (!p) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	const size_t w = (size_t) p->K * sizeof(double);
	if (L_block && p->uc)
		MF_HIP(hipMemcpy2DAsync(L_block, w, p->Lbuf[p->cur], (size_t) p->ldl * sizeof(double), w, (size_t) p->uc,
		                        hipMemcpyDeviceToHost, p->stream));
	if (R && p->items)
		MF_HIP(hipMemcpy2DAsync(R, w, p->Rbuf[p->cur], (size_t) p->ldr * sizeof(double), w, (size_t) p->items,
		                        hipMemcpyDeviceToHost, p->stream));
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

int mf_plan_sweep_users_seeded(mf_plan *p, int seed_from_old)
{
	if (!p) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	return launch_sweep(p, 1, seed_from_old ? 1 : 0);
}

int mf_plan_sweep_users(mf_plan *p) { return mf_plan_sweep_users_seeded(p, 1); }

void *mf_plan_items_next(mf_plan *p) { return p ? p->Rbuf[p->cur ^ 1] : nullptr; }
void *mf_plan_items_current(mf_plan *p) { return p ? p->Rbuf[p->cur] : nullptr; }
void *mf_plan_users_next(mf_plan *p) { return p ? p->Lbuf[p->cur ^ 1] : nullptr; }
void *mf_plan_users_current(mf_plan *p) { return p ? p->Lbuf[p->cur] : nullptr; }

int mf_plan_flip(mf_plan *p)
{
	if (!p) return MF_ERR_ARGUMENT;
	p->cur ^= 1;
	return MF_OK;
}

static int iterate_eager(mf_plan *p, int iters)
{
	if (p->es_mode) {
		for (int it = 0; it < iters; ++it) {
			const int rc = launch_es_iteration(p);
			if (rc != MF_OK) return rc;
			p->cur ^= 1;
		}
		return MF_OK;
	}
	for (int it = 0; it < iters; ++it) {
		// Both sweeps read only the frozen generation (matFact.c:38-39), so the ordered sums of the item sweep's
		// extreme rows may run on the side stream UNDER the whole user sweep; they are joined before the flip.
		// Not when the user sweep has extreme rows of its own: it would reuse the scratch buffer.
		int rc = launch_sweep(p, 0, 1, /*defer_join=*/p->n_long[1] == 0 && !p->cfg.no_defer);
		if (rc != MF_OK) return rc;
		rc = launch_sweep(p, 1, 1);
		if (rc != MF_OK) return rc;
		if (p->join_pending) {
			MF_HIP(hipStreamWaitEvent(p->stream, p->ev_join, 0));
			p->join_pending = false;
		}
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
	// Toy regime (inst0/1/2: a dozen entries, 1e5 iterations): the whole instance fits the LDS of one workgroup -> one
	// launch runs all the iterations with a workgroup barrier in between (sweep_resident_kernel).  Whole-instance
	// plans only, and only while an iteration is a few hundred multiply-adds: measured through the CLI, inst1
	// 0.70 -> 0.33 s and inst2 0.47 -> 0.21 s, but inst30-40 (170 entries x K=10) 0.32 -> 0.39 s -- one workgroup
	// on an otherwise idle chip runs slowly, and two graph-replayed launches per iteration win again.
	{
		const size_t need = mf::resident_lds_bytes(p->uc, p->items, p->K, p->nnz);
		const bool toy = p->uc == p->users_total && p->u0 == 0 && p->uc + p->items <= 1024 && p->uc + p->items > 0 &&
		                 need <= 60 * 1024 && (double) p->nnz * p->K <= 512.0 && !p->timing && iters >= 8 &&
		                 p->cfg.resident;
		if (toy) {
			mf::ResidentArgs ra;
			ra.users = p->uc;
			ra.items = p->items;
			ra.K = p->K;
			ra.ldl = p->ldl;
			ra.ldr = p->ldr;
			ra.iters = iters;
			ra.c2 = p->alpha * 2;
			ra.csr_ptr = p->csr_ptr;
			ra.csr_idx = p->csr_idx;
			ra.csr_val = p->csr_val;
			ra.csc_ptr = p->csc_ptr;
			ra.csc_idx = p->csc_idx;
			ra.csc_val = p->csc_val;
			ra.L_in = p->Lbuf[p->cur];
			ra.R_in = p->Rbuf[p->cur];
			// the result lands where `iters` flips of the two generations would have left it
			const int fin = (iters & 1) ? (p->cur ^ 1) : p->cur;
			ra.L_out = p->Lbuf[fin];
			ra.R_out = p->Rbuf[fin];
			ra.nnz = (int) p->nnz;
			const int threads = ((p->uc + p->items + 63) / 64) * 64;
			void (*rfn)(mf::ResidentArgs) = p->K <= 4    ? mf::sweep_resident_kernel<4>
			                                : p->K <= 16 ? mf::sweep_resident_kernel<16>
			                                : p->K <= 32 && threads <= mf::resident_max_threads(32)
			                                    ? mf::sweep_resident_kernel<32>
			                                    : mf::sweep_resident_kernel<0>;
			MF_HIP(raise_lds_limit((const void *) rfn, need));
			hipLaunchKernelGGL(rfn, dim3(1), dim3(threads), need, p->stream, ra);
			MF_HIP(hipGetLastError());
			p->cur = fin;
			return MF_OK;
		}
	}
	const bool small = (double) p->nnz * p->K < p->cfg.graph_max && !p->timing && p->n_long[0] == 0 &&
	                   p->n_long[1] == 0;
	constexpr int kGraphIters = 32;
	if (small && iters >= 4 * kGraphIters && p->cfg.graph) {
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

}   // extern "C"

// Pass 1 of the recommendation: row norms + recommend_mfma_kernel.  filt == nullptr: the kernel certifies per user
// against THIS plan's items (best / list of uncertain users); filt != nullptr: it only reports (best, second, arg,
// non-finite flag) per user, for a certification over several item blocks by the caller (2-D tiles).
static int launch_recommend_pass1(mf_plan *p, mf_filter *filt)
{
	const double *Lc = p->Lbuf[p->cur], *Rc = p->Rbuf[p->cur];
	// pass 1: scores on the FP64 matrix cores + certification margin; pass 2: exact re-scoring of the rest
	MF_HIP(hipMemsetAsync(p->rmax_bits, 0, sizeof(unsigned long long), p->stream));
	MF_HIP(hipMemsetAsync(p->ucount, 0, sizeof(int), p->stream));
	hipLaunchKernelGGL(mf::row_norm_kernel, dim3((p->uc + 63) / 64), dim3(64), 0, p->stream, Lc, p->uc,
	                   p->K, p->ldl, p->lnorm, (unsigned long long *) nullptr);
	if (p->items > 0)
		hipLaunchKernelGGL(mf::row_norm_kernel, dim3((p->items + 63) / 64), dim3(64), 0, p->stream, Rc,
		                   p->items, p->K, p->ldr, (double *) nullptr, p->rmax_bits);
	mf::RecMfmaArgs m;
	m.users = p->uc;
	m.items = p->items;
	m.K = p->K;
	m.ldl = p->ldl;
	m.ldr = p->ldr;
	m.L = Lc;
	m.R = Rc;
	m.csr_ptr = p->csr_ptr;
	m.csr_idx = p->mask_idx ? p->mask_idx : p->csr_idx;
	m.lnorm = p->lnorm;
	m.rnorm_max_bits = p->rmax_bits;
	m.thr_scale = mf_backend_recommend_margin(p->K);
	m.best = p->best_dev;
	m.ulist = p->ulist;
	m.ucount = p->ucount;
	m.filt = filt;
	// Form.  The L block's image stays resident in LDS (it is the same for every item tile) whenever it fits
	// beside the two R buffers, and the R chunks then go global -> LDS by LDS-DMA (even K): K <= 64 with 32-deep
	// chunks, up to K = 100 with 24- or 20-deep ones -- the depth with the fewest chunks wins, an exact divisor
	// of K on ties (K=100: 5 x 20 instead of 32+32+32+4).  Larger K: both operands staged through registers.
	// Measured on 1e6 x 1e5 (profiles/r01/recommend_resident_L_ab.txt): K=100 55.4 vs 48.9 TFLOP/s, K=64 54.3
	// vs 50.2, K=30 41.1 vs 37.7.
	typedef void (*RecFn)(mf::RecMfmaArgs);
	const bool vec = (p->K & 1) == 0;
	const bool allow = p->cfg.rec_ares;              // MF_RECOMMEND_ARES=0 disables the resident-L form (tests, A/B)
	const bool allow_dma = vec && p->cfg.rec_bdma;   // MF_RECOMMEND_BDMA=0: stage R chunks through registers (A/B)
	const size_t static_lds = 8 * 1024, cu_lds = 160 * 1024;   // masks + merge arrays, rounded up
	int kc = 32;
	bool ares = false;
	if (allow) {
		int best_nch = 1 << 30;
		for (int cand : {32, 24, 20}) {
			if (cand == 24 && !allow_dma) continue;                    // 24 exists in the DMA form only
			if (cand == 20 && p->K % 20 != 0 && !allow_dma) continue;   // register form: exact multiples only
			if (mf::rec_mfma_lds(p->K, cand, true) + static_lds > cu_lds) continue;
			const int nch = (p->K + cand - 1) / cand;
			if (nch < best_nch || (nch == best_nch && p->K % cand == 0 && p->K % kc != 0)) {
				best_nch = nch;
				kc = cand;
				ares = true;
			}
		}
	}
	const bool bdma = ares && allow_dma;
	RecFn fn;
	if (kc == 24)
		fn = mf::recommend_mfma_kernel<true, 24, true, true>;
	else if (kc == 20)   // even K here
		fn = bdma ? mf::recommend_mfma_kernel<true, 20, true, true> : mf::recommend_mfma_kernel<true, 20, true, false>;
	else if (bdma)
		fn = mf::recommend_mfma_kernel<true, 32, true, true>;
	else
		fn = ares ? (vec ? mf::recommend_mfma_kernel<true, 32, true> : mf::recommend_mfma_kernel<false, 32, true>)
		          : (vec ? mf::recommend_mfma_kernel<true, 32, false> : mf::recommend_mfma_kernel<false, 32, false>);
	size_t lds = mf::rec_mfma_lds(p->K, kc, ares);
	// K = 20, 40, .. 100 (a wave's L operand fits its registers; whole 20-deep chunks): workgroups of 64 users, two per CU,
	// whose barriers / arg-max steps / mask walks overlap each other's matrix instructions, with a gapless matrix stream
	// per wave.  K=100: 64.7 vs 57.6 TFLOP/s on the 131072 x 100000 probe, K=80 64.3 vs 57.4, K=40 57.7 vs 52.1, K=20 49.7 vs
	// 45.0.  Its general form (any even K <= 100, MF_RECOMMEND_HALF=all) has branches on K in the tile body that defeat
	// hipcc's s_waitcnt placement and is slower than the 128-user kernel (K=64: 51.7 vs 58.5): not chosen by the rule.
	// The same kernel with 16-deep chunks for K = 16, 32, .. 128 (two per CU as well) and, for K = 256, with 16 users per
	// wave and eight waves per workgroup (one per CU): DESIGN.md 5.8.
	const bool fits32 = (unsigned long long) p->items * (unsigned long long) p->ldr * 8ull < (1ull << 32);   // 32-bit row offsets
	RecFn hfn = nullptr;
	int hqc = 0, hwaves = 4;
	if (vec && allow_dma && p->cfg.rec_half && fits32) {
		const int K = p->K;
		if (K % 20 == 0 && K <= mf::kHKmax) {
			static const RecFn f20[5] = {mf::recommend_mfma2_kernel<1>, mf::recommend_mfma2_kernel<2>, mf::recommend_mfma2_kernel<3>,
			                             mf::recommend_mfma2_kernel<4>, mf::recommend_mfma2_kernel<5>};
			hfn = f20[K / 20 - 1];
			hqc = 5;
		} else if (K % 16 == 0 && K <= 128) {
			static const RecFn f16[8] = {mf::recommend_mfma2_kernel<1, 4>, mf::recommend_mfma2_kernel<2, 4>, mf::recommend_mfma2_kernel<3, 4>,
			                             mf::recommend_mfma2_kernel<4, 4>, mf::recommend_mfma2_kernel<5, 4>, mf::recommend_mfma2_kernel<6, 4>,
			                             mf::recommend_mfma2_kernel<7, 4>, mf::recommend_mfma2_kernel<8, 4>};
			hfn = f16[K / 16 - 1];
			hqc = 4;
		} else if (K == 256) {
			hfn = mf::recommend_mfma2_kernel<8, 8, 1, 8>;
			hqc = 8;
			hwaves = 8;
		}
		if (K == 128 && p->cfg.rec_wide) {   // MF_RECOMMEND_WIDE (experiments build): K=128 in the eight-wave shape of K=256
			hfn = mf::recommend_mfma2_kernel<4, 8, 1, 8>;
			hqc = 8;
			hwaves = 8;
		}
		if (!hfn && p->cfg.rec_half == 2 && K <= mf::kHKmax) {
			hfn = mf::recommend_mfma2_kernel<0>;
			hqc = 5;
		}
	}
	const bool half = hfn != nullptr;
	int block_users = mf::kMU, threads = mf::kMThreads;
	if (half) {
		fn = hfn;
		lds = mf::rec_mfma2_lds(hqc);
		block_users = mf::kHU;
		threads = 64 * hwaves;
	}
	p->rec_half_used = half;
	MF_HIP(raise_lds_limit((const void *) fn, lds));
	// Small problems: a workgroup owns 128 (64) users and ALL items, so few users leave most of the chip idle (cfg3: 48
	// workgroups on 256 CUs).  The items are then split over gridDim.y -- whole 128-item tiles, about two workgroups per
	// CU in all (four of the half-size ones) -- and the per-split top-2 reports merged and certified by merge_splits_kernel.
	const int ublocks = (p->uc + block_users - 1) / block_users, tiles = (p->items + mf::kMI - 1) / mf::kMI;
	const int chip = half && hwaves == 4 ? 1024 : 512;   // workgroups the chip holds at once, times two
	int nsplit = 1;
	if (p->cfg.rec_split != 0 && ublocks < chip * 3 / 8 && tiles >= 2) {
		nsplit = p->cfg.rec_split > 0 ? p->cfg.rec_split : (chip + ublocks - 1) / ublocks;
		nsplit = std::max(1, std::min(nsplit, tiles));
	}
	m.split_items = 0;
	m.part = nullptr;
	if (nsplit > 1) {
		const int tiles_per = (tiles + nsplit - 1) / nsplit;
		nsplit = (tiles + tiles_per - 1) / tiles_per;
		m.split_items = tiles_per * mf::kMI;
		if (p->part_cap < nsplit) {
			(void) hipFree(p->part_dev);
			p->part_dev = nullptr;
			p->part_cap = 0;
			const int rc = dev_alloc(&p->part_dev, (size_t) nsplit * (size_t) p->uc);
			if (rc != MF_OK) return rc;
			p->part_cap = nsplit;
		}
		m.part = p->part_dev;
	}
	hipLaunchKernelGGL(fn, dim3(ublocks, nsplit), dim3(threads), lds, p->stream, m);
	MF_HIP(hipGetLastError());
	if (nsplit > 1) {
		hipLaunchKernelGGL(mf::merge_splits_kernel, dim3((p->uc + 255) / 256), dim3(256), 0, p->stream, m, nsplit);
		MF_HIP(hipGetLastError());
	}
	return MF_OK;
}

extern "C" {

int mf_plan_recommend(mf_plan *p, int32_t *best)
{
	if (!p || (!best && p->uc > 0)) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	if (p->uc == 0) return MF_OK;
	const bool use_mfma = !p->cfg.rec_exact;   // MF_RECOMMEND_IMPL=mfma (default) | exact
	mf::RecArgs ex;
	ex.users = p->uc;
	ex.items = p->items;
	ex.K = p->K;
	ex.ldl = p->ldl;
	ex.ldr = p->ldr;
	ex.L = p->Lbuf[p->cur];
	ex.R = p->Rbuf[p->cur];
	ex.csr_ptr = p->csr_ptr;
	ex.csr_idx = p->mask_idx ? p->mask_idx : p->csr_idx;
	ex.best = p->best_dev;
	ex.ulist = nullptr;
	ex.cand = nullptr;
	if (!use_mfma) {
		const int grid = (p->uc + mf::kRT - 1) / mf::kRT;
		hipLaunchKernelGGL(mf::recommend_kernel, dim3(grid), dim3(256), 0, p->stream, ex);
		MF_HIP(hipGetLastError());
		p->last_uncertain = -1;
	} else {
		// pass 1: scores on the FP64 matrix cores + certification margin; pass 2: exact re-scoring of the rest
		{
			const int rc1 = launch_recommend_pass1(p, nullptr);
			if (rc1 != MF_OK) return rc1;
		}
		// the list travels with the count of uncertified users: one synchronisation when nobody needs the exact pass
		// (the common case: 0 of 1e6 users at cfg4), a second copy of the list only behind an exact pass
		int cnt = 0;
		MF_HIP(hipMemcpyAsync(&cnt, p->ucount, sizeof(int), hipMemcpyDeviceToHost, p->stream));
		MF_HIP(hipMemcpyAsync(best, p->best_dev, (size_t) p->uc * sizeof(int), hipMemcpyDeviceToHost, p->stream));
		MF_HIP(hipStreamSynchronize(p->stream));
		p->last_uncertain = cnt;
		if (cnt == 0) return MF_OK;
		ex.users = cnt;
		ex.ulist = p->ulist;
		hipLaunchKernelGGL(mf::recommend_kernel, dim3((cnt + mf::kRT - 1) / mf::kRT), dim3(256), 0, p->stream, ex);
		MF_HIP(hipGetLastError());
	}
	MF_HIP(hipMemcpyAsync(best, p->best_dev, (size_t) p->uc * sizeof(int), hipMemcpyDeviceToHost, p->stream));
	MF_HIP(hipStreamSynchronize(p->stream));
	return MF_OK;
}

int mf_plan_recommend_scored(mf_plan *p, mf_candidate *out)
{
	if (!p || (!out && p->uc > 0)) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	if (p->uc == 0) return MF_OK;
	if (!p->cand_dev) {
		const int rc = dev_alloc(&p->cand_dev, (size_t) p->uc);
		if (rc != MF_OK) return rc;
	}
	mf::RecArgs ex;
	ex.users = p->uc;
	ex.items = p->items;
	ex.K = p->K;
	ex.ldl = p->ldl;
	ex.ldr = p->ldr;
	ex.L = p->Lbuf[p->cur];
	ex.R = p->Rbuf[p->cur];
	ex.csr_ptr = p->csr_ptr;
	ex.csr_idx = p->mask_idx ? p->mask_idx : p->csr_idx;
	ex.best = p->best_dev;
	ex.ulist = nullptr;
	ex.cand = p->cand_dev;
	hipLaunchKernelGGL(mf::recommend_kernel, dim3((p->uc + mf::kRT - 1) / mf::kRT), dim3(256), 0, p->stream, ex);
	MF_HIP(hipGetLastError());
	MF_HIP(hipMemcpyAsync(out, p->cand_dev, (size_t) p->uc * sizeof(mf_candidate), hipMemcpyDeviceToHost, p->stream));
	MF_HIP(hipStreamSynchronize(p->stream));
	return MF_OK;
}

int mf_plan_recommend_scored_users(mf_plan *p, const int32_t *users, int32_t n, mf_candidate *out)
{
	if (!p || n < 0 || (n > 0 && (!users || !out)) || n > p->uc) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	for (int32_t t = 0; t < n; ++t)
		if (users[t] < 0 || users[t] >= p->uc) return MF_ERR_ARGUMENT;
	MF_HIP(hipSetDevice(p->device));
	if (n == 0) return MF_OK;
	if (!p->cand_dev) {
		const int rc = dev_alloc(&p->cand_dev, (size_t) p->uc);
		if (rc != MF_OK) return rc;
	}
	MF_HIP(hipMemcpyAsync(p->ulist, users, (size_t) n * sizeof(int), hipMemcpyHostToDevice, p->stream));
	mf::RecArgs ex;
	ex.users = n;
	ex.items = p->items;
	ex.K = p->K;
	ex.ldl = p->ldl;
	ex.ldr = p->ldr;
	ex.L = p->Lbuf[p->cur];
	ex.R = p->Rbuf[p->cur];
	ex.csr_ptr = p->csr_ptr;
	ex.csr_idx = p->mask_idx ? p->mask_idx : p->csr_idx;
	ex.best = p->best_dev;
	ex.ulist = p->ulist;
	ex.cand = p->cand_dev;   // written at the user's own index
	hipLaunchKernelGGL(mf::recommend_kernel, dim3((n + mf::kRT - 1) / mf::kRT), dim3(256), 0, p->stream, ex);
	MF_HIP(hipGetLastError());
	// only the n requested records travel: packed on the device in list order
	if (!p->cand_pack) {
		const int rc = dev_alloc(&p->cand_pack, (size_t) p->uc);
		if (rc != MF_OK) return rc;
	}
	hipLaunchKernelGGL(mf::pack_candidates_kernel, dim3((n + 255) / 256), dim3(256), 0, p->stream, p->cand_dev, p->ulist,
	                   n, p->cand_pack);
	MF_HIP(hipGetLastError());
	MF_HIP(hipMemcpyAsync(out, p->cand_pack, (size_t) n * sizeof(mf_candidate), hipMemcpyDeviceToHost, p->stream));
	MF_HIP(hipStreamSynchronize(p->stream));
	return MF_OK;
}

double mf_backend_recommend_margin(int features) { return 8.0 * (double) (features + 8) * 1.1102230246251565e-16; }

int mf_plan_recommend_filter(mf_plan *p, mf_filter *out, double *norm, double *rmax)
{
	if (!p || (p->uc > 0 && (!out || !norm)) || !rmax) return MF_ERR_ARGUMENT;
	if (!p->have_factors) return MF_ERR_STATE;
	MF_HIP(hipSetDevice(p->device));
	*rmax = 0.0;
	if (p->uc == 0) return MF_OK;
	if (!p->filt_dev) {
		const int rc = dev_alloc(&p->filt_dev, (size_t) p->uc);
		if (rc != MF_OK) return rc;
	}
	const int rc = launch_recommend_pass1(p, p->filt_dev);
	if (rc != MF_OK) return rc;
	unsigned long long bits = 0;
	MF_HIP(hipMemcpyAsync(out, p->filt_dev, (size_t) p->uc * sizeof(mf_filter), hipMemcpyDeviceToHost, p->stream));
	MF_HIP(hipMemcpyAsync(norm, p->lnorm, (size_t) p->uc * sizeof(double), hipMemcpyDeviceToHost, p->stream));
	MF_HIP(hipMemcpyAsync(&bits, p->rmax_bits, sizeof bits, hipMemcpyDeviceToHost, p->stream));
	MF_HIP(hipStreamSynchronize(p->stream));
	memcpy(rmax, &bits, sizeof bits);   // a NaN norm arrives as NaN: the caller then certifies nobody
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
	                   p->Lbuf[p->cur], p->Rbuf[p->cur], p->uc, p->items, p->K, p->ldl, p->ldr, dB);
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
	int n;
	if (p->sweep.dma)
		n = snprintf(buf, (size_t) buflen,
		             "sweep_dma_kernel<KT=%d,NPASS=%d> K=%d pitch=%d/%d nch=%d row_bytes=%d lds=%zu long_rows=%d/%d coop_nch=%d double_buffered=%d/%d(nch=%d) mid_rows=%d/%d(nch=%d) wave_pair=%d/%d(nch=%d) trio=%d/%d",
		             p->sweep.kt, p->sweep.kt ? (p->K / 2 + 63) / 64 : p->sweep.kpmax, p->K, p->ldl, p->ldr, p->nch, p->sweep.row_bytes,
		             p->lds_bytes, p->n_long[0] + (p->coop_all[0] ? p->items : 0), p->n_long[1] + (p->coop_all[1] ? p->uc : 0),
		             p->coop_all[0] || p->coop_all[1] ? p->nch_coop : 0, (int) p->use_db[0], (int) p->use_db[1], p->nch_db, p->n_mid[0], p->n_mid[1], p->nch_mid, (int) p->use_pair[0], (int) p->use_pair[1], p->nch_pair, (int) p->use_trio[0], (int) p->use_trio[1]);
	else
		n = snprintf(buf, (size_t) buflen, "sweep_kernel<KT=%d,KPMAX=%d> K=%d nch=%d stride=%d lds=%zu",
		             p->sweep.kt, p->sweep.kpmax, p->K, p->nch, p->stride, p->lds_bytes);
	// how mf_plan_iterate runs an iteration: the two sweeps above, or errors + streams (mf_stream.hip.h)
	if (n > 0 && n < buflen) {
		if (!p->es_mode)
			snprintf(buf + n, (size_t) (buflen - n), " iterate=sweeps");
		else
			snprintf(buf + n, (size_t) (buflen - n),
			         " iterate=errors+resident-streams(segments=%d x<=%d, %d-column slices of Y in LDS, %d workgroups, lds=%zu/%zu)",
			         p->es_nseg, p->es_nch, p->res_sw, p->res_nwg, p->es_lds_errors, p->res_lds);
	}
	// the environment switches this plan was created under, when any differs from its default (mf_config.hip.h)
	const std::string cfg = p->cfg.describe();
	const size_t used = strlen(buf);
	if (!cfg.empty() && used + 1 < (size_t) buflen) snprintf(buf + used, (size_t) buflen - used, " config{%s}", cfg.c_str());
	return MF_OK;
}

/* ---------------------------------------------------------------------------------------- LEVEL 1 */

static int make_single_plan(const mf_problem *pr, int device, mf_plan **out)
{
	if (!pr || pr->users < 0 || pr->items < 0 || pr->features < 1 || pr->nnz < 0 || pr->iters < 0 ||
	    (pr->nnz > 0 && !pr->entries))
		return MF_ERR_ARGUMENT;
	mf_shard s;
	memset(&s, 0, sizeof s);
	s.users_total = pr->users;
	s.items = pr->items;
	s.features = pr->features;
	s.user_begin = 0;
	s.user_count = pr->users;
	s.nnz = pr->nnz;
	s.alpha = pr->alpha;
	s.device = device;
	return plan_create_impl(out, &s, pr->entries);
}

int mf_backend_run(const mf_problem *pr, double *L, double *R, int32_t *best, int device)
{
	if (!pr || !L || !R) return MF_ERR_ARGUMENT;   // L and R carry the initial factors in
	mf_plan *p = nullptr;
	int rc = make_single_plan(pr, device, &p);
	if (rc != MF_OK) return rc;
	rc = mf_plan_upload_factors(p, L, R);
	if (rc == MF_OK) rc = mf_plan_iterate(p, pr->iters);
	if (rc == MF_OK && best) rc = mf_plan_recommend(p, best);
	if (rc == MF_OK) rc = mf_plan_download_factors(p, L, R);
	mf_plan_destroy(p);
	return rc;
}

int mf_backend_run_top1(const mf_problem *pr, const double *L0, const double *R0, int32_t *best, int device)
{
	if (!pr || !L0 || !R0 || (!best && pr->users > 0)) return MF_ERR_ARGUMENT;
	mf_plan *p = nullptr;
	int rc = make_single_plan(pr, device, &p);
	if (rc != MF_OK) return rc;
	rc = mf_plan_upload_factors(p, L0, R0);
	if (rc == MF_OK) rc = mf_plan_iterate(p, pr->iters);
	if (rc == MF_OK) rc = mf_plan_recommend(p, best);
	mf_plan_destroy(p);
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
	int rc = make_single_plan(pr, device, &p);
	if (rc != MF_OK) return rc;
	rc = mf_plan_upload_factors(p, L, R);
	if (rc == MF_OK) rc = mf_plan_recommend(p, best);
	mf_plan_destroy(p);
	return rc;
}

#ifdef MF_OS_DIAG
// diagnostic build only: read and clear the records of ordered_sum_task_diag (tools/os_diag.py)
int mf_debug_read_os_diag(unsigned long long *out, int words)
{
	static unsigned long long zero[2 + 8 * 32];
	if (words > (int) (sizeof zero / sizeof zero[0])) words = (int) (sizeof zero / sizeof zero[0]);
	MF_HIP(hipDeviceSynchronize());
	MF_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(mf::mf_os_diag), sizeof(unsigned long long) * (size_t) words));
	MF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(mf::mf_os_diag), zero, sizeof zero));
	return MF_OK;
}
#endif

#ifdef MF_STAMPS
// diagnostic build only: read and clear the phase clocks of sweep_dma_kernel (tools/stamps.py)
int mf_debug_read_stamps(unsigned long long *out8)
{
	unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	MF_HIP(hipDeviceSynchronize());
	MF_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(mf::mf_stamp_buf), sizeof zero));
	MF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(mf::mf_stamp_buf), zero, sizeof zero));
	return MF_OK;
}

// per-wave clocks of the streams launch of the errors + streams iteration (tools/es_stamps.py)
int mf_debug_read_es_stamps(unsigned long long *out, int words)
{
	MF_HIP(hipDeviceSynchronize());
	MF_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(mf::mf_es_stamp_buf), sizeof(unsigned long long) * (size_t) words));
	return MF_OK;
}

// the same for recommend_mfma_kernel (tools/rec_stamps.py): waves 0 and 7 of workgroup 0
int mf_debug_read_rec_stamps(unsigned long long *out32)
{
	unsigned long long zero[32] = {};
	MF_HIP(hipDeviceSynchronize());
	MF_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(mf::mf_rec_stamp_buf), sizeof zero));
	MF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(mf::mf_rec_stamp_buf), zero, sizeof zero));
	return MF_OK;
}
#endif

}  // extern "C"

#include "mf_multi.hip.h"
