/*
 * matfact_hip.h -- C ABI of the MI355X (gfx950) backend for the matrix-factorisation hot path of
 * vladstojna/recommender-system.  Plain C: pointers and sizes only, callable from the reference's C
 * `main` (or from cgo/ctypes/JNI) with no C++ or torch types in any signature.
 *
 * The reference has no plugin API; the seam a maintainer would cut is the pair of calls its main makes
 * (matFact.c:124 and :127).  Each entry point below names the reference interface it replaces.
 *
 *   LEVEL 1 -- host-buffer drop-ins (what the reference's main would call)
 *     mf_backend_factorize   replaces  matrix_factorization()           matFact.c:29-59 (iteration loop)
 *     mf_backend_recommend   replaces  mat2d_prod() + print_output()    mat2d.c:100-113, matFact.c:10-27
 *     mf_backend_run         both, factors stay in HBM between the two  matFact.c:124-127
 *
 *   LEVEL 2 -- resident shard plan (one per GPU / per rank; what the MPI variant's per-rank state is)
 *     mf_plan_*              replaces  the per-rank body of matrix_factorization()  matFact-mpi.c:155-214
 *                            and compute_reduce_output()                            matFact-mpi.c:51-103
 *
 * Conventions: the caller owns every host buffer; the backend owns device memory for the duration of a
 * level-1 call or the lifetime of a plan.  Every function returns MF_OK (0) or a negative mf_status and
 * never calls exit() (the reference's die(), util.c:7-10, stays in the caller).  The library is HIP-only:
 * there is no CPU fallback -- with no usable GPU every compute entry returns MF_ERR_NO_DEVICE.
 */
#ifndef MATFACT_HIP_H
#define MATFACT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MATFACT_HIP_ABI_VERSION 5   /* 2: mf_shard.users_ext, seeded user sweep, scored recommend (2-D tiles)
                                       3: mf_backend_multi_last_timing, MF_MULTI_REDUCE=peer|rccl; the reserved
                                          MF_PLAN_RELAXED_ORDER flag is gone (never implemented: an atomic sum is
                                          order-nondeterministic and slower than the owner-computes gather)
                                       4: mf_shard.items_pitch / users_pitch, mf_backend_row_pitch, mf_plan_row_pitch
                                       5: mf_backend_multi_last_counters (one host thread per shard enqueues its
                                          iterations; MF_MULTI_THREADS=0 keeps the single enqueueing thread) */

/* == non_zero_entry, datatypes.h:10-15: the (user, item, rating) triple, 16 bytes, array-of-structs */
typedef struct mf_entry {
	int32_t row;
	int32_t col;
	double value;
} mf_entry;

/* The parsed `.in` header plus the entries (matFact.c:79-105; dataset_info datatypes.h:17-27). */
typedef struct mf_problem {
	int32_t users;          /* rows of A, rows of L                 */
	int32_t items;          /* columns of A, rows of R (R is kept transposed, matFact.c:117) */
	int32_t features;       /* K                                    */
	int32_t iters;
	double alpha;
	int64_t nnz;            /* the reference holds this in an int   */
	const mf_entry *entries; /* file order; the reference's inputs are (row, col)-sorted and the
	                            recommendation mask (print_output's cursor) relies on it */
} mf_problem;

typedef enum mf_status {
	MF_OK = 0,
	MF_ERR_ARGUMENT = -1,     /* NULL pointer, negative size, index out of range                    */
	MF_ERR_NO_DEVICE = -2,    /* no HIP device / device index out of range                          */
	MF_ERR_HIP = -3,          /* a HIP runtime call failed; mf_backend_last_hip_error() has its text */
	MF_ERR_NO_MEMORY = -4,    /* host or device allocation failed                                   */
	MF_ERR_UNSUPPORTED = -5,  /* shape outside what the kernels are built for (e.g. K too large)    */
	MF_ERR_STATE = -6         /* plan used before factors were uploaded, etc.                       */
} mf_status;

const char *mf_backend_strerror(int status);
const char *mf_backend_last_hip_error(void);
int mf_backend_abi_version(void);
int mf_backend_device_count(void);            /* >= 0, or a negative mf_status */

/* ------------------------------------------------------------------------------------------ LEVEL 1 */

/* L (users x K) and R (items x K), row-major fp64, are updated in place by `iters` iterations. */
int mf_backend_factorize(const mf_problem *p, double *L, double *R, int device);

/* best[i] = arg max_j (L R^T)[i][j] over the items user i has NOT rated (strict '>', ascending j, so the
 * lowest index wins ties); -1 when user i rated every item (the reference then prints no line). */
int mf_backend_recommend(const mf_problem *p, const double *L, const double *R, int32_t *best, int device);

/* factorize + recommend with the factors resident in HBM in between; L/R receive the final factors
 * (either may be NULL if the caller does not want them back). */
int mf_backend_run(const mf_problem *p, double *L, double *R, int32_t *best, int device);

/* The same when only the recommendation list is wanted -- what the reference's main prints (matFact.c:127):
 * the initial factors go in, nothing but best[] comes back (no device-to-host copy of L and R). */
int mf_backend_run_top1(const mf_problem *p, const double *L0, const double *R0, int32_t *best, int device);

/* The same on several GPUs of ONE process: users are cut into ndev contiguous blocks balanced by entry count,
 * L blocks are private, R is replicated and summed after every item sweep (the decomposition of
 * matFact-mpi.c:155-214 with the 8x1 grid of mpiutil.c:54-88; items are cut instead when items > users).  The sum
 * is the MPI_Iallreduce of matFact-mpi.c:207-208: environment MF_MULTI_REDUCE=peer (default) uses a hand-written
 * peer-to-peer reduce over xGMI (needs peer access between distinct devices, MF_ERR_UNSUPPORTED otherwise),
 * MF_MULTI_REDUCE=rccl uses ncclAllReduce(ncclDouble, ncclSum) on a communicator made by ncclCommInitAll (needs
 * distinct devices).  The reduce runs on its own stream beside the user sweep.  devices[] lists HIP ordinals; with
 * the peer reducer an ordinal may repeat (several shards on one GPU -- how the path is tested on a one-GPU box).
 * ndev <= 16. */
int mf_backend_run_multi(const mf_problem *p, double *L, double *R, int32_t *best, const int *devices, int ndev);
/* Host wall-clock of the last mf_backend_run_multi of this process: set-up (bucketing + plan builds + uploads),
 * iterations, recommendations; info[0] = shards, info[1] = reducer (0 peer, 1 rccl), info[2] = 1 when the shards
 * were slices of the caller's array (input sorted by the cut key: no bucketing pass).  Any pointer may be NULL. */
int mf_backend_multi_last_timing(double *setup_s, double *iterate_s, double *recommend_s, int *info);
/* More of the same run: enqueue_s = host time the slowest enqueueing thread spent issuing the iterations' launches, event
 * records and waits (everything of iterate_s but the final synchronize; small against iterate_s = the host is not the
 * bound); entry_passes = passes of the host over all nnz entries during set-up (1: the counting pass; 2: + the stable
 * scatter of an input not sorted by the cut key) -- independent of the shard count; host_threads = threads that
 * enqueued (the shard count, or 1 with MF_MULTI_THREADS=0).  Any pointer may be NULL. */
int mf_backend_multi_last_counters(double *enqueue_s, int64_t *entry_passes, int *host_threads);

/* ------------------------------------------------------------------------------------------ LEVEL 2 */

typedef struct mf_plan mf_plan;

/* One contiguous block of users (all of them for a single-GPU run) with its entries in SoA form. */
typedef struct mf_shard {
	int32_t users_total;
	int32_t items;
	int32_t features;
	int32_t user_begin;      /* first user of this shard (BLOCK_LOW, mpiutil.h:8)        */
	int32_t user_count;      /* users in this shard (BLOCK_SIZE, mpiutil.h:10-11)        */
	int64_t nnz;             /* entries whose row lies in [user_begin, user_begin+count) */
	const int32_t *row;      /* GLOBAL user ids, file order                              */
	const int32_t *col;
	const double *val;
	double alpha;
	int32_t device;          /* HIP device ordinal                                       */
	int32_t flags;           /* MF_PLAN_* bits                                           */
	void *items_ext[2];      /* optional caller-owned DEVICE buffers (items*features doubles each) for the
	                            two generations of R, e.g. torch tensors handed to a collective; NULL = own */
	void *users_ext[2];      /* the same for the two generations of this shard's L block (user_count*features
	                            doubles each); only a 2-D tile needs them (L summed over the grid row) */
	int32_t items_pitch;     /* row pitch, in doubles, of the caller-owned items_ext / users_ext buffers: 0 = features */
	int32_t users_pitch;     /* (rows packed); otherwise even and >= features -- mf_backend_row_pitch(features) is the
	                            pitch the plan gives its own buffers (rows padded to whole 128-byte lines where that
	                            saves gathered lines); the buffers then hold rows * pitch doubles */
} mf_shard;

/* A TILE of the reference's 2-D process grid (matFact-mpi.c:155-214, grid from create_balanced_grid,
 * mpiutil.c:54-88) is a shard that also holds only a block of the items: the caller passes `items` = the
 * block's item count and `col` relative to the block's first item (entries[n].col - offset_col,
 * matFact-mpi.c:193), exactly as user ids are relative to user_begin inside the plan.  Item indices that
 * mf_plan_recommend* return are then block-relative too. */

#define MF_PLAN_DEFAULT 0   /* no flag bits are defined: every sum is formed in the serial order */

int mf_plan_create(mf_plan **out, const mf_shard *shard);
/* Row pitch (doubles) the plan uses for factor buffers it owns, for this K: features, or features rounded up so that
 * a row is a whole number of 128-byte lines (a gathered row of 8K bytes otherwise touches a line more than its bytes
 * wherever it happens to start: 80-byte rows 1.5 lines on average instead of 1).  The pitch of a live plan's L and R
 * buffers -- what mf_plan_items_next() and friends point at -- is reported by mf_plan_row_pitch. */
int mf_backend_row_pitch(int features);
int mf_plan_row_pitch(mf_plan *plan, int32_t *users_pitch, int32_t *items_pitch);
void mf_plan_destroy(mf_plan *plan);

/* hipStream_t as void*; NULL = the plan's own stream.  All plan work is enqueued on it. */
int mf_plan_set_stream(mf_plan *plan, void *hip_stream);

/* host -> HBM: this shard's rows of L (user_count x K) and the whole of R (items x K). */
int mf_plan_upload_factors(mf_plan *plan, const double *L_block, const double *R);
int mf_plan_download_factors(mf_plan *plan, double *L_block, double *R);

/* Single-shard iteration loop: per iteration one item sweep and one user sweep from the frozen
 * generation into the next one, then flip (matFact.c:36-54; the two mat2d_copy are the ping-pong). */
int mf_plan_iterate(mf_plan *plan, int iters);

/* Sharded iteration, mirroring matFact-mpi.c:185-209 --
 *   mf_plan_sweep_items: R_next = (seed_from_old ? R_cur : 0) + sum over LOCAL entries   (:187,:190-205)
 *   mf_plan_sweep_users: L_next = L_cur + sum over local entries (L is private to the shard)
 *   caller SUM-all-reduces the buffer mf_plan_items_next() over the ranks             (:208)
 *   mf_plan_flip: next becomes current                                               */
int mf_plan_sweep_items(mf_plan *plan, int seed_from_old);
int mf_plan_sweep_users(mf_plan *plan);
void *mf_plan_items_next(mf_plan *plan);     /* device pointer, items rows of items_pitch doubles */
void *mf_plan_items_current(mf_plan *plan);
int mf_plan_flip(mf_plan *plan);

/* 2-D tiles: the L block is shared by the ranks of a grid row and summed over them like R is over a grid
 * column (the two MPI_Iallreduce of matFact-mpi.c:207-208) --
 *   mf_plan_sweep_users_seeded: L_next = (seed_from_old ? L_cur : 0) + sum over LOCAL entries  (:188)
 *   caller SUM-all-reduces mf_plan_users_next() over the grid row, mf_plan_items_next() over the grid column. */
int mf_plan_sweep_users_seeded(mf_plan *plan, int seed_from_old);
void *mf_plan_users_next(mf_plan *plan);     /* device pointer, user_count rows of users_pitch doubles */
void *mf_plan_users_current(mf_plan *plan);

/* Recommendations for this shard's users against the current R; best has user_count entries.
 * Default form: scores on the FP64 matrix cores (MFMA), every user whose best/second-best margin is not
 * provably larger than the rounding bound re-scored in the reference's exact order, so the result is the
 * reference's arg-max in all cases.  MF_RECOMMEND_IMPL=exact forces the exact form for every user. */
int mf_plan_recommend(mf_plan *plan, int32_t *best);
/* users the last mf_plan_recommend sent through the exact pass (-1 when the exact form ran for all) */
int mf_plan_recommend_info(mf_plan *plan, int64_t *exact_pass_users);

/* Partial result of the sequential scan of print_output (matFact.c:13-23) over this plan's items, in a form
 * that can be combined over the item blocks of a grid row (what MPI_Reduce(max_cmp) does at matFact-mpi.c:98):
 * the scan keeps the FIRST unrated item until a strictly greater score appears, and a NaN score never compares
 * greater, so per user the partial state is
 *   best/score  arg-max and max over the unrated items with a non-NaN score, lowest index on ties (-1: none),
 *               score being the reference's B[i][j] bit for bit (sequential k, unfused);
 *   first       the first unrated item (-1: every item rated);  first_nan  its score is NaN.
 * Combining blocks left to right: first = the first block's that has one; best = the greater score, the earlier
 * block on ties; answer = first < 0 ? -1 : first_nan ? first : best.  Exact form for every user. */
typedef struct mf_candidate {
	double score;
	int32_t best;
	int32_t first;
	int32_t first_nan;
	int32_t reserved;
} mf_candidate;
int mf_plan_recommend_scored(mf_plan *plan, mf_candidate *out);   /* user_count entries */
/* the same for n listed users of this shard only (local ids); out[t] belongs to users[t] */
int mf_plan_recommend_scored_users(mf_plan *plan, const int32_t *users, int32_t n, mf_candidate *out);

/* Pass 1 alone, for a certification ACROSS the item blocks of a grid row: per user the best and second-best
 * MATRIX-CORE (approximate) scores over this plan's unrated items, the arg-best (-1: none) and whether a
 * non-finite score was seen; norm[i] = ||L[i]||_2; *rmax = max_j ||R[j]||_2 over this plan's items.  Any score
 * is within 2*gamma_K*||l||*||r|| of the reference's, so with R = the largest rmax of the row and the margin
 * thr_i = mf_backend_recommend_margin(K) * norm[i] * R:  best_w - max(second_w, best_c for c != w) > thr_i and no
 * non-finite flag in any block  =>  arg_w IS the reference's answer.  Everyone else is re-scored exactly
 * (mf_plan_recommend_scored_users) and merged as described above. */
typedef struct mf_filter {
	double best, second;
	int32_t arg;
	int32_t nonfinite;
} mf_filter;
int mf_plan_recommend_filter(mf_plan *plan, mf_filter *out, double *norm, double *rmax);
double mf_backend_recommend_margin(int features);   /* 8 * (K + 8) * 2^-53 */

/* Dense predictions of this shard's users, B (user_count x items, row-major) = L R^T exactly as mat2d_prod
 * (mat2d.c:100-113) forms them; for debug dumps of SMALL instances (user_count*items <= 2^26). */
int mf_plan_predict(mf_plan *plan, double *B);

int mf_plan_synchronize(mf_plan *plan);

/* Per-launch device timing (HIP events on the plan's stream).  After mf_plan_timing(plan, 1) every
 * sweep launch is bracketed by events; mf_plan_timing_read drains them (synchronises) and reports
 * launch counts and summed milliseconds of the item sweeps and of the user sweeps. */
int mf_plan_timing(mf_plan *plan, int enable);
int mf_plan_timing_read(mf_plan *plan, int64_t *item_launches, double *item_ms,
                        int64_t *user_launches, double *user_ms);

/* Introspection for tests/bench: name of the sweep kernel variant chosen for this K, LDS bytes, chunk. */
int mf_plan_describe(mf_plan *plan, char *buf, int buflen);

#ifdef __cplusplus
}
#endif
#endif /* MATFACT_HIP_H */
