// mf_recommend.hip.h -- the recommendation step: exact form, FP64-MFMA form with certification, row norms,
// dense predictions for the debug dump.
#pragma once
#include "mf_common.hip.h"
#include "../../include/matfact_hip.h"   // mf_candidate

namespace mf {
// ------------------------------------------------------------------------------------------------
// Recommend kernel (exact form): fused L_blk * R^T with a masked running arg-max; B is never stored.
// One 256-thread workgroup owns 64 users and walks all item tiles (64 items) in ascending order.
// Each thread accumulates a 4x4 register block sequentially in k from 0.0 (mat2d.c:100-113 order), so
// every score equals the reference's B[i][j] bit for bit.  Rated (i, j) are excluded by a per-user
// cursor over the shard's CSR row (print_output's `aix`, matFact.c:13-23); ties keep the lower j.
// ------------------------------------------------------------------------------------------------
struct RecArgs {
	int users;    // users in this shard
	int items;
	int K;
	int ldl, ldr;                    // row pitch of L and of R in doubles (>= K)
	const double *__restrict__ L;    // users x K
	const double *__restrict__ R;    // items x K
	const int *__restrict__ csr_ptr; // users + 1
	const int *__restrict__ csr_idx; // item ids, ascending within a user
	int *__restrict__ best;          // users
	const int *__restrict__ ulist;   // optional: only these users (indices into the shard), `users` = its length
	mf_candidate *__restrict__ cand; // optional: the partial scan state per user (2-D tiles combine it over item blocks)
};

struct Cand {
	double bv;   // best non-NaN value so far
	int bi;      // its index, -1 if none
	int first;   // first unrated index, -1 if none
	int fnan;    // that first unrated score is NaN
};

__device__ __forceinline__ void cand_insert(Cand &c, double s, int j)
{
	const bool nan = s != s;
	if (c.first < 0) {
		c.first = j;
		c.fnan = nan;
	}
	if (!nan && (c.bi < 0 || s > c.bv)) {
		c.bv = s;
		c.bi = j;
	}
}

// left = earlier items, right = later items
__device__ __forceinline__ void cand_merge(Cand &l, const Cand &r)
{
	if (l.first < 0) {
		l.first = r.first;
		l.fnan = r.fnan;
	}
	if (r.bi >= 0 && (l.bi < 0 || r.bv > l.bv)) {
		l.bv = r.bv;
		l.bi = r.bi;
	}
}

constexpr int kRT = 64;   // users per workgroup, items per tile
constexpr int kRKC = 16;  // k chunk staged in LDS
constexpr int kRLD = kRT + 2;

__global__ void __launch_bounds__(256) recommend_kernel(RecArgs a)
{
	__shared__ double Ls[kRKC][kRLD];
	__shared__ double Rs[kRKC][kRLD];
	__shared__ unsigned long long maskw[kRT];

	const int tid = threadIdx.x;
	const int tx = tid & 15, ty = tid >> 4;
	const int i0 = blockIdx.x * kRT;
	const int K = a.K;

	// cursor state of the mask walker (threads 0..63: one user each)
	int cur = 0, cend = 0, nextcol = INT32_MAX;
	if (tid < kRT && i0 + tid < a.users) {
		const int uid = a.ulist ? a.ulist[i0 + tid] : i0 + tid;
		cur = a.csr_ptr[uid];
		cend = a.csr_ptr[uid + 1];
		nextcol = cur < cend ? a.csr_idx[cur] : INT32_MAX;
	}
	// running result, kept by the tx == 0 lane of each 16-lane group for its 4 users
	Cand run[4];
#pragma unroll
	for (int u = 0; u < 4; ++u) run[u] = Cand{0.0, -1, -1, 0};

	// staging roles: thread -> (row = tid / 4, 4 consecutive k starting at (tid % 4) * 4)
	const int srow = tid >> 2, sk = (tid & 3) * 4;
	const int suid = (i0 + srow < a.users) ? (a.ulist ? a.ulist[i0 + srow] : i0 + srow) : -1;

	for (int j0 = 0; j0 < a.items; j0 += kRT) {
		double acc[4][4];
#pragma unroll
		for (int u = 0; u < 4; ++u)
#pragma unroll
			for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;

		for (int kc = 0; kc < K; kc += kRKC) {
			{
				const int ij = j0 + srow;
#pragma unroll
				for (int x = 0; x < 4; ++x) {
					const int k = kc + sk + x;
					Ls[sk + x][srow] = (suid >= 0 && k < K) ? a.L[(size_t) suid * a.ldl + k] : 0.0;
					Rs[sk + x][srow] = (ij < a.items && k < K) ? a.R[(size_t) ij * a.ldr + k] : 0.0;
				}
			}
			__syncthreads();
			const int kmax = min(kRKC, K - kc);
			for (int k = 0; k < kmax; ++k) {
				double l[4], r[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) l[u] = Ls[k][ty * 4 + u];
#pragma unroll
				for (int v = 0; v < 4; ++v) r[v] = Rs[k][tx * 4 + v];
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int v = 0; v < 4; ++v) acc[u][v] = acc[u][v] + l[u] * r[v];
			}
			__syncthreads();
		}

		// rated-item mask of this tile, one 64-bit word per user
		if (tid < kRT) {
			unsigned long long m = 0;
			while (nextcol < j0 + kRT) {
				if (nextcol >= j0) m |= 1ull << (nextcol - j0);
				++cur;
				nextcol = cur < cend ? a.csr_idx[cur] : INT32_MAX;
			}
			maskw[tid] = m;
		}
		__syncthreads();

#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const unsigned long long m = maskw[ty * 4 + u];
			Cand c{0.0, -1, -1, 0};
#pragma unroll
			for (int v = 0; v < 4; ++v) {
				const int jj = tx * 4 + v;
				if (j0 + jj < a.items && !((m >> jj) & 1ull)) cand_insert(c, acc[u][v], j0 + jj);
			}
			// ordered merge over the 16 lanes that hold this user's 64 items (ascending tx)
#pragma unroll
			for (int d = 1; d < 16; d <<= 1) {
				Cand o;
				o.bv = __shfl_down(c.bv, d, 16);
				o.bi = __shfl_down(c.bi, d, 16);
				o.first = __shfl_down(c.first, d, 16);
				o.fnan = __shfl_down(c.fnan, d, 16);
				if (tx + d < 16) cand_merge(c, o);
			}
			if (tx == 0) cand_merge(run[u], c);
		}
		__syncthreads();   // maskw is rewritten by the next tile
	}

	if (tx == 0) {
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int i = i0 + ty * 4 + u;
			if (i < a.users) {
				const int uid = a.ulist ? a.ulist[i] : i;
				a.best[uid] = run[u].first < 0 ? -1 : (run[u].fnan ? run[u].first : run[u].bi);
				if (a.cand) a.cand[uid] = mf_candidate{run[u].bv, run[u].bi, run[u].first, run[u].fnan, 0};
			}
		}
	}
}


// out[t] = cand[ulist[t]]: the records of the listed users in list order (only those are copied to the host)
__global__ void __launch_bounds__(256) pack_candidates_kernel(const mf_candidate *__restrict__ cand,
                                                              const int *__restrict__ ulist, int n,
                                                              mf_candidate *__restrict__ out)
{
	const int t = blockIdx.x * 256 + threadIdx.x;
	if (t < n) out[t] = cand[ulist[t]];
}

// Dense B = L R^T (mat2d_prod, mat2d.c:100-113) for the debug dump of small instances: one thread per
// (i, j), sequential k from 0.0, separate multiply and add -- every element equals the reference's B[i][j].
__global__ void __launch_bounds__(256) predict_kernel(const double *__restrict__ L, const double *__restrict__ R,
                                                      int users, int items, int K, int ldl, int ldr,
                                                      double *__restrict__ B)
{
	const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (t >= (size_t) users * items) return;
	const double *l = L + (t / items) * (size_t) ldl, *r = R + (t % items) * (size_t) ldr;
	double b = 0.0;
	for (int k = 0; k < K; ++k) b = b + l[k] * r[k];
	B[t] = b;
}

// ------------------------------------------------------------------------------------------------
// Recommend kernel, MFMA form: scores on the FP64 matrix cores, answers certified exact.
//   pass 1 (this kernel)  S~ = L_blk * R^T with v_mfma_f64_16x16x4_f64; per user the best and the
//          second-best score over unrated items are tracked.  For ANY summation order and fusing,
//          |S~ - B| <= 2*gamma_K * |l|.|r| <= 2*gamma_K*||l||*||r||  (B = the reference's sequential,
//          unfused value), so when best - second > thr_i = c*(K+8)*2^-53*||L[i]||*max_j||R[j]|| (c = 8, a
//          4x margin) the approximate arg-max IS the reference's arg-max and no tie rule is involved.
//   pass 2 (recommend_kernel with `ulist`)  every other user -- near-ties, exact ties (the lowest index
//          must win), non-finite scores -- is re-scored in the reference's exact order.
// Tile: 512 threads = 8 waves (4 x 2) own 128 users x 128 items per step; each wave holds 2 x 4
// accumulator tiles of 16 x 16 (64 VGPRs); operands go through an LDS image stored [k-pair][row] whose
// fragment reads and staging writes are bank-conflict-free (kMLD2 below).
// MFMA operand maps (f64 16x16x4): A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15],
// D lane l reg r = D[(l>>4) + 4r][l&15].
// ------------------------------------------------------------------------------------------------
struct RecMfmaArgs {
	int users, items, K;
	int ldl, ldr;                              // row pitch of L and of R in doubles (>= K)
	const double *__restrict__ L;
	const double *__restrict__ R;
	const int *__restrict__ csr_ptr;
	const int *__restrict__ csr_idx;
	const double *__restrict__ lnorm;          // ||L[i]||_2 per user
	const unsigned long long *__restrict__ rnorm_max_bits;   // max_j ||R[j]||_2 as the bits of a double
	double thr_scale;                          // c * (K + 8) * 2^-53
	int *__restrict__ best;
	int *__restrict__ ulist;                   // out: users that need the exact pass
	int *__restrict__ ucount;
	mf_filter *__restrict__ filt;              // optional: report (best, second, arg, non-finite) instead of certifying
	// item split (small problems): blockIdx.y = split s scores items [s * split_items, (s + 1) * split_items) only and
	// writes its report to part[s * users + user]; merge_splits_kernel certifies over the splits afterwards
	int split_items;                           // 0: no split (gridDim.y == 1), else a multiple of 128
	mf_filter *__restrict__ part;
};

__global__ void __launch_bounds__(kWave) row_norm_kernel(const double *__restrict__ X, int rows, int K, int ld,
                                                          double *__restrict__ norm,
                                                          unsigned long long *__restrict__ max_bits)
{
	const int r = blockIdx.x * kWave + threadIdx.x;
	double s = 0.0;
	if (r < rows)
		for (int k = 0; k < K; ++k) {
			const double v = X[(size_t) r * ld + k];
			s += v * v;
		}
	s = sqrt(s);
	if (r < rows && norm) norm[r] = s;
	if (max_bits) {
		// NaN compares as a huge unsigned pattern: it poisons the maximum, which sends every user to pass 2
		unsigned long long b = (r < rows) ? (unsigned long long) __double_as_longlong(s) : 0ull;
		for (int d = 32; d >= 1; d >>= 1) {
			const unsigned long long o = __shfl_xor(b, d);
			b = o > b ? o : b;
		}
		if (threadIdx.x == 0) atomicMax(max_bits, b);
	}
}

constexpr int kMU = 128, kMI = 128;
// LDS image of a chunk: [k-pair][row] of double2 {x[row][2p], x[row][2p+1]}, 130 rows per k-pair.
//   fragment read (ds_read_b64): lanes 0..31 = 16 rows x (k, k+1) of one pair -> 256 contiguous bytes;
//   staging store (ds_write_b128): an 8-lane group = consecutive k-pairs of a row, pair stride 130*16 B = 8 banks
//   mod 64.  Both are bank-conflict-free (the first version's [k][row] image conflicted 4-way on the stores).
constexpr int kMLD2 = 130;
constexpr int kMThreads = 512;
// dynamic LDS of recommend_mfma_kernel<., KC, ARES> for a given K
constexpr size_t rec_mfma_lds(int K, int KC, bool ares)
{
	const size_t chunk = (size_t) (KC / 2) * kMLD2 * sizeof(double2);
	const size_t nch = (size_t) ((K + KC - 1) / KC);
	return (ares ? nch * chunk : 2 * chunk) + 2 * chunk;
}

struct Top2 {
	double b1, b2;
	int i1;
};

// b1 = -inf / i1 = -1 encode "no candidate"; all values are finite or -inf, so plain comparisons suffice
__device__ __forceinline__ void top2_merge(Top2 &a, const Top2 &b)
{
	const bool take = b.b1 > a.b1;
	const double lo1 = take ? a.b1 : b.b1;          // the smaller of the two bests
	const double hi2 = take ? b.b2 : a.b2;          // the winner's own runner-up
	a.b2 = lo1 > hi2 ? lo1 : hi2;
	a.b1 = take ? b.b1 : a.b1;
	a.i1 = take ? b.i1 : a.i1;
}

typedef double mf_d4 __attribute__((ext_vector_type(4)));

#ifdef MF_STAMPS
// diagnostic build only (tools/rec_stamps.py): shader-clock totals of wave 0 (slots 0..15) and wave 7 (16..31) of workgroup 0 --
// [0] tiles, [1] chunks, [2] mask walk, [3] prefetch issue, [4] k-steps, [5] landing wait, [6] barrier, [7] arg-max, [8] kernel
__device__ unsigned long long mf_rec_stamp_buf[32];
#define MF_RSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define MF_RSTAMP(var)
#endif

// 512 threads = 8 waves as 4 (user quarters of 32) x 2 (item halves of 64): two waves per SIMD, so one
// wave's staging, LDS traffic and arg-max bookkeeping run under the other's matrix instructions.
//   VEC   K even -> rows are 16-B aligned, 16-byte global loads
//   KC    k-chunk staged per barrier (a multiple of 4); the host picks one that divides K when it can, so that
//         no chunk is a short remainder that pays a whole fetch/stage/barrier for a few matrix instructions
//   ARES  the L block's image (128 users x K) stays resident in LDS for the whole kernel -- it is the same for
//         every item tile -- and only R chunks are staged (half the global->LDS traffic); needs
//         ceil(K/KC)*KC/2 + KC pairs of 2080 B to fit the 160 KB of a CU
//   BDMA  (needs VEC and ARES) R chunks go global -> LDS by LDS-DMA instead of through registers: one
//         instruction fills 64 rows of one k-pair of the image (lane = row, 16 B each); no staging registers,
//         no ds_write, no address arithmetic per piece.  Inline asm with a hand-placed s_waitcnt vmcnt(0) in
//         front of the chunk barrier: the builtin would make hipcc wait for the DMA before every fragment read.
template <bool VEC, int KC, bool ARES, bool BDMA = false>
__global__ void __launch_bounds__(kMThreads) recommend_mfma_kernel(RecMfmaArgs a)
{
	static_assert(!BDMA || (VEC && ARES), "LDS-DMA staging needs 16-B aligned rows and a resident L image");
	// (DMA staging of BOTH operands for larger K was measured slower than register staging: K=128 51.5 vs 53.5,
	// K=256 52.8 vs 55.1 TFLOP/s -- not instantiated.)
	constexpr int PC = KC / 2;                  // k-pairs per chunk
	constexpr int kChunkD2 = PC * kMLD2;        // double2 elements of one chunk image
	extern __shared__ double2 rec_lds[];
	// [ARES: all chunks of A | else: two A buffers][two B buffers]
	double2 *const Bs0 = rec_lds + (ARES ? ((a.K + KC - 1) / KC) * kChunkD2 : 2 * kChunkD2);
	__shared__ unsigned long long maskw[2][kMU][2];   // [tile parity][user][item half]
	__shared__ double red_b1[kMU][2], red_b2[kMU][2];
	__shared__ int red_i1[kMU][2], red_bad[kMU][2];

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wr = wave >> 1, wc = wave & 1;
	const int lr = lane & 15, lq = lane >> 4;
	const int i0 = blockIdx.x * kMU;
	const int K = a.K;
	const double ninf = -__builtin_inf();

	// mask walker: threads 0..127, one user each.  The next TWO rated items are held in registers: consuming one
	// only issues the load of the one after the next, so the walk does not wait on memory (it used to stall the
	// two walker waves -- and with them the whole workgroup at the next barrier -- for a load latency per tile).
	int cur = 0, cend = 0, nextcol = INT32_MAX, nextcol2 = INT32_MAX;
	if (tid < kMU && i0 + tid < a.users) {
		cur = a.csr_ptr[i0 + tid];
		cend = a.csr_ptr[i0 + tid + 1];
		nextcol = cur < cend ? a.csr_idx[cur] : INT32_MAX;
		nextcol2 = cur + 1 < cend ? a.csr_idx[cur + 1] : INT32_MAX;
	}

	// Running top-2 PER ROW AND ITEM HALF, in LDS (red_b1 / red_b2 / red_i1 / red_bad[row][wc]); a lane only keeps the
	// runner-up of each of the 8 rows it sees -- row(tu, r) = 32*wr + 16*tu + lq + 4*r -- as the rejection threshold
	// thr2[x].  (A top-2 per lane cost 41 registers of a kernel that sits at the 256-register limit, and its
	// thresholds were those of a sixteenth of the row's items each: sixteen times more slow-path visits.)
	double thr2[8];
#pragma unroll
	for (int x = 0; x < 8; ++x) thr2[x] = ninf;
	if (tid < kMU) {
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			red_b1[tid][h] = ninf;
			red_b2[tid][h] = ninf;
			red_i1[tid][h] = -1;
			red_bad[tid][h] = 0;
		}
	}

	// staging roles: a chunk image is 128 rows x PC k-pairs of 16 B; thread -> row tid/4, k-pairs 4m + (tid%4)
	// (one row pointer per matrix, constant offsets between the pieces; PC=10: the last round is half empty)
	constexpr int SP = (PC + 3) / 4;
	const int srow = tid >> 2, sq = tid & 3;
	const bool a_ok = i0 + srow < a.users;
	const double *__restrict__ aptr = a.L + (size_t) (a_ok ? i0 + srow : 0) * a.ldl;
	double2 av[ARES ? 1 : SP], bv[SP];

	auto load2 = [&](const double *__restrict__ rowp, bool ok, int k) {
		double2 v = make_double2(0.0, 0.0);
		if (VEC) {
			if (ok && k < K) v = *reinterpret_cast<const double2 *>(rowp + k);
		} else {
			if (ok && k < K) v.x = rowp[k];
			if (ok && k + 1 < K) v.y = rowp[k + 1];
		}
		return v;
	};
	// global -> registers for chunk (tile jt, k offset kc); zero outside the matrices
	auto fetch = [&](int jt, int kc) {
		const bool b_ok = jt + srow < a.items;
		const double *__restrict__ bptr = a.R + (size_t) (b_ok ? jt + srow : 0) * a.ldr;
#pragma unroll
		for (int m = 0; m < SP; ++m) {
			if (4 * m + 3 >= PC && 4 * m + sq >= PC) continue;   // only the last round of PC % 4 != 0 can be cut
			const int k = kc + 8 * m + 2 * sq;
			if (!ARES) av[m] = load2(aptr, a_ok, k);
			bv[m] = load2(bptr, b_ok, k);
		}
	};
	auto stage = [&](int buf) {
#pragma unroll
		for (int m = 0; m < SP; ++m) {
			if (4 * m + 3 >= PC && 4 * m + sq >= PC) continue;
			if (!ARES) rec_lds[buf * kChunkD2 + (4 * m + sq) * kMLD2 + srow] = av[m];
			Bs0[buf * kChunkD2 + (4 * m + sq) * kMLD2 + srow] = bv[m];
		}
	};
	// LDS-DMA staging of an R chunk: instruction t (t = wave, wave + 8, ...) covers k-pair t/2, rows 64*(t%2)..+63
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);
	const unsigned bs_lds = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) Bs0;
	auto dma_image = [&](const double *__restrict__ X, int ld, int first_row, int last_row, unsigned lds_base, int kc, int buf) {
#pragma unroll
		for (int t0 = 0; t0 < 2 * PC; t0 += 8) {
			const int t = t0 + wave_u;
			const int pr = t >> 1, rb = (t & 1) * 64, k = kc + 2 * pr;
			if (t < 2 * PC && k < K) {   // wave-uniform
				const int row = min(first_row + rb + lane, last_row);   // rows beyond the matrix are masked / never stored
				const char *g = reinterpret_cast<const char *>(X + (size_t) row * ld + k);
				const unsigned m0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned) ((buf * kChunkD2 + pr * kMLD2 + rb) * 16));
				asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0) : "memory");
			}
		}
	};
	auto dma_chunk = [&](int jt, int kc, int buf) { dma_image(a.R, a.ldr, jt, a.items - 1, bs_lds, kc, buf); };
	if (BDMA) {
		// pairs beyond K are never written: zero the staged buffers once so that they hold no NaN patterns
		for (int sl = tid; sl < 2 * kChunkD2; sl += kMThreads) Bs0[sl] = make_double2(0.0, 0.0);
		__syncthreads();
	}
	if (ARES) {
		// the whole L block once: pairs 0 .. nch*PC-1 (zero beyond K and beyond the last user)
		const int pairs = ((K + KC - 1) / KC) * PC;
		for (int sl = tid; sl < kMU * pairs; sl += kMThreads) {
			const int row = sl / pairs, pr = sl - row * pairs;
			const bool a_ok = i0 + row < a.users;
			rec_lds[pr * kMLD2 + row] = load2(a.L + (size_t) (a_ok ? i0 + row : 0) * a.ldl, a_ok, 2 * pr);
		}
	}

	// Can a score of this workgroup be non-finite at all?  |score| <= ||L[i]|| * ||R[j]|| (Cauchy-Schwarz): when the
	// largest of its 128 user norms times the largest item norm is a finite number well below the overflow threshold,
	// every partial sum of every score is finite and the arg-max step needs no NaN / inf screening.  A NaN or inf
	// anywhere in the rows involved makes a norm NaN or inf (compared as bit patterns, a NaN is the largest value).
	__shared__ unsigned long long lmax_bits[2];
	if (tid < kMU) {
		unsigned long long b = i0 + tid < a.users ? (unsigned long long) __double_as_longlong(a.lnorm[i0 + tid]) : 0ull;
		for (int d = 32; d >= 1; d >>= 1) {
			const unsigned long long o = __shfl_xor(b, d);
			b = o > b ? o : b;
		}
		if (lane == 0) lmax_bits[wave] = b;
	}

	// the items this workgroup scores: all of them, or split blockIdx.y of a small problem
	const int j_first = a.split_items ? (int) blockIdx.y * a.split_items : 0;
	const int j_end = a.split_items ? min(a.items, j_first + a.split_items) : a.items;
	int buf = 0;
#ifdef MF_STAMPS
	unsigned long long rs_tiles = 0, rs_chunks = 0, rs_mask = 0, rs_issue = 0, rs_k = 0, rs_land = 0, rs_bar = 0, rs_arg = 0;
#endif
	MF_RSTAMP(rt_begin);
	if (BDMA) {
		dma_chunk(j_first, 0, 0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	} else {
		fetch(j_first, 0);
		stage(0);
	}
	__syncthreads();
	bool all_finite;
	{
		const unsigned long long lb = lmax_bits[0] > lmax_bits[1] ? lmax_bits[0] : lmax_bits[1];
		const double bound = __longlong_as_double((long long) lb) * __longlong_as_double((long long) *a.rnorm_max_bits);
		all_finite = bound <= 1e300;   // false for NaN
	}
	for (int j0 = j_first; j0 < j_end; j0 += kMI) {
		mf_d4 acc[2][4];
#pragma unroll
		for (int tu = 0; tu < 2; ++tu)
#pragma unroll
			for (int ti = 0; ti < 4; ++ti) acc[tu][ti] = mf_d4{0.0, 0.0, 0.0, 0.0};

		// rated-item mask of this tile: bit jj of word w = item j0 + 64*w + jj is rated or beyond the last item.
		// Written at the START of the tile into the parity's copy: the chunk barriers below publish it before the
		// arg-max step reads it, and the other parity is not rewritten before every wave has passed them again.
		const int par = ((j0 - j_first) / kMI) & 1;
		MF_RSTAMP(rt_m0);
		if (tid < kMU) {
			unsigned long long m0 = 0, m1 = 0;
			while (nextcol < j0 + kMI) {
				const int o = nextcol - j0;
				if (o >= 64)
					m1 |= 1ull << (o - 64);
				else if (o >= 0)
					m0 |= 1ull << o;
				++cur;
				nextcol = nextcol2;
				nextcol2 = cur + 1 < cend ? a.csr_idx[cur + 1] : INT32_MAX;
			}
			const int left = j_end - j0;   // > 0
			if (left < 64) {
				m0 |= ~0ull << left;
				m1 = ~0ull;
			} else if (left < 128) {
				m1 |= ~0ull << (left - 64);
			}
			maskw[par][tid][0] = m0;
			maskw[par][tid][1] = m1;
		}
#ifdef MF_STAMPS
		rs_mask += __builtin_amdgcn_s_memtime() - rt_m0;
		++rs_tiles;
#endif

		for (int kc = 0; kc < K; kc += KC) {
			// next chunk (of this tile, or the first of the next tile): global loads fly behind the MFMAs
			const bool more = kc + KC < K || j0 + kMI < j_end;
			MF_RSTAMP(rt_c0);
			if (BDMA) {   // straight into the OTHER buffer (last read one chunk ago; every wave passed a barrier since)
				if (kc + KC < K)
					dma_chunk(j0, kc + KC, buf ^ 1);
				else if (j0 + kMI < j_end)
					dma_chunk(j0 + kMI, 0, buf ^ 1);
			} else if (kc + KC < K)
				fetch(j0, kc + KC);
			else if (j0 + kMI < j_end)
				fetch(j0 + kMI, 0);
			const double *Ab = reinterpret_cast<const double *>(rec_lds + (ARES ? (kc / KC) : buf) * kChunkD2);
			const double *Bb = reinterpret_cast<const double *>(Bs0 + buf * kChunkD2);
			MF_RSTAMP(rt_c1);
			auto kstep = [&](int ks) {
				// k = 4*ks + lq -> pair 2*ks + (lq >> 1), half lq & 1
				const int po = ((2 * ks + (lq >> 1)) * kMLD2) * 2 + (lq & 1);
				double fa[2], fb[4];
#pragma unroll
				for (int tu = 0; tu < 2; ++tu) fa[tu] = Ab[po + (32 * wr + 16 * tu + lr) * 2];
#pragma unroll
				for (int ti = 0; ti < 4; ++ti) fb[ti] = Bb[po + (64 * wc + 16 * ti + lr) * 2];
#pragma unroll
				for (int tu = 0; tu < 2; ++tu)
#pragma unroll
					for (int ti = 0; ti < 4; ++ti)
						acc[tu][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[tu], fb[ti], acc[tu][ti], 0, 0, 0);
			};
			if (kc + KC <= K) {
#pragma unroll
				for (int ks = 0; ks < KC / 4; ++ks) kstep(ks);
			} else {   // last chunk: skip the zero padding beyond K
				const int ksteps = (K - kc + 3) >> 2;
				for (int ks = 0; ks < ksteps; ++ks) kstep(ks);
			}
			// registers -> the OTHER buffer (last read one chunk ago; every wave passed a barrier since)
			MF_RSTAMP(rt_c2);
			if (BDMA)
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			else if (more)
				stage(buf ^ 1);
			MF_RSTAMP(rt_c3);
			__syncthreads();
#ifdef MF_STAMPS
			rs_issue += rt_c1 - rt_c0;
			rs_k += rt_c2 - rt_c1;
			rs_land += rt_c3 - rt_c2;
			rs_bar += __builtin_amdgcn_s_memtime() - rt_c3;
			++rs_chunks;
#endif
			buf ^= 1;
		}
		MF_RSTAMP(rt_e0);

		// Cheap reject: after the first tiles almost no score beats its row's runner-up.  One compare per score register,
		// masks not even looked at: the 32 lane masks land in scalar registers and are OR-ed there, one scalar branch decides
		// (fmax() would add a canonicalising v_max per operand, a ballot two more vector instructions per row -- and every
		// vector instruction of this step waits for the matrix pipe of its SIMD).  !(v <= thr) is also true for a NaN.  Only
		// when the norms do not rule out non-finite scores (all_finite) a sum per row is formed as well: it is non-finite
		// whenever a score is NaN or +-inf (a sum that merely overflows only costs the slow path).
		constexpr int kUGT = 10;   // llvm::FCmpInst::FCMP_UGT: unordered or greater than
		unsigned long long rowm[8];
		unsigned long long anym = 0;
#pragma unroll
		for (int tu = 0; tu < 2; ++tu)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int x = tu * 4 + r;
				rowm[x] = __builtin_amdgcn_fcmp(acc[tu][0][r], thr2[x], kUGT) | __builtin_amdgcn_fcmp(acc[tu][1][r], thr2[x], kUGT) |
				          __builtin_amdgcn_fcmp(acc[tu][2][r], thr2[x], kUGT) | __builtin_amdgcn_fcmp(acc[tu][3][r], thr2[x], kUGT);
				anym |= rowm[x];
			}
		if (!all_finite) {
#pragma unroll
			for (int tu = 0; tu < 2; ++tu)
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const double sum = (acc[tu][0][r] + acc[tu][1][r]) + (acc[tu][2][r] + acc[tu][3][r]);
					rowm[tu * 4 + r] |= __builtin_amdgcn_fcmp(fabs(sum), 1.7976931348623157e308, kUGT);
					anym |= rowm[tu * 4 + r];
				}
		}
		if (anym != 0)
#pragma unroll
		for (int tu = 0; tu < 2; ++tu)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int x = tu * 4 + r;
				if (rowm[x] != 0) {
					// slow path (the whole wave, a few dozen times per row over the kernel): the lane's top-2 of the
					// row among its unrated items, merged over the 16 lanes that share the row, folded into the
					// row's state in LDS by the first of them, and the new runner-up handed back to all 16
					const int row = 32 * wr + 16 * tu + lq + 4 * r;
					const unsigned long long m = maskw[par][row][wc] >> lr;
					Top2 t{ninf, ninf, -1};
					int bd = 0;
#pragma unroll
					for (int ti = 0; ti < 4; ++ti) {
						const double v = acc[tu][ti][r];
						const bool open = !((m >> (16 * ti)) & 1ull);
						const bool fin = fabs(v) <= 1.7976931348623157e308;
						bd |= open && !fin;
						if (open && fin) top2_merge(t, Top2{v, ninf, j0 + 64 * wc + 16 * ti + lr});
					}
#pragma unroll
					for (int d = 1; d < 16; d <<= 1) {
						Top2 o;
						o.b1 = __shfl_xor(t.b1, d, 16);
						o.b2 = __shfl_xor(t.b2, d, 16);
						o.i1 = __shfl_xor(t.i1, d, 16);
						bd |= __shfl_xor(bd, d, 16);
						top2_merge(t, o);
					}
					if (lr == 0) {
						Top2 st{red_b1[row][wc], red_b2[row][wc], red_i1[row][wc]};
						top2_merge(st, t);
						red_b1[row][wc] = st.b1;
						red_b2[row][wc] = st.b2;
						red_i1[row][wc] = st.i1;
						if (bd) red_bad[row][wc] = 1;
						t.b2 = st.b2;
					}
					thr2[x] = __shfl(t.b2, lane & ~15);
				}
			}
#ifdef MF_STAMPS
		rs_arg += __builtin_amdgcn_s_memtime() - rt_e0;
#endif
	}
#ifdef MF_STAMPS
	if (blockIdx.x == 0 && blockIdx.y == 0 && (tid == 0 || tid == 7 * 64)) {
		unsigned long long *o = mf_rec_stamp_buf + (tid ? 16 : 0);
		o[0] += rs_tiles; o[1] += rs_chunks; o[2] += rs_mask; o[3] += rs_issue; o[4] += rs_k; o[5] += rs_land; o[6] += rs_bar;
		o[7] += rs_arg; o[8] += __builtin_amdgcn_s_memtime() - rt_begin;
	}
#endif

	// merge the two item halves (wc) of every row, then decide
	__syncthreads();
	if (tid < kMU && i0 + tid < a.users) {
		Top2 t{red_b1[tid][0], red_b2[tid][0], red_i1[tid][0]};
		const Top2 o{red_b1[tid][1], red_b2[tid][1], red_i1[tid][1]};
		top2_merge(t, o);
		const int bd = red_bad[tid][0] | red_bad[tid][1];
		if (a.split_items) {   // certification over the splits: merge_splits_kernel
			a.part[(size_t) blockIdx.y * a.users + i0 + tid] = mf_filter{t.b1, t.b2, t.i1, bd};
			return;
		}
		if (a.filt) {   // certification is the caller's, over several item blocks
			a.filt[i0 + tid] = mf_filter{t.b1, t.b2, t.i1, bd};
			return;
		}
		const double rmax = __longlong_as_double((long long) *a.rnorm_max_bits);
		const double thr = a.thr_scale * a.lnorm[i0 + tid] * rmax + 1e-300;
		const bool certain = !bd && (t.i1 < 0 || (t.b1 - t.b2) > thr);
		if (certain) {
			a.best[i0 + tid] = t.i1;
		} else {
			a.best[i0 + tid] = -2;
			a.ulist[atomicAdd(a.ucount, 1)] = i0 + tid;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Recommend kernel, MFMA form, TWO WORKGROUPS PER CU with the L fragments IN REGISTERS (even K <= 100: the headline
// configuration).  recommend_mfma_kernel keeps one 8-wave workgroup per CU: every chunk barrier, the arg-max step of a
// finished tile and the rated-item walk stop BOTH waves of every SIMD at the same moment, and the matrix pipe idles
// until the first fragment of the next chunk is back (clocks of waves 0 and 7, tools/rec_stamps.py: 34.8 k cycles per
// tile against the 25.6 k of its 2 x 25 x 8 matrix instructions; ~0.5-0.7 k idle around each of 5 barriers, ~2 k
// arg-max, 0.7 k masks).  Here a workgroup is 4 waves = 64 users x 128 items per step, ONE wave per SIMD, and two
// workgroups share a CU: the second wave of every SIMD belongs to the OTHER workgroup, which has its own barriers, so
// one workgroup's barrier, arg-max or mask walk runs under the other's matrix instructions -- no new synchronisation,
// just independent phases.  What makes two fit: a wave's L operand -- 32 users x K, the same for every item tile --
// lives in REGISTERS (2 doubles per lane and k-step: 100 VGPRs at K=100, beside 64 accumulators), so LDS holds only a
// ring of three 20-deep R chunks (10 k-pairs x 128 items x 16 B = 20 KB each): 60 KB per workgroup.  The transfer of
// chunk s+2 is issued under the matrix instructions of chunk s -- a chunk has two chunk times to land --, fragment reads
// per matrix instruction drop from 6/8 to 4/8, and the images need no padding rows (written only by LDS-DMA, 1 KB
// contiguous per instruction; the fragment read of 32 lanes is 256 contiguous bytes).  R streams from L2 twice as often
// (64 users per pass instead of 128): 3.8 TB/s of L2 -> LDS at cfg4, HBM traffic unchanged (the workgroups of an XCD
// walk R = 80 MB in step).  Masks, top-2 bookkeeping and certification are those of recommend_mfma_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kHU = 64, kHNB = 3, kHKmax = 100;
constexpr size_t kHStaticLds = 2 * kHU * 2 * 8 + kHU * 2 * (8 + 8 + 4 + 4) + 64;
// dynamic LDS: the ring of kHNB chunks of QC k-steps (2 QC k-pairs x 128 items x 16 B each)
inline size_t rec_mfma2_lds(int qc) { return (size_t) kHNB * (2 * qc) * kMI * sizeof(double2); }

// NC > 0: K == 20 * NC exactly -- every chunk whole, no branch of the tile body depends on K (hipcc's s_waitcnt placement
// follows the fragment pipeline only through straight-line code); NC == 0: any even K <= 100.
// The same kernel at other shapes: QC = k-steps per chunk (5: the 20-deep chunks above; 4 for K = 16 NC up to 128, two per
// CU as well; 8 for K = 256), TU = 16-user tiles per wave, WAVES = 4 (two workgroups per CU) or 8.  K = 256 does not fit the
// registers of a wave at 32 users (256 VGPRs for the L operand alone): there a wave owns 16 users x 64 items (TU = 1: 128
// VGPRs of L, 32 accumulators, 4 matrix instructions per k-step) and EIGHT waves form the 64-user workgroup, one per CU --
// the two waves of a SIMD then share their barriers, which costs little once each keeps the pipe full alone (ablation above).
template <int NC, int QC = 5, int TU = 2, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) recommend_mfma2_kernel(RecMfmaArgs a)
{
	static_assert(16 * TU * (WAVES / 2) == kHU && (2 * QC) % (WAVES / 2) == 0 && (2 * QC) / (WAVES / 2) <= 5, "shape");
	static_assert(NC > 0 || (QC == 5 && TU == 2 && WAVES == 4), "the general form exists for the 20-deep chunks only");
	constexpr int kHThreads = 64 * WAVES, kHKC = 4 * QC, kHPC = 2 * QC, kHQ = QC, kHChunkD2 = kHPC * kMI;
	constexpr int NCH = NC ? NC : kHKmax / kHKC, KSTEPS = NCH * kHQ;
	extern __shared__ double2 rec_lds[];   // ring of kHNB R chunks: [k-pair][128 items]
	const int K = a.K;
	__shared__ unsigned long long maskw[2][kHU][2];   // [tile parity][user][item half]
	__shared__ double red_b1[kHU][2], red_b2[kHU][2];
	__shared__ int red_i1[kHU][2], red_bad[kHU][2];
	__shared__ unsigned long long lmax_bits;

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wr = wave >> 1, wc = wave & 1;
	const int lr = lane & 15, lq = lane >> 4;
	const int i0 = blockIdx.x * kHU;
	const double ninf = -__builtin_inf();

	// mask walker: wave 0, one user per lane, the next two rated items held in registers (recommend_mfma_kernel)
	int cur = 0, cend = 0, nextcol = INT32_MAX, nextcol2 = INT32_MAX;
	if (tid < kHU && i0 + tid < a.users) {
		cur = a.csr_ptr[i0 + tid];
		cend = a.csr_ptr[i0 + tid + 1];
		nextcol = cur < cend ? a.csr_idx[cur] : INT32_MAX;
		nextcol2 = cur + 1 < cend ? a.csr_idx[cur + 1] : INT32_MAX;
	}
	double thr2[4 * TU];   // runner-up of each of the lane's rows: row(tu, r) = 16*TU*wr + 16*tu + lq + 4*r
#pragma unroll
	for (int x = 0; x < 4 * TU; ++x) thr2[x] = ninf;
	if (tid < kHU) {
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			red_b1[tid][h] = ninf;
			red_b2[tid][h] = ninf;
			red_i1[tid][h] = -1;
			red_bad[tid][h] = 0;
		}
		unsigned long long b = i0 + tid < a.users ? (unsigned long long) __double_as_longlong(a.lnorm[i0 + tid]) : 0ull;
		for (int d = 32; d >= 1; d >>= 1) {
			const unsigned long long o = __shfl_xor(b, d);
			b = o > b ? o : b;
		}
		if (lane == 0) lmax_bits = b;
	}
	// pairs beyond K are never transferred: the ring holds zeros there at first, not NaN patterns
	for (int sl = tid; sl < kHNB * kHChunkD2; sl += kHThreads) rec_lds[sl] = make_double2(0.0, 0.0);
	// the wave's L operand, once: k-step ks, user tile tu -> A[32*wr + 16*tu + lr][4*ks + lq] (zero beyond K / the last user)
	double fa[KSTEPS][TU];
#pragma unroll
	for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
		for (int tu = 0; tu < TU; ++tu) {
			const int row = i0 + 16 * TU * wr + 16 * tu + lr, k = 4 * ks + lq;
			fa[ks][tu] = row < a.users && k < K ? a.L[(size_t) row * a.ldl + k] : 0.0;
		}
	__syncthreads();

	// LDS-DMA of one R chunk (k offset kc of the tile `voff` points into) into ring slot `slot`: 20 instructions of 64
	// rows x 16 B, five per wave (k-pair wr + 2h, rows 64*wc..+63).  Scalar base + per-lane row offset: no vector
	// arithmetic per transfer -- every VALU instruction of a wave waits for the matrix pipe of its SIMD to drain
	// (tools/micro/valu_under_mfma.hip).  Returns how many this wave issued (pairs beyond K: none).
	const unsigned bs_lds = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) rec_lds;
	unsigned voff = 0;
	auto set_rows = [&](int jt) {
		// lane i of a transfer lands at byte 16 i of window w = wc of its k-pair's row: the item the chunk image keeps there
		const int item = ((lane >> 4) & 1) * 64 + (2 * wc + (lane >> 5)) * 16 + (lane & 15);
		const int row = min(jt + item, a.items - 1);      // rows beyond the matrix are masked
		voff = (unsigned) row * (unsigned) (a.ldr * 8);   // the host admits R below 4 GB only
	};
	auto dma_chunk = [&](int kc, int slot) -> int {
		int n = 0;
#pragma unroll
		for (int h = 0; h < kHPC / (WAVES / 2); ++h) {
			const int pr = wr + (WAVES / 2) * h, k = kc + 2 * pr;
			if (k < K) {   // wave-uniform
				const char *sbase = reinterpret_cast<const char *>(a.R + k);
				const unsigned m0 = bs_lds + (unsigned) ((slot * kHChunkD2 + pr * kMI + 64 * wc) * 16);
				asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0));
				++n;
			}
		}
		return n;
	};
	// at most n of this wave's transfers still in flight (hipcc does not count the asm transfers)
	auto wait_vm = [&](int n) {
		switch (n) {
		case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
		case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
		case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
		case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
		case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
		default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
		}
	};

	const int j_first = a.split_items ? (int) blockIdx.y * a.split_items : 0;
	const int j_end = a.split_items ? min(a.items, j_first + a.split_items) : a.items;
	// the chunk sequence: tiles in ascending order, k-chunks within; (pj, pk, pslot) = the next chunk to transfer
	int pj = j_first, pk = 0, pslot = 0;
	auto issue_next = [&]() -> int {
		if (pj >= j_end) return 0;
		if (pk == 0) set_rows(pj);
		const int n = dma_chunk(pk, pslot);
		pk += kHKC;
		if (pk >= K) {
			pk = 0;
			pj += kMI;
		}
		pslot = pslot == kHNB - 1 ? 0 : pslot + 1;
		return n;
	};
	issue_next();
	wait_vm(issue_next());   // chunk 0 has landed; chunk 1 may still be in flight
	__syncthreads();
	bool all_finite;
	{
		const double bound = __longlong_as_double((long long) lmax_bits) * __longlong_as_double((long long) *a.rnorm_max_bits);
		all_finite = bound <= 1e300;   // false for NaN
	}
	// R fragment of k-step q of the chunk in ring slot s: k = 4q + lq -> pair 2q + (lq >> 1), half lq & 1.  Within the 2 KB
	// row of a k-pair, item 64 wc + 16 ti + lr sits at byte 512 ti + 256 wc + 16 lr: a wave's four fragments of a k-step are
	// 512 B apart, the k-steps 4 KB, the slots 20 KB -- all multiples of 512, so every fragment read of a chunk is one base
	// register plus an immediate (ds_read2st64_b64) and the k-loop holds no vector arithmetic at all.
	const int boff = (lq >> 1) * (kMI * 2) + wc * 32 + lr * 2 + (lq & 1);
	auto frag = [&](int s, int q, double (&f)[4]) {
		const double *Bb = reinterpret_cast<const double *>(rec_lds) + s * (kHChunkD2 * 2) + boff;
#pragma unroll
		for (int ti = 0; ti < 4; ++ti) f[ti] = Bb[(8 * q + ti) * 64];
	};
	// The matrix stream of a wave has no gap of its own: the fragment of the NEXT k-step -- of this chunk, of the next
	// chunk, of the next tile -- is read in front of the matrix instructions of the current one.  For that the barrier
	// that publishes chunk s+1 stands in front of the LAST k-step of chunk s (whose transfer, issued under the first k-step
	// of chunk s-1, has had almost two chunk times), and the transfer of chunk s+2 goes into the slot of chunk s-1 -- whose
	// last fragment every wave had read before it passed that barrier one chunk ago.
	double fc[4];
	frag(0, 0, fc);
	int slot = 0, pending = 0;   // pending: this wave's transfers issued AFTER those of the chunk the next barrier publishes
	for (int j0 = j_first; j0 < j_end; j0 += kMI) {
		mf_d4 acc[TU][4];

		// rated-item mask of this tile into the parity's copy; published by the chunk barriers below, and the other
		// parity is not rewritten before every wave has passed them again (recommend_mfma_kernel)
		const int par = ((j0 - j_first) / kMI) & 1;
		if (tid < kHU) {
			unsigned long long m0 = 0, m1 = 0;
			while (nextcol < j0 + kMI) {
				const int o = nextcol - j0;
				if (o >= 64)
					m1 |= 1ull << (o - 64);
				else if (o >= 0)
					m0 |= 1ull << o;
				++cur;
				nextcol = nextcol2;
				nextcol2 = cur + 1 < cend ? a.csr_idx[cur + 1] : INT32_MAX;
			}
			const int left = j_end - j0;   // > 0
			if (left < 64) {
				m0 |= ~0ull << left;
				m1 = ~0ull;
			} else if (left < 128) {
				m1 |= ~0ull << (left - 64);
			}
			maskw[par][tid][0] = m0;
			maskw[par][tid][1] = m1;
		}

#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			const int kc = c * kHKC;
			if (NC || kc < K) {   // wave-uniform
				const int nq = NC ? kHQ : min(kHQ, (K - kc + 3) >> 2);   // k-steps of this chunk (only the last chunk can be short)
				const int nslot = slot == kHNB - 1 ? 0 : slot + 1;
#pragma unroll
				for (int q = 0; q < kHQ; ++q) {
					if (q < nq) {   // wave-uniform
						double fn[4];
						if (q == nq - 1) {
							wait_vm(pending);   // chunk s+1 has landed ...
							pending = 0;
#ifndef MF_REC_NOBAR
							__syncthreads();    // ... for every wave
#endif
							frag(nslot, 0, fn);
						} else {
							frag(slot, q + 1, fn);
						}
						// the reads stay in front of the matrix instructions (the scheduler would sink them behind the last
						// use of the current fragment's registers to save eight VGPRs -- and expose the LDS latency per k-step)
						__builtin_amdgcn_sched_barrier(0);
#pragma unroll
						for (int tu = 0; tu < TU; ++tu)
#pragma unroll
							for (int ti = 0; ti < 4; ++ti)
								acc[tu][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[c * kHQ + q][tu], fc[ti],
								                                                   c + q == 0 ? mf_d4{0.0, 0.0, 0.0, 0.0} : acc[tu][ti], 0, 0, 0);
						if (q == 0) {   // chunk s+2 under the matrix instructions just issued
							const int n = issue_next();
							if (q != nq - 1) pending = n;
						}
#pragma unroll
						for (int ti = 0; ti < 4; ++ti) fc[ti] = fn[ti];
					}
				}
				slot = nslot;
			}
		}

		// Cheap reject: after the first tiles almost no score beats its row's runner-up.  Every vector instruction of this
		// step costs matrix-pipe time (nothing else of the SIMD runs while an FP64 matrix instruction executes, and vice
		// versa), so the common case is ONE compare per score register -- 32 v_cmp whose lane masks land in scalar registers
		// and are OR-ed there -- and one scalar branch; fmax() would add a canonicalising v_max per operand and a ballot two
		// more instructions per row.  (!(v <= thr) is also true for a NaN.)  Only when the norms do not rule out non-finite
		// scores a sum per row is formed as well: it is non-finite whenever a score is NaN or +-inf.
		constexpr int kUGT = 10;   // llvm::FCmpInst::FCMP_UGT: unordered or greater than
		unsigned long long rowm[4 * TU];
		unsigned long long anym = 0;
#pragma unroll
		for (int tu = 0; tu < TU; ++tu)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int x = tu * 4 + r;
				rowm[x] = __builtin_amdgcn_fcmp(acc[tu][0][r], thr2[x], kUGT) | __builtin_amdgcn_fcmp(acc[tu][1][r], thr2[x], kUGT) |
				          __builtin_amdgcn_fcmp(acc[tu][2][r], thr2[x], kUGT) | __builtin_amdgcn_fcmp(acc[tu][3][r], thr2[x], kUGT);
				anym |= rowm[x];
			}
		if (!all_finite) {
#pragma unroll
			for (int tu = 0; tu < TU; ++tu)
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const double sum = (acc[tu][0][r] + acc[tu][1][r]) + (acc[tu][2][r] + acc[tu][3][r]);
					rowm[tu * 4 + r] |= __builtin_amdgcn_fcmp(fabs(sum), 1.7976931348623157e308, kUGT);
					anym |= rowm[tu * 4 + r];
				}
		}
#ifdef MF_REC_NOEPI
		if (j0 + kMI >= j_end)
#endif
		if (anym != 0)
#pragma unroll
		for (int tu = 0; tu < TU; ++tu)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int x = tu * 4 + r;
				// slow path exactly as in recommend_mfma_kernel
				if (rowm[x] != 0) {
					const int row = 16 * TU * wr + 16 * tu + lq + 4 * r;
					const unsigned long long m = maskw[par][row][wc] >> lr;
					Top2 t{ninf, ninf, -1};
					int bd = 0;
#pragma unroll
					for (int ti = 0; ti < 4; ++ti) {
						const double v = acc[tu][ti][r];
						const bool open = !((m >> (16 * ti)) & 1ull);
						const bool fin = fabs(v) <= 1.7976931348623157e308;
						bd |= open && !fin;
						if (open && fin) top2_merge(t, Top2{v, ninf, j0 + 64 * wc + 16 * ti + lr});
					}
#pragma unroll
					for (int d = 1; d < 16; d <<= 1) {
						Top2 o;
						o.b1 = __shfl_xor(t.b1, d, 16);
						o.b2 = __shfl_xor(t.b2, d, 16);
						o.i1 = __shfl_xor(t.i1, d, 16);
						bd |= __shfl_xor(bd, d, 16);
						top2_merge(t, o);
					}
					if (lr == 0) {
						Top2 st{red_b1[row][wc], red_b2[row][wc], red_i1[row][wc]};
						top2_merge(st, t);
						red_b1[row][wc] = st.b1;
						red_b2[row][wc] = st.b2;
						red_i1[row][wc] = st.i1;
						if (bd) red_bad[row][wc] = 1;
						t.b2 = st.b2;
					}
					thr2[x] = __shfl(t.b2, lane & ~15);
				}
			}
	}

	// merge the two item halves (wc) of every row, then decide
	__syncthreads();
	if (tid < kHU && i0 + tid < a.users) {
		Top2 t{red_b1[tid][0], red_b2[tid][0], red_i1[tid][0]};
		const Top2 o{red_b1[tid][1], red_b2[tid][1], red_i1[tid][1]};
		top2_merge(t, o);
		const int bd = red_bad[tid][0] | red_bad[tid][1];
		if (a.split_items) {   // certification over the splits: merge_splits_kernel
			a.part[(size_t) blockIdx.y * a.users + i0 + tid] = mf_filter{t.b1, t.b2, t.i1, bd};
			return;
		}
		if (a.filt) {   // certification is the caller's, over several item blocks
			a.filt[i0 + tid] = mf_filter{t.b1, t.b2, t.i1, bd};
			return;
		}
		const double rmax = __longlong_as_double((long long) *a.rnorm_max_bits);
		const double thr = a.thr_scale * a.lnorm[i0 + tid] * rmax + 1e-300;
		const bool certain = !bd && (t.i1 < 0 || (t.b1 - t.b2) > thr);
		if (certain) {
			a.best[i0 + tid] = t.i1;
		} else {
			a.best[i0 + tid] = -2;
			a.ulist[atomicAdd(a.ucount, 1)] = i0 + tid;
		}
	}
}

// Item split of a SMALL recommendation (few user blocks: cfg3 has 48 for 256 CUs): every split reported its top-2 per
// user; merged here in item order (top2_merge: the winner's runner-up is the larger of its own and the other splits'
// bests), then certified or reported exactly as the unsplit kernel does.
__global__ void __launch_bounds__(256) merge_splits_kernel(RecMfmaArgs a, int nsplit)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= a.users) return;
	Top2 t{-__builtin_inf(), -__builtin_inf(), -1};
	int bd = 0;
	for (int s = 0; s < nsplit; ++s) {
		const mf_filter f = a.part[(size_t) s * a.users + i];
		top2_merge(t, Top2{f.best, f.second, f.arg});
		bd |= f.nonfinite;
	}
	if (a.filt) {
		a.filt[i] = mf_filter{t.b1, t.b2, t.i1, bd};
		return;
	}
	const double rmax = __longlong_as_double((long long) *a.rnorm_max_bits);
	const double thr = a.thr_scale * a.lnorm[i] * rmax + 1e-300;
	const bool certain = !bd && (t.i1 < 0 || (t.b1 - t.b2) > thr);
	if (certain) {
		a.best[i] = t.i1;
	} else {
		a.best[i] = -2;
		a.ulist[atomicAdd(a.ucount, 1)] = i;
	}
}

}  // namespace mf
