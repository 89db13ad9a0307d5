// mf_kernels.hip.h -- gfx950 (CDNA4, wave64) kernels of the matrix-factorisation hot path.
//
// Both kernels are "owner computes": one wavefront owns one row of the factor it updates, so no atomics
// are needed and every floating-point sum is formed in exactly the order the serial reference forms it
// (matFact.c:41-53): the factor matrices come out BIT-IDENTICAL to matFact.c, not merely close.
// Build with -ffp-contract=off: the reference multiplies and adds separately (no FMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mf {

constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------------
// Sweep kernel.  One launch updates ONE factor from the frozen generation of both:
//   user sweep:  X = L (rows = users of the shard),  Y = R,  (ptr, idx, val) = CSR of the shard
//   item sweep:  X = R (rows = items),               Y = L,  (ptr, idx, val) = CSC of the shard
// For the owned row r and each of its entries n (file order == ascending idx for sorted inputs):
//   dot_n = sum_k X_old[r][k] * Y_old[idx_n][k]      sequential k, from 0.0       (mat2d.c:126-139)
//   e_n   = (alpha*2) * (val_n - dot_n)                                            (matFact.c:45)
//   X_new[r][k] = (...((seed + e_0*Y[idx_0][k]) + e_1*Y[idx_1][k]) + ...)          (matFact.c:47-51)
// where seed = X_old[r][k], or 0 for the non-root contribution of a sharded item sweep
// (matFact-mpi.c:187).  Mapping onto the wave, per chunk of <= nch entries of the row:
//   stage   the nch gathered Y rows are copied, coalesced (16 B per lane), into an LDS tile whose row
//           stride is odd in doubles, so both access patterns below are bank-conflict-free;
//   phase A lane n walks row n of the tile and forms dot_n sequentially in k (the serial order) with
//           X_old[r][k] as a scalar (SGPR) operand -> e_n;
//   phase B lane l owns k = l, l+64, ...; loops n ascending, acc[k] += e_n * tile[n][k] with e_n
//           broadcast by v_readlane -> the serial accumulation order into X[r][k].
// Algorithmic HBM bytes per entry and sweep: 8K (the gathered row) + 12 (idx, val).
// ------------------------------------------------------------------------------------------------
struct SweepArgs {
	int nrows;
	int K;
	int nch;      // entries per chunk (<= 64)
	int stride;   // LDS row stride in doubles (odd)
	int seed;     // 1: accumulate onto X_old, 0: onto zero
	double c2;    // alpha * 2
	const int *__restrict__ ptr;
	const int *__restrict__ idx;
	const double *__restrict__ val;
	const double *__restrict__ X_old;
	const double *__restrict__ Y_old;
	double *__restrict__ X_new;
	const int *__restrict__ rowlist;   // optional: the launch covers rows rowlist[0..nrows) instead of 0..nrows
	// products mode (extreme rows): the launch covers SEGMENTS of rows; segment s = entries [seg_beg, seg_end) of row
	// seg_row, whose scaled rows e_n * Y[idx_n][:] go to the scratch buffer at entry offset seg_out + (n - seg_beg)
	const int *__restrict__ seg_row;
	const int *__restrict__ seg_beg;
	const int *__restrict__ seg_end;
	const long long *__restrict__ seg_out;
	double *__restrict__ scratch;
	size_t scratch_entries;            // entries per 16-column slice of the scratch buffer
};

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
	return __hiloint2double(hi, lo);
}

template <int KT, int KPMAX>
__global__ void __launch_bounds__(kWave) sweep_kernel(SweepArgs a)
{
	extern __shared__ double tile[];
	const int K = KT > 0 ? KT : a.K;
	const int stride = KT > 0 ? (KT | 1) : a.stride;
	const int nch = a.nch;
	const int lane = threadIdx.x;

	for (int it = blockIdx.x; it < a.nrows; it += gridDim.x) {
		const int r = a.rowlist ? a.rowlist[it] : it;
		const int beg = a.ptr[r], end = a.ptr[r + 1];
		const double *__restrict__ xrow = a.X_old + (size_t) r * K;

		double acc[KPMAX];
#pragma unroll
		for (int kk = 0; kk < KPMAX; ++kk) {
			const int k = lane + kWave * kk;
			acc[kk] = (a.seed && k < K) ? xrow[k] : 0.0;
		}

		for (int c = beg; c < end; c += nch) {
			const int cnt = min(nch, end - c);
			int my_idx = 0;
			double my_val = 0.0;
			if (lane < cnt) {
				my_idx = a.idx[c + lane];
				my_val = a.val[c + lane];
			}
			// ---- stage: gathered rows -> LDS tile (row n of the tile = Y_old[idx_n][:])
			for (int n = 0; n < cnt; ++n) {
				const int j = __builtin_amdgcn_readlane(my_idx, n);
				const double *__restrict__ yrow = a.Y_old + (size_t) j * K;
				double *trow = tile + n * stride;
				if ((K & 1) == 0) {
#pragma unroll 2
					for (int q = lane; q < (K >> 1); q += kWave) {
						const double2 v = *reinterpret_cast<const double2 *>(yrow + 2 * q);
						trow[2 * q] = v.x;
						trow[2 * q + 1] = v.y;
					}
				} else {
					for (int q = lane; q < K; q += kWave)
						trow[q] = yrow[q];
				}
			}
			__syncthreads();
			// ---- phase A: lane n -> e_n (all lanes run it; lanes >= cnt produce unused garbage)
			double e;
			{
				const double *t = tile + (lane < nch ? lane : 0) * stride;   // lanes beyond the tile re-read row 0
				double dot = 0.0;
#pragma unroll 8
				for (int k = 0; k < K; ++k)
					dot = dot + xrow[k] * t[k];
				e = a.c2 * (my_val - dot);
			}
			// ---- phase B: lane l -> columns l, l+64, ...; entries in order
			for (int n = 0; n < cnt; ++n) {
				const double en = readlane_f64(e, n);
				const double *t = tile + n * stride;
#pragma unroll
				for (int kk = 0; kk < KPMAX; ++kk) {
					const int k = lane + kWave * kk;
					if (k < K)
						acc[kk] = acc[kk] + en * t[k];
				}
			}
			__syncthreads();
		}
#pragma unroll
		for (int kk = 0; kk < KPMAX; ++kk) {
			const int k = lane + kWave * kk;
			if (k < K)
				a.X_new[(size_t) r * K + k] = acc[kk];
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Sweep kernel, LDS-DMA form (the production kernel for even, compile-time K).
// Same arithmetic, same order, as sweep_kernel above; what changes is how the bytes move:
//   stage   one `global_load_lds_dwordx4` per gathered row (K/2 lanes x 16 B, per-lane source address,
//           wave-uniform LDS row base): the whole chunk -- up to 64 rows, 51 KB at K=100 -- is in flight
//           at once with no VGPR staging and no ds_write; one vmcnt(0) retires it.
//   tile    row stride = 16 B x (odd), so phase A's ds_read_b128 (lane n -> row n, 16-lane groups) is
//           bank-conflict-free while every row stays 16-B aligned for the DMA.
//   phase A lane n: 16 B of its row + 16 B of x (LDS broadcast) per step, two sequential mul/add pairs.
//   phase B lane l owns columns 2l, 2l+1 (+128 per pass): one ds_read_b128 per entry and pass, entries
//           in order, e_n broadcast through v_readlane into a scalar operand.
// LDS: [ x row: XS bytes ][ tile: nch rows x S bytes ].
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void mf_gvoid;
typedef __attribute__((address_space(3))) void mf_lvoid;

template <int KT>
struct DmaGeom {
	static_assert(KT % 2 == 0 && KT >= 2, "LDS-DMA sweep needs an even K");
	static constexpr int kPieces = KT / 2;                        // 16-B pieces per row
	static constexpr int kPasses = (kPieces + kWave - 1) / kWave; // DMA instructions per row
	static constexpr int kStride = 16 * (kPieces | 1);            // bytes, odd multiple of 16
	static constexpr int kXsBytes = ((KT * 8 + 255) / 256) * 256;
};

// KT > 0: K is a compile-time constant (phase A fully unrolled).  KT == 0: any even K up to 128*NPASS at run
// time (phase A unrolled by four) -- same data movement, so an unusual K does not fall back to the
// register-staged kernel.
// PRODUCTS = true: the "extreme row" form -- one wave per SEGMENT of a very long row; instead of accumulating, the
// scaled rows p_n[k] = e_n * y_n[k] (the rounded product the serial loop adds) are stored to a scratch buffer in
// entry order, and ordered_sum_kernel adds them up in that order afterwards.  Thousands of segments run in
// parallel, so a row rated by every user costs a chip-wide pass plus one serial chain of adds.
template <int KT, int NPASS, bool PRODUCTS = false>
__global__ void __launch_bounds__(kWave) sweep_dma_kernel(SweepArgs a)
{
	const int K = KT > 0 ? KT : a.K;
	const int P = K >> 1;                                   // 16-B pieces per row
	constexpr int NP = NPASS;                               // DMA instructions per row
	const int S = 16 * (P | 1);                             // tile row stride, odd multiple of 16 B
	const int xs_bytes = ((K * 8 + 255) / 256) * 256;
	extern __shared__ __attribute__((aligned(16))) char lds[];
	double2 *xs = reinterpret_cast<double2 *>(lds);
	char *tile = lds + xs_bytes;
	const int nch = a.nch;
	const int lane = threadIdx.x;
	const unsigned voff = (unsigned) lane * 16u;
	const unsigned long long ybase = (unsigned long long) a.Y_old;

	for (int it = blockIdx.x; it < a.nrows; it += gridDim.x) {
		const int r = PRODUCTS ? a.seg_row[it] : (a.rowlist ? a.rowlist[it] : it);
		const int beg = PRODUCTS ? a.seg_beg[it] : a.ptr[r];
		const int end = PRODUCTS ? a.seg_end[it] : a.ptr[r + 1];
		const double2 *__restrict__ xrow2 = reinterpret_cast<const double2 *>(a.X_old + (size_t) r * K);

		double2 acc[NP];
#pragma unroll
		for (int p = 0; p < NP; ++p) {
			const int q = lane + kWave * p;
			double2 v = make_double2(0.0, 0.0);
			if (q < P) {
				v = xrow2[q];
				xs[q] = v;
			}
			acc[p] = a.seed ? v : make_double2(0.0, 0.0);
		}

		// (idx, val) of a chunk are loaded one chunk ahead, so the gather of chunk c never waits on them
		int nx_idx = 0;
		double nx_val = 0.0;
		if (beg + lane < min(end, beg + nch)) {
			nx_idx = a.idx[beg + lane];
			nx_val = a.val[beg + lane];
		}
		for (int c = beg; c < end; c += nch) {
			const int cnt = min(nch, end - c);
			const int my_idx = nx_idx;
			const double my_val = nx_val;
			if (c + nch + lane < min(end, c + 2 * nch)) {
				nx_idx = a.idx[c + nch + lane];
				nx_val = a.val[c + nch + lane];
			}
			// ---- stage: one DMA per gathered row and pass
			for (int n = 0; n < cnt; ++n) {
				const int j = __builtin_amdgcn_readlane(my_idx, n);
				unsigned long long base = ybase + (unsigned long long) (unsigned) j * (unsigned long long) (K * 8);
				asm volatile("" : "+s"(base));   // keep the row base scalar
#pragma unroll
				for (int p = 0; p < NP; ++p) {
					const char *src = reinterpret_cast<const char *>(base) + voff + 1024u * p;
					if (lane + kWave * p < P)
						__builtin_amdgcn_global_load_lds((mf_gvoid *) src, (mf_lvoid *) (tile + n * S + 1024 * p),
						                                 16, 0, 0);
				}
			}
			__syncthreads();   // single-wave workgroup: this is the vmcnt(0)/lgkmcnt(0) that retires the DMA
			// ---- phase A
			double e;
			{
				const double2 *t2 = reinterpret_cast<const double2 *>(tile + (lane < nch ? lane : 0) * S);   // lanes beyond the tile re-read row 0
				double dot = 0.0;
				if (KT > 0) {
#pragma unroll
					for (int q = 0; q < KT / 2; ++q) {
						const double2 t = t2[q];
						const double2 x = xs[q];
						dot = dot + x.x * t.x;
						dot = dot + x.y * t.y;
					}
				} else {
					int q = 0;
					for (; q + 4 <= P; q += 4) {
						double2 t[4], x[4];
#pragma unroll
						for (int u = 0; u < 4; ++u) {
							t[u] = t2[q + u];
							x[u] = xs[q + u];
						}
#pragma unroll
						for (int u = 0; u < 4; ++u) {
							dot = dot + x[u].x * t[u].x;
							dot = dot + x[u].y * t[u].y;
						}
					}
					for (; q < P; ++q) {
						const double2 t = t2[q];
						const double2 x = xs[q];
						dot = dot + x.x * t.x;
						dot = dot + x.y * t.y;
					}
				}
				e = a.c2 * (my_val - dot);
			}
			if (PRODUCTS) {
				// scratch layout: [k-slice of 16 columns][entry][16 doubles] -- every (slice, entry) is one aligned
				// 128-B line and a slice is contiguous over the entries, so ordered_sum_kernel streams it linearly
				const char *tb = tile + voff;
				const size_t pos = (size_t) (a.seg_out[it] + (c - beg));
				for (int n = 0; n < cnt; ++n) {
					const double en = readlane_f64(e, n);
#pragma unroll
					for (int p = 0; p < NP; ++p) {
						const int q = lane + kWave * p;   // 16-B piece: slice q / 8, position q % 8 in its line
						if (q < P) {
							double2 t = *reinterpret_cast<const double2 *>(tb + n * S + 1024 * p);
							t.x = en * t.x;
							t.y = en * t.y;
							*reinterpret_cast<double2 *>(a.scratch + (((size_t) (q >> 3) * a.scratch_entries + pos + n) << 4) +
							                             2 * (q & 7)) = t;
						}
					}
				}
				__syncthreads();
				continue;
			}
			// ---- phase B
			const char *tb = tile + voff;
			int n = 0;
			for (; n + 4 <= cnt; n += 4) {
				double2 t[4][NP];
				double en[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					en[u] = readlane_f64(e, n + u);
#pragma unroll
					for (int p = 0; p < NP; ++p)
						t[u][p] = (lane + kWave * p < P)
						              ? *reinterpret_cast<const double2 *>(tb + (n + u) * S + 1024 * p)
						              : make_double2(0.0, 0.0);
				}
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int p = 0; p < NP; ++p) {
						acc[p].x = acc[p].x + en[u] * t[u][p].x;
						acc[p].y = acc[p].y + en[u] * t[u][p].y;
					}
			}
			for (; n < cnt; ++n) {
				const double en = readlane_f64(e, n);
#pragma unroll
				for (int p = 0; p < NP; ++p)
					if (lane + kWave * p < P) {
						const double2 t = *reinterpret_cast<const double2 *>(tb + n * S + 1024 * p);
						acc[p].x = acc[p].x + en * t.x;
						acc[p].y = acc[p].y + en * t.y;
					}
			}
			__syncthreads();   // tile is overwritten by the next chunk's DMA
		}
		if (!PRODUCTS) {
			double2 *__restrict__ out2 = reinterpret_cast<double2 *>(a.X_new + (size_t) r * K);
#pragma unroll
			for (int p = 0; p < NP; ++p) {
				const int q = lane + kWave * p;
				if (q < P) out2[q] = acc[p];
			}
		}
	}
}

// Ordered sum of the scaled rows of one extreme row: X_new[r][k] = (...((seed + p_0[k]) + p_1[k]) + ...), the
// serial accumulation order.  One wave per (row, 16-column slice).  The slice is contiguous over the entries
// (128 B each), so ONE LDS-DMA instruction brings a block of 8 consecutive entries (1 KiB, lane-linear) into a
// slot of a 32-slot LDS ring, and the wave keeps 31 blocks -- 248 entries -- in flight ahead of the block it is
// adding: that hides the ~2 us read latency behind the only true critical path, the chain of dependent adds.
// The DMA and its s_waitcnt are inline asm with hand-counted vmcnt (hipcc would otherwise wait vmcnt(0) before
// every LDS read that may alias a pending LDS-DMA); no prefetch registers exist, so nothing can be sunk or
// spilled.  Every lane (piece = lane & 7) walks the 8 entries of a block in order; the eight lane groups hold
// identical sums.
struct OrderedSumArgs {
	int nrows, K, seed, nslices;
	const int *__restrict__ row;          // extreme row ids
	const long long *__restrict__ sbeg;   // first scratch entry of the row
	const int *__restrict__ cnt;          // entries of the row
	const double *__restrict__ scratch;   // [slice][entry][16], each slice padded by 8 entries
	size_t scratch_entries;
	const double *__restrict__ X_old;
	double *__restrict__ X_new;
};

constexpr int kRing = 32;   // LDS ring slots of 1 KiB

__global__ void __launch_bounds__(kWave) ordered_sum_kernel(OrderedSumArgs a)
{
	__shared__ __attribute__((aligned(1024))) char ring[kRing * 1024];
	const int lane = threadIdx.x, K = a.K;
	const unsigned ring_base = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) ring;
	const char *my = ring + 16 * (lane & 7);          // this lane's piece inside an entry
	const int total = a.nrows * a.nslices;
	for (int it = blockIdx.x; it < total; it += gridDim.x) {
		const int li = it / a.nslices, slice = it % a.nslices;
		const int r = a.row[li], cnt = a.cnt[li];
		const int k0 = slice * 16 + 2 * (lane & 7);       // this lane's two columns (all 8 lane groups agree)
		const bool live = k0 < K;                         // K is even: k0 + 1 < K too
		double2 acc = (a.seed && live) ? *reinterpret_cast<const double2 *>(a.X_old + (size_t) r * K + k0)
		                               : make_double2(0.0, 0.0);
		// block b of the row in this slice: 8 entries = 1 KiB at ((slice * entries + sbeg + 8b) * 128) bytes
		const char *src = reinterpret_cast<const char *>(
		                      a.scratch + (((size_t) slice * a.scratch_entries + (size_t) a.sbeg[li]) << 4)) +
		                  16 * lane;
		const int nblk = (cnt + 7) >> 3;

		// every ordinary load above must have landed before the hand-counted region starts
		asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		auto issue = [&](int b) {
			const char *g = src + (size_t) b * 1024;
			const unsigned m0 = ring_base + (unsigned) (b & (kRing - 1)) * 1024u;
			asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0) : "memory");   // m0 is a reserved register: hipcc re-loads it before each of its own uses
		};
		auto add_block = [&](int b, int entries) {
			const char *slot = my + (b & (kRing - 1)) * 1024;
			if (entries == 8) {
				double2 v[8];
#pragma unroll
				for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const double2 *>(slot + 128 * e);
#pragma unroll
				for (int e = 0; e < 8; ++e) {
					acc.x = acc.x + v[e].x;
					acc.y = acc.y + v[e].y;
				}
			} else {
				for (int e = 0; e < entries; ++e) {
					const double2 v = *reinterpret_cast<const double2 *>(slot + 128 * e);
					acc.x = acc.x + v.x;
					acc.y = acc.y + v.y;
				}
			}
		};
		const int ahead = min(nblk, kRing - 1);
		for (int b = 0; b < ahead; ++b) issue(b);
		int b = 0;
		// steady state: kRing-1 blocks are issued beyond b-1, so block b has landed once at most kRing-2 newer DMAs
		// are outstanding; after adding it, its predecessor's slot is refilled (its LDS reads were consumed by the adds)
		for (; b + (kRing - 1) < nblk; ++b) {
			asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
			add_block(b, 8);
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			issue(b + kRing - 1);
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		for (; b < nblk; ++b) add_block(b, min(8, cnt - 8 * b));
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the ring is reused by the next (row, slice)
		if (live && lane < 8) *reinterpret_cast<double2 *>(a.X_new + (size_t) r * K + k0) = acc;
	}
}

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
	const double *__restrict__ L;    // users x K
	const double *__restrict__ R;    // items x K
	const int *__restrict__ csr_ptr; // users + 1
	const int *__restrict__ csr_idx; // item ids, ascending within a user
	int *__restrict__ best;          // users
	const int *__restrict__ ulist;   // optional: only these users (indices into the shard), `users` = its length
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
					Ls[sk + x][srow] = (suid >= 0 && k < K) ? a.L[(size_t) suid * K + k] : 0.0;
					Rs[sk + x][srow] = (ij < a.items && k < K) ? a.R[(size_t) ij * K + k] : 0.0;
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
			if (i < a.users)
				a.best[a.ulist ? a.ulist[i] : i] =
				    run[u].first < 0 ? -1 : (run[u].fnan ? run[u].first : run[u].bi);
		}
	}
}


// ------------------------------------------------------------------------------------------------
// Sweep kernel, row-cooperative form: for launches with FEW rows (ML100k: 943 users / 1682 items), where
// one wave walking a long row alone (737 entries = 47 chunks) is the whole launch time.  A workgroup of
// 8 waves owns one row.  Waves 1..7 ("producers") each take one chunk per round: LDS-DMA gather, phase A
// (sequential-k dots -> e_n) and then SCALE their tile in place, p_n[k] = e_n * y_n[k] (the same rounded
// product the serial loop forms).  Wave 0 (the "accumulator") only walks the finished tiles in entry order
// doing acc[k] = acc[k] + p_n[k]: the serial chain per entry is one dependent add instead of a whole chunk
// pipeline, while the producers already fill the other tile buffer for the next round (double-buffered,
// one barrier per round).  Same arithmetic, same order: results stay bit-identical to the serial reference.
// LDS: [ x row ][ 2 buffers x 7 producers x nch rows x S bytes ].
// ------------------------------------------------------------------------------------------------
constexpr int kCoopWaves = 8;
constexpr int kCoopProducers = kCoopWaves - 1;

template <int KT>
__global__ void __launch_bounds__(kCoopWaves *kWave) sweep_coop_kernel(SweepArgs a)
{
	using G = DmaGeom<KT>;
	constexpr int K = KT, P = G::kPieces, NP = G::kPasses, S = G::kStride;
	extern __shared__ __attribute__((aligned(16))) char lds[];
	double2 *xs = reinterpret_cast<double2 *>(lds);
	const int nch = a.nch;
	const int tile_bytes = nch * S;
	char *tiles = lds + G::kXsBytes;   // tile(buf, p) = tiles + (buf * kCoopProducers + p) * tile_bytes
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned voff = (unsigned) lane * 16u;
	const unsigned long long ybase = (unsigned long long) a.Y_old;
	const int per_round = kCoopProducers * nch;

	for (int it = blockIdx.x; it < a.nrows; it += gridDim.x) {
		const int r = a.rowlist ? a.rowlist[it] : it;
		const int beg = a.ptr[r], end = a.ptr[r + 1];
		const double2 *__restrict__ xrow2 = reinterpret_cast<const double2 *>(a.X_old + (size_t) r * K);
		double2 acc[NP];
#pragma unroll
		for (int p = 0; p < NP; ++p) acc[p] = make_double2(0.0, 0.0);
		if (wave == 0) {
#pragma unroll
			for (int p = 0; p < NP; ++p) {
				const int q = lane + kWave * p;
				if (q < P) {
					const double2 v = xrow2[q];
					xs[q] = v;
					if (a.seed) acc[p] = v;
				}
			}
		}
		__syncthreads();
		const int rounds = (end - beg + per_round - 1) / per_round;
		for (int round = 0; round <= rounds; ++round) {
			if (wave > 0) {
				// ---- producer: chunk (round, wave-1) -> buffer round&1
				const int c = beg + round * per_round + (wave - 1) * nch;
				const int cnt = round < rounds ? max(0, min(nch, end - c)) : 0;
				if (cnt > 0) {
					char *tile = tiles + ((round & 1) * kCoopProducers + (wave - 1)) * tile_bytes;
					int my_idx = 0;
					double my_val = 0.0;
					if (lane < cnt) {
						my_idx = a.idx[c + lane];
						my_val = a.val[c + lane];
					}
					for (int n = 0; n < cnt; ++n) {
						const int j = __builtin_amdgcn_readlane(my_idx, n);
						unsigned long long base = ybase + (unsigned long long) (unsigned) j * (unsigned long long) (K * 8);
						asm volatile("" : "+s"(base));
#pragma unroll
						for (int p = 0; p < NP; ++p) {
							const char *src = reinterpret_cast<const char *>(base) + voff + 1024u * p;
							if (lane + kWave * p < P)
								__builtin_amdgcn_global_load_lds((mf_gvoid *) src,
								                                 (mf_lvoid *) (tile + n * S + 1024 * p), 16, 0, 0);
						}
					}
					__builtin_amdgcn_s_waitcnt(0);          // vmcnt(0): the DMA has landed (single wave owns the tile)
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					double e;
					{
						const double2 *t2 = reinterpret_cast<const double2 *>(tile + (lane < nch ? lane : 0) * S);   // lanes beyond the tile re-read row 0
						double dot = 0.0;
#pragma unroll
						for (int q = 0; q < P; ++q) {
							const double2 t = t2[q];
							const double2 x = xs[q];
							dot = dot + x.x * t.x;
							dot = dot + x.y * t.y;
						}
						e = a.c2 * (my_val - dot);
					}
					// scale in place: p_n[k] = e_n * y_n[k]
					char *tb = tile + voff;
					for (int n = 0; n < cnt; ++n) {
						const double en = readlane_f64(e, n);
#pragma unroll
						for (int p = 0; p < NP; ++p)
							if (lane + kWave * p < P) {
								double2 *slot = reinterpret_cast<double2 *>(tb + n * S + 1024 * p);
								double2 t = *slot;
								t.x = en * t.x;
								t.y = en * t.y;
								*slot = t;
							}
					}
				}
			} else if (round > 0) {
				// ---- accumulator: the tiles of round-1, producers in order, entries in order
				const int base_c = beg + (round - 1) * per_round;
				for (int pw = 0; pw < kCoopProducers; ++pw) {
					const int cnt = max(0, min(nch, end - (base_c + pw * nch)));
					const char *tb = tiles + (((round - 1) & 1) * kCoopProducers + pw) * tile_bytes + voff;
					int n = 0;
					for (; n + 8 <= cnt; n += 8) {
						double2 t[8][NP];
#pragma unroll
						for (int u = 0; u < 8; ++u)
#pragma unroll
							for (int p = 0; p < NP; ++p)
								t[u][p] = (lane + kWave * p < P)
								              ? *reinterpret_cast<const double2 *>(tb + (n + u) * S + 1024 * p)
								              : make_double2(0.0, 0.0);
#pragma unroll
						for (int u = 0; u < 8; ++u)
#pragma unroll
							for (int p = 0; p < NP; ++p) {
								acc[p].x = acc[p].x + t[u][p].x;
								acc[p].y = acc[p].y + t[u][p].y;
							}
					}
					for (; n < cnt; ++n)
#pragma unroll
						for (int p = 0; p < NP; ++p)
							if (lane + kWave * p < P) {
								const double2 t = *reinterpret_cast<const double2 *>(tb + n * S + 1024 * p);
								acc[p].x = acc[p].x + t.x;
								acc[p].y = acc[p].y + t.y;
							}
				}
			}
			__syncthreads();
		}
		if (wave == 0) {
			double2 *__restrict__ out2 = reinterpret_cast<double2 *>(a.X_new + (size_t) r * K);
#pragma unroll
			for (int p = 0; p < NP; ++p) {
				const int q = lane + kWave * p;
				if (q < P) out2[q] = acc[p];
			}
		}
		__syncthreads();   // xs is rewritten for the next row
	}
}

// In-process all-reduce of the item factor over peer-mapped buffers (the MPI_Iallreduce of
// matFact-mpi.c:208 for the single-process multi-GPU path).  Shard g owns slice g of the buffer: it reads
// that slice from every shard's buffer (xGMI peer loads), sums in shard order 0..N-1 -- a fixed order, so the
// result is reproducible -- and writes the sum back into every shard's buffer (peer stores).  Slices are
// disjoint, so the N kernels (one per device) never touch the same element.
constexpr int kMaxShards = 16;
struct PeerReduceArgs {
	double *buf[kMaxShards];
	int nshards;
	size_t begin, end;   // element range of this shard's slice (both even)
};

__global__ void __launch_bounds__(256) peer_allreduce_kernel(PeerReduceArgs a)
{
	const size_t stride = (size_t) gridDim.x * 256 * 2;
	for (size_t e = a.begin + ((size_t) blockIdx.x * 256 + threadIdx.x) * 2; e < a.end; e += stride) {
		if (e + 1 < a.end) {
			double2 v = *reinterpret_cast<const double2 *>(a.buf[0] + e);
			for (int h = 1; h < a.nshards; ++h) {
				const double2 w = *reinterpret_cast<const double2 *>(a.buf[h] + e);
				v.x = v.x + w.x;
				v.y = v.y + w.y;
			}
			for (int h = 0; h < a.nshards; ++h) *reinterpret_cast<double2 *>(a.buf[h] + e) = v;
		} else {
			double v = a.buf[0][e];
			for (int h = 1; h < a.nshards; ++h) v = v + a.buf[h][e];
			for (int h = 0; h < a.nshards; ++h) a.buf[h][e] = v;
		}
	}
}

// Dense B = L R^T (mat2d_prod, mat2d.c:100-113) for the debug dump of small instances: one thread per
// (i, j), sequential k from 0.0, separate multiply and add -- every element equals the reference's B[i][j].
__global__ void __launch_bounds__(256) predict_kernel(const double *__restrict__ L, const double *__restrict__ R,
                                                      int users, int items, int K, double *__restrict__ B)
{
	const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (t >= (size_t) users * items) return;
	const double *l = L + (t / items) * K, *r = R + (t % items) * K;
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
// Tile: 256 threads = 4 waves (2 x 2) own 128 users x 64 items per step; each wave holds 4 x 2
// accumulator tiles of 16 x 16 (64 VGPRs); L and R k-chunks of 16 go through LDS stored k-major with a
// leading dimension that puts the two 16-lane halves of a ds_read_b64 group on disjoint banks.
// MFMA operand maps (f64 16x16x4): A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15],
// D lane l reg r = D[(l>>4) + 4r][l&15].
// ------------------------------------------------------------------------------------------------
struct RecMfmaArgs {
	int users, items, K;
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
};

__global__ void __launch_bounds__(kWave) row_norm_kernel(const double *__restrict__ X, int rows, int K,
                                                          double *__restrict__ norm,
                                                          unsigned long long *__restrict__ max_bits)
{
	const int r = blockIdx.x * kWave + threadIdx.x;
	double s = 0.0;
	if (r < rows)
		for (int k = 0; k < K; ++k) {
			const double v = X[(size_t) r * K + k];
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

constexpr int kMU = 128, kMI = 128, kMKC = 32;
// LDS image of a chunk: [k-pair][row] of double2 {x[row][2p], x[row][2p+1]}, 130 rows per k-pair.
//   fragment read (ds_read_b64): lanes 0..31 = 16 rows x (k, k+1) of one pair -> 256 contiguous bytes;
//   staging store (ds_write_b128): an 8-lane group = 2 rows x 4 k-pairs, pair stride 130*16 B = 8 banks mod 32.
// Both are bank-conflict-free (the first version's [k][row] image conflicted 4-way on the stores).
constexpr int kMLD2 = 130;
constexpr int kMThreads = 512;

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

// 512 threads = 8 waves as 4 (user quarters of 32) x 2 (item halves of 64): two waves per SIMD, so one
// wave's staging, LDS traffic and arg-max bookkeeping run under the other's matrix instructions.
template <bool VEC>   // VEC: K even -> rows are 16-B aligned, 16-byte global loads
__global__ void __launch_bounds__(kMThreads) recommend_mfma_kernel(RecMfmaArgs a)
{
	__shared__ double2 As[2][kMKC / 2][kMLD2];   // double-buffered: one barrier per chunk
	__shared__ double2 Bs[2][kMKC / 2][kMLD2];
	__shared__ unsigned long long maskw[kMU][2];
	__shared__ double red_b1[kMU][2], red_b2[kMU][2];
	__shared__ int red_i1[kMU][2], red_bad[kMU][2];

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wr = wave >> 1, wc = wave & 1;
	const int lr = lane & 15, lq = lane >> 4;
	const int i0 = blockIdx.x * kMU;
	const int K = a.K;
	const double ninf = -__builtin_inf();

	// mask walker: threads 0..127, one user each
	int cur = 0, cend = 0, nextcol = INT32_MAX;
	if (tid < kMU && i0 + tid < a.users) {
		cur = a.csr_ptr[i0 + tid];
		cend = a.csr_ptr[i0 + tid + 1];
		nextcol = cur < cend ? a.csr_idx[cur] : INT32_MAX;
	}

	// running top-2 of the 8 rows this lane sees: row(tu, r) = 32*wr + 16*tu + lq + 4*r
	double b1[8], b2[8];
	int i1[8];
	unsigned bad = 0;
#pragma unroll
	for (int x = 0; x < 8; ++x) {
		b1[x] = ninf;
		b2[x] = ninf;
		i1[x] = -1;
	}

	// staging roles: A and B chunks are 128 rows x 32 k; thread -> row tid/4, k-pairs 4m + (tid%4), m = 0..3
	constexpr int SP = kMKC / 8;
	const int srow = tid >> 2, sq = tid & 3;
	const bool a_ok = i0 + srow < a.users;
	const double *__restrict__ aptr = a.L + (size_t) (a_ok ? i0 + srow : 0) * K;
	double2 av[SP], bv[SP];

	// global -> registers for chunk (tile jt, k offset kc); zero outside the matrices
	auto fetch = [&](int jt, int kc) {
		const bool b_ok = jt + srow < a.items;
		const double *__restrict__ bptr = a.R + (size_t) (b_ok ? jt + srow : 0) * K;
#pragma unroll
		for (int m = 0; m < SP; ++m) {
			const int k = kc + 8 * m + 2 * sq;
			if (VEC) {
				av[m] = (a_ok && k < K) ? *reinterpret_cast<const double2 *>(aptr + k) : make_double2(0.0, 0.0);
				bv[m] = (b_ok && k < K) ? *reinterpret_cast<const double2 *>(bptr + k) : make_double2(0.0, 0.0);
			} else {
				av[m].x = (a_ok && k < K) ? aptr[k] : 0.0;
				av[m].y = (a_ok && k + 1 < K) ? aptr[k + 1] : 0.0;
				bv[m].x = (b_ok && k < K) ? bptr[k] : 0.0;
				bv[m].y = (b_ok && k + 1 < K) ? bptr[k + 1] : 0.0;
			}
		}
	};
	auto stage = [&](int buf) {
#pragma unroll
		for (int m = 0; m < SP; ++m) {
			As[buf][4 * m + sq][srow] = av[m];
			Bs[buf][4 * m + sq][srow] = bv[m];
		}
	};

	int buf = 0;
	fetch(0, 0);
	stage(0);
	__syncthreads();
	for (int j0 = 0; j0 < a.items; j0 += kMI) {
		mf_d4 acc[2][4];
#pragma unroll
		for (int tu = 0; tu < 2; ++tu)
#pragma unroll
			for (int ti = 0; ti < 4; ++ti) acc[tu][ti] = mf_d4{0.0, 0.0, 0.0, 0.0};

		for (int kc = 0; kc < K; kc += kMKC) {
			// next chunk (of this tile, or the first of the next tile): global loads fly behind the MFMAs
			const bool more = kc + kMKC < K || j0 + kMI < a.items;
			if (kc + kMKC < K)
				fetch(j0, kc + kMKC);
			else if (j0 + kMI < a.items)
				fetch(j0 + kMI, 0);
			const double *Ab = reinterpret_cast<const double *>(&As[buf][0][0]);
			const double *Bb = reinterpret_cast<const double *>(&Bs[buf][0][0]);
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
			if (kc + kMKC <= K) {
#pragma unroll
				for (int ks = 0; ks < kMKC / 4; ++ks) kstep(ks);
			} else {   // last chunk: skip the zero padding beyond K
				const int ksteps = (K - kc + 3) >> 2;
				for (int ks = 0; ks < ksteps; ++ks) kstep(ks);
			}
			// registers -> the OTHER buffer (last read one chunk ago; every wave passed a barrier since)
			if (more) stage(buf ^ 1);
			__syncthreads();
			buf ^= 1;
		}

		// rated-item mask of this tile: bit jj of word w = item j0 + 64*w + jj is rated or beyond the last item
		if (tid < kMU) {
			unsigned long long m0 = 0, m1 = 0;
			while (nextcol < j0 + kMI) {
				const int o = nextcol - j0;
				if (o >= 64)
					m1 |= 1ull << (o - 64);
				else if (o >= 0)
					m0 |= 1ull << o;
				++cur;
				nextcol = cur < cend ? a.csr_idx[cur] : INT32_MAX;
			}
			const int left = a.items - j0;   // > 0
			if (left < 64) {
				m0 |= ~0ull << left;
				m1 = ~0ull;
			} else if (left < 128) {
				m1 |= ~0ull << (left - 64);
			}
			maskw[tid][0] = m0;
			maskw[tid][1] = m1;
		}
		__syncthreads();
#pragma unroll
		for (int tu = 0; tu < 2; ++tu)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int x = tu * 4 + r;
				const unsigned long long m = maskw[32 * wr + 16 * tu + lq + 4 * r][wc] >> lr;
				// cheap reject: after the first tiles almost no score beats the row's runner-up
				bool any = false;
#pragma unroll
				for (int ti = 0; ti < 4; ++ti) {
					const double v = acc[tu][ti][r];
					any |= !((m >> (16 * ti)) & 1ull) && !(v <= b2[x] && v >= -1.7976931348623157e308);   // v > b2, NaN, +-inf
				}
				if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
					for (int ti = 0; ti < 4; ++ti) {
						// branch-free insert: b1 starts at -inf, so "first candidate" needs no special case
						const double v = acc[tu][ti][r];
						const int j = j0 + 64 * wc + 16 * ti + lr;
						const bool open = !((m >> (16 * ti)) & 1ull);
						const bool fin = fabs(v) <= 1.7976931348623157e308;
						bad |= (unsigned) (open && !fin) << x;
						const bool use = open && fin;
						const bool gt1 = use && v > b1[x];
						const bool gt2 = use && !gt1 && v > b2[x];
						b2[x] = gt1 ? b1[x] : (gt2 ? v : b2[x]);
						b1[x] = gt1 ? v : b1[x];
						i1[x] = gt1 ? j : i1[x];
					}
				}
			}
		__syncthreads();   // maskw is rewritten by the next tile
	}

	// merge the 16 lanes (lr) that share a row, then the two item halves (wc), then decide
#pragma unroll
	for (int x = 0; x < 8; ++x) {
		Top2 t{b1[x], b2[x], i1[x]};
		int bd = (bad >> x) & 1;
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
			const int row = 32 * wr + 16 * (x >> 2) + lq + 4 * (x & 3);
			red_b1[row][wc] = t.b1;
			red_b2[row][wc] = t.b2;
			red_i1[row][wc] = t.i1;
			red_bad[row][wc] = bd;
		}
	}
	__syncthreads();
	if (tid < kMU && i0 + tid < a.users) {
		Top2 t{red_b1[tid][0], red_b2[tid][0], red_i1[tid][0]};
		const Top2 o{red_b1[tid][1], red_b2[tid][1], red_i1[tid][1]};
		top2_merge(t, o);
		const int bd = red_bad[tid][0] | red_bad[tid][1];
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

}  // namespace mf
