// mf_sweep.hip.h -- gfx950 (CDNA4, wave64) kernels of the matrix-factorisation hot path.
//
// Both kernels are "owner computes": one wavefront owns one row of the factor it updates, so no atomics
// are needed and every floating-point sum is formed in exactly the order the serial reference forms it
// (matFact.c:41-53): the factor matrices come out BIT-IDENTICAL to matFact.c, not merely close.
// Build with -ffp-contract=off: the reference multiplies and adds separately (no FMA).
// This file: the sweep kernels (single-wave LDS-DMA form, products form, ordered sum, register-staged form)
// and the row-cooperative form.
#pragma once
#include "mf_common.hip.h"

namespace mf {

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
	int ldx, ldy; // row pitch of X and of Y in doubles (>= K: rows of 8K bytes padded to whole 128-byte lines, mf_plan)
	int seed;     // 1: accumulate onto X_old, 0: onto zero
	int prio_len; // rows of at least this many entries run at raised wave priority (0: none)
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
	// errors mode (mf_stream.hip.h): e_n of every entry of the segment goes to the 16-byte records {idx, pad, err} of
	// both sides -- record n of err_a (this side's entry order), record map[n] of err_b; nothing is accumulated
	double *__restrict__ err_a;
	double *__restrict__ err_b;
	const int *__restrict__ map;
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
		const double *__restrict__ xrow = a.X_old + (size_t) r * a.ldx;

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
				const double *__restrict__ yrow = a.Y_old + (size_t) j * a.ldy;
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
				a.X_new[(size_t) r * a.ldx + k] = acc[kk];
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

// Extreme-row scratch geometry: the scaled rows are stored [slice of kSliceCols columns][entry][kSliceCols doubles];
// one ordered-sum wave owns one (row, slice).  Narrower slices = more waves per extreme row, each streaming fewer
// bytes per entry through its ring of in-flight blocks: the chain of the longest row is bound by what ONE wave can
// keep in flight, so 8-column slices halve it.
constexpr int kSliceCols = 8;
constexpr int kSlicePieces = kSliceCols / 2;          // 16-B pieces per (slice, entry)
constexpr int kBlockEntries = kWave / kSlicePieces;   // entries per 1-KiB DMA block
constexpr int kSliceShift = kSliceCols == 8 ? 3 : 4;  // log2(kSliceCols)
constexpr int kPieceShift = kSliceShift - 1;
static_assert(kSliceCols == 8 || kSliceCols == 16, "slice width");

// Phase A / phase B with the LDS reads kept IN FLIGHT (PF steps / PB entries ahead) instead of the two steps per round
// trip hipcc schedules by itself (it aims at 64 registers): one wave walking a long row alone -- the tail of every launch of
// a few thousand rows -- is bound by the LDS latency of these reads, 25 round trips per chunk at K=100, not by the chain of
// dependent adds.  Same operations in the same order: same bits.  Empty asm statements with a memory clobber keep the reads where they
// are written (a sched_barrier does not: the loads are hoisted above it before the machine scheduler runs).
template <int Q, int PF>
__device__ __forceinline__ double phase_a_dot_pipelined(const double2 *t2, const double2 *xs)
{
	double2 t[PF], x[PF];
#pragma unroll
	for (int i = 0; i < PF; ++i)
		if (i < Q) {
			t[i] = t2[i];
			x[i] = xs[i];
		}
	double dot = 0.0;
#pragma unroll
	for (int q = 0; q < Q; ++q) {
		const double2 tt = t[q % PF], xx = x[q % PF];
		if (q + PF < Q) {
			t[q % PF] = t2[q + PF];
			x[q % PF] = xs[q + PF];
		}
		asm volatile("" ::: "memory");   // pins the issue order: the reads above go out before the arithmetic below
		dot = dot + xx.x * tt.x;
		dot = dot + xx.y * tt.y;
	}
	return dot;
}

// KT > 0: K is a compile-time constant (phase A fully unrolled).  KT == 0: any even K up to 128*NPASS at run
// time (phase A unrolled by four) -- same data movement, so an unusual K does not fall back to the
// register-staged kernel.
// PRODUCTS = true: the "extreme row" form -- one wave per SEGMENT of a very long row; instead of accumulating, the
// scaled rows p_n[k] = e_n * y_n[k] (the rounded product the serial loop adds) are stored to a scratch buffer in
// entry order, and ordered_sum_kernel adds them up in that order afterwards.  Thousands of segments run in
// parallel, so a row rated by every user costs a chip-wide pass plus one serial chain of adds.
// MODE: 0 accumulate (the sweep), 1 products (extreme rows), 2 errors (first half of the errors + streams iteration)
constexpr int kSweepAccumulate = 0, kSweepProducts = 1, kSweepErrors = 2;
#ifdef MF_STAMPS
// diagnostic build only (tools/stamps.py): shader-clock totals of the phases of the rows of at least 1024 entries --
// [0] rows, [1] chunks, [2] gather issue, [3] landing wait, [4] phase A, [5] phase B, [6] whole row
__device__ unsigned long long mf_stamp_buf[8];
#define MF_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define MF_STAMP(var)
#endif

template <int KT, int NPASS, int MODE = kSweepAccumulate, int PF = 0>
__global__ void __launch_bounds__(kWave) sweep_dma_kernel(SweepArgs a)
{
	constexpr bool PRODUCTS = MODE == kSweepProducts, ERRORS = MODE == kSweepErrors, SEGMENTS = MODE != kSweepAccumulate;
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
	const size_t ybytes = (size_t) a.ldy * 8;   // bytes between rows of Y

	for (int it = blockIdx.x; it < a.nrows; it += gridDim.x) {
		const int r = SEGMENTS ? a.seg_row[it] : (a.rowlist ? a.rowlist[it] : it);
		const int beg = SEGMENTS ? a.seg_beg[it] : a.ptr[r];
		const int end = SEGMENTS ? a.seg_end[it] : a.ptr[r + 1];
		const double2 *__restrict__ xrow2 = reinterpret_cast<const double2 *>(a.X_old + (size_t) r * a.ldx);
		// A wave walking a long row is bound by its own instruction stream and shares its SIMD's issue slots with the
		// waves of short rows; it is also what the launch ends on.  Rows of at least prio_len entries run at raised
		// priority (issue arbitration is by priority, then age), the others at the default.
		if (a.prio_len > 0) {
			if (end - beg >= a.prio_len)
				__builtin_amdgcn_s_setprio(3);
			else
				__builtin_amdgcn_s_setprio(0);
		}

#ifdef MF_STAMPS
		unsigned long long st_issue = 0, st_wait = 0, st_a = 0, st_b = 0, st_chunks = 0;
#endif
		MF_STAMP(t_row0);
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
		int nx_idx = 0, nx_map = 0;
		double nx_val = 0.0;
		if (beg + lane < min(end, beg + nch)) {
			nx_idx = a.idx[beg + lane];
			nx_val = a.val[beg + lane];
			if (ERRORS) nx_map = a.map[beg + lane];
		}
		for (int c = beg; c < end; c += nch) {
			const int cnt = min(nch, end - c);
			const int my_idx = nx_idx;
			const int my_map = nx_map;
			const double my_val = nx_val;
			if (c + nch + lane < min(end, c + 2 * nch)) {
				nx_idx = a.idx[c + nch + lane];
				nx_val = a.val[c + nch + lane];
				if (ERRORS) nx_map = a.map[c + nch + lane];
			}
			// ---- stage.  Short rows (K <= 62) are gathered SEVERAL per instruction: lane -> (row lane / PS, piece
			// lane % PS) with PS = the tile row stride in pieces (P, or P + 1 when P is even: that lane is the padding
			// and stays off), 12 rows at K=10, 5 at K=20 -- instead of one instruction with five active lanes per row.
			MF_STAMP(t_s0);
			constexpr bool kMultiRow = KT > 0 && ((KT / 2) | 1) <= 32;
			if constexpr (kMultiRow) {
				constexpr int PP = KT / 2, PS = PP | 1, RPI = kWave / PS;   // pieces, stride in pieces, rows per instruction
				const int rr = lane / PS, piece = lane - rr * PS;
				for (int n0 = 0; n0 < cnt; n0 += RPI) {
					const int n = n0 + rr;
					const int j = __shfl(my_idx, n < cnt ? n : 0);
					const char *src = reinterpret_cast<const char *>(ybase) + (size_t) (unsigned) j * ybytes + 16 * piece;
					if (rr < RPI && piece < PP && n < cnt)
						__builtin_amdgcn_global_load_lds((mf_gvoid *) src, (mf_lvoid *) (tile + n0 * S), 16, 0, 0);
				}
			} else if constexpr (PF > 0 && (NP == 1 || KT == 256)) {
				// Lean issue (a wave walking a long row alone is bound by its own instruction stream, ~17 instructions per
				// gathered row in the loop below): every lane forms the address of ITS entry's row once per chunk (one
				// 64-bit multiply-add per chunk instead of four scalar multiplies per row); per row two v_readlane give the
				// row base as a scalar pair and the transfer takes it as its scalar address with the lane's 16-byte offset
				// as the vector part.  The asm opens with s_nop 4: an SGPR written by v_readlane needs five wait states
				// before a VMEM instruction reads it as its address, and hipcc pads nothing inside an asm statement.
				// K = 256 (two 1-KiB instructions per row, all lanes active in both): the second one is the first with
				// offset:1024 -- the instruction offset of an LDS-DMA load moves the LDS side as well as the global one
				// (tools/micro/lds_dma_offset.hip), so neither M0 nor the base is touched between the two.
				const unsigned long long rowaddr = ybase + (unsigned long long) (unsigned) my_idx * (unsigned long long) ybytes;
				const int alo = (int) (unsigned) rowaddr, ahi = (int) (unsigned) (rowaddr >> 32);
				const unsigned tile_lds = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) tile;
				int n = 0;
				for (; n + 4 <= cnt; n += 4) {
					unsigned long long b[4];
#pragma unroll
					for (int u = 0; u < 4; ++u)
						b[u] = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(ahi, n + u) << 32) |
						       (unsigned long long) (unsigned) __builtin_amdgcn_readlane(alo, n + u);
					if (lane < P) {
#pragma unroll
						for (int u = 0; u < 4; ++u)
							if constexpr (NP == 2)
								asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0\n\t"
								             "global_load_lds_dwordx4 %2, %0 offset:1024"
								             :
								             : "s"(b[u]), "s"(tile_lds + (unsigned) ((n + u) * S)), "v"(voff)
								             : "memory");
							else
								asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0"
								             :
								             : "s"(b[u]), "s"(tile_lds + (unsigned) ((n + u) * S)), "v"(voff)
								             : "memory");
					}
				}
				for (; n < cnt; ++n) {
					const unsigned long long b = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(ahi, n) << 32) |
					                             (unsigned long long) (unsigned) __builtin_amdgcn_readlane(alo, n);
					if (lane < P) {
						if constexpr (NP == 2)
							asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0\n\t"
							             "global_load_lds_dwordx4 %2, %0 offset:1024"
							             :
							             : "s"(b), "s"(tile_lds + (unsigned) (n * S)), "v"(voff)
							             : "memory");
						else
							asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0"
							             :
							             : "s"(b), "s"(tile_lds + (unsigned) (n * S)), "v"(voff)
							             : "memory");
					}
				}
				// hipcc knows nothing of these transfers: the barrier below would not wait for them
				MF_STAMP(t_s1);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MF_STAMPS
				st_issue += t_s1 - t_s0;
				st_wait += __builtin_amdgcn_s_memtime() - t_s1;
#endif
			} else
			for (int n = 0; n < cnt; ++n) {
				const int j = __builtin_amdgcn_readlane(my_idx, n);
				unsigned long long base = ybase + (unsigned long long) (unsigned) j * (unsigned long long) ybytes;
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
			MF_STAMP(t_a0);
			// ---- phase A
			double e;
			{
				const double2 *t2 = reinterpret_cast<const double2 *>(tile + (lane < nch ? lane : 0) * S);   // lanes beyond the tile re-read row 0
				double dot = 0.0;
				if constexpr (KT > 0 && PF > 0) {
					dot = phase_a_dot_pipelined<KT / 2, PF>(t2, xs);
				} else if (KT > 0) {
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
#ifdef MF_STAMPS
			asm volatile("" : "+v"(e));
#endif
			MF_STAMP(t_a1);
			if (ERRORS) {
				if (lane < cnt) {
					a.err_a[2 * (size_t) (c + lane) + 1] = e;
					a.err_b[2 * (size_t) my_map + 1] = e;
				}
				__syncthreads();   // tile is overwritten by the next chunk's DMA
				continue;
			}
			if (PRODUCTS) {
				// scratch layout: [k-slice][entry][kSliceCols doubles] -- a slice is contiguous over the entries, so
				// ordered_sum_kernel streams it linearly
				const char *tb = tile + voff;
				const size_t pos = (size_t) (a.seg_out[it] + (c - beg));
				for (int n = 0; n < cnt; ++n) {
					const double en = readlane_f64(e, n);
#pragma unroll
					for (int p = 0; p < NP; ++p) {
						const int q = lane + kWave * p;   // 16-B piece: slice q / kSlicePieces, position q % kSlicePieces
						if (q < P) {
							double2 t = *reinterpret_cast<const double2 *>(tb + n * S + 1024 * p);
							t.x = en * t.x;
							t.y = en * t.y;
							*reinterpret_cast<double2 *>(a.scratch +
							                             (((size_t) (q >> kPieceShift) * a.scratch_entries + pos + n) << kSliceShift) +
							                             2 * (q & (kSlicePieces - 1))) = t;
						}
					}
				}
				__syncthreads();
				continue;
			}
			// ---- phase B
			const char *tb = tile + voff;
			int n = 0;
			if constexpr (PF > 0 && NP == 1) {   // sixteen entries' reads in flight at once
				for (; n + 16 <= cnt; n += 16) {
					double2 t[16];
#pragma unroll
					for (int u = 0; u < 16; ++u)
						t[u] = (lane < P) ? *reinterpret_cast<const double2 *>(tb + (n + u) * S) : make_double2(0.0, 0.0);
					asm volatile("" ::: "memory");   // pins the issue order: the reads above go out before the arithmetic below
#pragma unroll
					for (int u = 0; u < 16; ++u) {
						const double en = readlane_f64(e, n + u);
						acc[0].x = acc[0].x + en * t[u].x;
						acc[0].y = acc[0].y + en * t[u].y;
					}
				}
			}
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
#ifdef MF_STAMPS
			st_a += t_a1 - t_a0;
			st_b += __builtin_amdgcn_s_memtime() - t_a1;
			++st_chunks;
#endif
		}
#ifdef MF_STAMPS
		if (end - beg >= 1024 && lane == 0 && PF > 0) {
			atomicAdd(&mf_stamp_buf[0], 1ull);
			atomicAdd(&mf_stamp_buf[1], st_chunks);
			atomicAdd(&mf_stamp_buf[2], st_issue);
			atomicAdd(&mf_stamp_buf[3], st_wait);
			atomicAdd(&mf_stamp_buf[4], st_a);
			atomicAdd(&mf_stamp_buf[5], st_b);
			atomicAdd(&mf_stamp_buf[6], __builtin_amdgcn_s_memtime() - t_row0);
		}
#endif
		if (!SEGMENTS) {
			double2 *__restrict__ out2 = reinterpret_cast<double2 *>(a.X_new + (size_t) r * a.ldx);
#pragma unroll
			for (int p = 0; p < NP; ++p) {
				const int q = lane + kWave * p;
				if (q < P) out2[q] = acc[p];
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Sweep kernel, intra-wave double-buffered form: for launches that cannot fill the chip (a few thousand rows, the
// cfg3 shapes; the rows left beside the extreme-row path).  sweep_dma_kernel alternates "gather a chunk" and "compute on
// it" and relies on the OTHER resident workgroups for bytes in flight; with ~15 rows per CU, and one or two long rows
// left per CU at the end, nothing hides the gather.  Here the wave itself keeps the gather of chunk c+1 in flight under
// phases A and B of chunk c: two tiles, the LDS-DMA from inline asm (so hipcc does not drain vmcnt in front of every LDS
// read that might alias a transfer it knows of), one `s_waitcnt vmcnt(0)` per chunk at the top of the loop.  No counted
// vmcnt is needed: in issue order the queue holds [DMA(c) | idx,val(c+1)] when iteration c starts and everything in it
// must have arrived before chunk c is computed and chunk c+1 is issued; what stays in flight under the phases is
// [DMA(c+1) | idx,val(c+2)], which nothing touches before the next vmcnt(0) (hipcc's own waits for idx/val count only its
// own loads: with foreign transfers behind them they can only wait longer, never too little).
// Same arithmetic in the same order as sweep_dma_kernel: same bits.  Twice the LDS per workgroup, so only chosen where
// occupancy is not what hides latency (choose_sweep / launch_sweep: few rows per CU).  Accumulate mode only.
// ------------------------------------------------------------------------------------------------
template <int KT, int NPASS>
__global__ void __launch_bounds__(kWave) sweep_db_kernel(SweepArgs a)
{
	const int K = KT > 0 ? KT : a.K;
	const int P = K >> 1;
	constexpr int NP = NPASS;
	const int S = 16 * (P | 1);
	const int xs_bytes = ((K * 8 + 255) / 256) * 256;
	extern __shared__ __attribute__((aligned(16))) char lds[];
	double2 *xs = reinterpret_cast<double2 *>(lds);
	const int nch = a.nch;
	const int tile_bytes = nch * S;
	char *tile0 = lds + xs_bytes;
	const unsigned tile0_lds = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) tile0;
	const int lane = threadIdx.x;
	const unsigned voff = (unsigned) lane * 16u;
	const unsigned long long ybase = (unsigned long long) a.Y_old;
	const size_t ybytes = (size_t) a.ldy * 8;

	// gather of one chunk into tile `buf`: the rows of lanes 0..cnt-1 of `idx`
	auto stage = [&](int idx, int cnt, int buf) {
		const unsigned tb = tile0_lds + (unsigned) (buf * tile_bytes);
		constexpr bool kMultiRow = KT > 0 && ((KT / 2) | 1) <= 32;
		if constexpr (kMultiRow) {
			constexpr int PP = KT / 2, PS = PP | 1, RPI = kWave / PS;
			const int rr = lane / PS, piece = lane - rr * PS;
			for (int n0 = 0; n0 < cnt; n0 += RPI) {
				const int n = n0 + rr;
				const int j = __shfl(idx, n < cnt ? n : 0);
				const char *src = reinterpret_cast<const char *>(ybase) + (size_t) (unsigned) j * ybytes + 16 * piece;
				const unsigned m0 = __builtin_amdgcn_readfirstlane(tb + (unsigned) (n0 * S));
				if (rr < RPI && piece < PP && n < cnt)
					asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0) : "memory");
			}
		} else
			for (int n = 0; n < cnt; ++n) {
				const int j = __builtin_amdgcn_readlane(idx, n);
				unsigned long long base = ybase + (unsigned long long) (unsigned) j * (unsigned long long) ybytes;
				asm volatile("" : "+s"(base));
#pragma unroll
				for (int p = 0; p < NP; ++p) {
					const char *src = reinterpret_cast<const char *>(base) + voff + 1024u * p;
					const unsigned m0 = __builtin_amdgcn_readfirstlane(tb + (unsigned) (n * S + 1024 * p));
					if (lane + kWave * p < P)
						asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0) : "memory");
				}
			}
	};

	for (int it = blockIdx.x; it < a.nrows; it += gridDim.x) {
		const int r = a.rowlist ? a.rowlist[it] : it;
		const int beg = a.ptr[r], end = a.ptr[r + 1];
		const double2 *__restrict__ xrow2 = reinterpret_cast<const double2 *>(a.X_old + (size_t) r * a.ldx);
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
		// chunk c's (idx, val) in cur_*, chunk c+1's in nx_*; the gather of chunk c is issued one iteration ahead
		int cur_idx = 0, nx_idx = 0;
		double cur_val = 0.0, nx_val = 0.0;
		if (beg + lane < min(end, beg + nch)) {
			cur_idx = a.idx[beg + lane];
			cur_val = a.val[beg + lane];
		}
		if (beg + nch + lane < min(end, beg + 2 * nch)) {
			nx_idx = a.idx[beg + nch + lane];
			nx_val = a.val[beg + nch + lane];
		}
		// (the asm statements on idx/val below pin hipcc's own vmcnt for these loads HERE, in front of the gather loop:
		// left alone it re-waits for them inside the loop, behind every transfer it knows nothing of -- one row at a time)
		asm volatile("" : "+v"(cur_idx));
		if (beg < end) stage(cur_idx, min(nch, end - beg), 0);
		int buf = 0;
		for (int c = beg; c < end; c += nch, buf ^= 1) {
			const int cnt = min(nch, end - c);
			const double my_val = cur_val;
			// everything in flight has arrived: the tile of chunk c, and (idx, val) of chunk c+1
			asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx_idx), "+v"(nx_val), "+v"(cur_val)::"memory");
			// the other tile was last read by phase B of chunk c-1 (its reads are complete: their results were consumed)
			if (c + nch < end) stage(nx_idx, min(nch, end - (c + nch)), buf ^ 1);
			cur_idx = nx_idx;
			cur_val = nx_val;
			nx_idx = 0;
			nx_val = 0.0;
			if (c + 2 * nch + lane < min(end, c + 3 * nch)) {
				nx_idx = a.idx[c + 2 * nch + lane];
				nx_val = a.val[c + 2 * nch + lane];
			}
			const char *tile = tile0 + buf * tile_bytes;
			// ---- phase A
			double e;
			{
				const double2 *t2 = reinterpret_cast<const double2 *>(tile + (lane < nch ? lane : 0) * S);
				double dot = 0.0;
				if constexpr (KT > 0) {
					dot = phase_a_dot_pipelined<KT / 2, 8>(t2, xs);
				} else if (KT > 0) {
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
			// ---- phase B
			const char *tb = tile + voff;
			int n = 0;
			if constexpr (NP == 1) {   // sixteen entries' reads in flight at once
				for (; n + 16 <= cnt; n += 16) {
					double2 t[16];
#pragma unroll
					for (int u = 0; u < 16; ++u)
						t[u] = (lane < P) ? *reinterpret_cast<const double2 *>(tb + (n + u) * S) : make_double2(0.0, 0.0);
					asm volatile("" ::: "memory");   // pins the issue order: the reads above go out before the arithmetic below
#pragma unroll
					for (int u = 0; u < 16; ++u) {
						const double en = readlane_f64(e, n + u);
						acc[0].x = acc[0].x + en * t[u].x;
						acc[0].y = acc[0].y + en * t[u].y;
					}
				}
			}
			for (; n + 4 <= cnt; n += 4) {
				double2 t[4][NP];
				double en[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					en[u] = readlane_f64(e, n + u);
#pragma unroll
					for (int p = 0; p < NP; ++p)
						t[u][p] = (lane + kWave * p < P) ? *reinterpret_cast<const double2 *>(tb + (n + u) * S + 1024 * p)
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
			// all LDS reads of this tile are complete before a later iteration's transfer may land in it
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
		{
			double2 *__restrict__ out2 = reinterpret_cast<double2 *>(a.X_new + (size_t) r * a.ldx);
#pragma unroll
			for (int p = 0; p < NP; ++p) {
				const int q = lane + kWave * p;
				if (q < P) out2[q] = acc[p];
			}
		}
		// xs is rewritten for the next row: its reads (phase A) are complete (lgkmcnt(0) above)
	}
}

// ------------------------------------------------------------------------------------------------
// Sweep kernel, wave-pair form: for launches that end on a few long rows (a few thousand rows, skewed lengths).
// In-kernel clocks of one wave walking a long row alone (tools/stamps.py, K=100, 16-entry chunks; cycles per chunk):
// gather issue 1430 (an LDS-DMA instruction costs its wave ~90 cycles), landing 370, phase A 1770, phase B 1100 --
// the wave is bound by its own instruction stream, so neither more bytes in flight nor a second tile helps it
// (sweep_db_kernel: no gain).  Here TWO waves own the row: wave 0 only gathers -- chunk c+1 into the other tile while
// wave 1 computes phases A and B on chunk c -- with one s_barrier per chunk: "chunk c has landed and chunk c-1 is
// consumed".  The row's walk costs max(issue + landing, A + B) per chunk instead of their sum, and the arithmetic is
// wave 1's alone, in the order of sweep_dma_kernel: same bits.  Compile-time K with one DMA instruction per row
// (64 <= K <= 128).
// ------------------------------------------------------------------------------------------------
template <int Q, int PF>
__device__ __forceinline__ double phase_a_dot_ahead(const double2 *t2, const double2 *xs)
{
	// as phase_a_dot_pipelined, and the two products of step q+1 are formed before the two adds of step q: the chain of
	// dependent adds never waits for a multiply
	double2 t[PF], x[PF];
#pragma unroll
	for (int i = 0; i < PF; ++i)
		if (i < Q) {
			t[i] = t2[i];
			x[i] = xs[i];
		}
	double dot = 0.0;
	double px = x[0].x * t[0].x, py = x[0].y * t[0].y;
#pragma unroll
	for (int q = 0; q < Q; ++q) {
		double nx = 0.0, ny = 0.0;
		if (q + 1 < Q) {
			nx = x[(q + 1) % PF].x * t[(q + 1) % PF].x;
			ny = x[(q + 1) % PF].y * t[(q + 1) % PF].y;
		}
		if (q + PF < Q) {
			t[q % PF] = t2[q + PF];
			x[q % PF] = xs[q + PF];
		}
		// (pinning the multiplies of step q+1 between the adds of steps q-1 and q with register operands on this statement
		// was measured SLOWER -- a lone 5993-entry row 0.327 -> 0.347 ms --: hipcc's own placement, each multiply in front
		// of its add, stays)
		asm volatile("" ::: "memory");
		dot = dot + px;
		dot = dot + py;
		px = nx;
		py = ny;
	}
	return dot;
}

// NL loader waves + one compute wave.  NL = 2: the two loaders issue alternate groups of four rows of a chunk, so the
// ~90 cycles per gathered row -- what a pair is bound by -- are paid in parallel and the walk becomes bound by the compute
// wave (phases A + B).
template <int KT, int NL = 1>
__global__ void __launch_bounds__((NL + 1) * kWave) sweep_pair_kernel(SweepArgs a)
{
	using G = DmaGeom<KT>;
	static_assert(G::kPasses == 1 && (G::kPieces | 1) > 32, "one LDS-DMA instruction per gathered row");
	constexpr int P = G::kPieces, S = G::kStride;
	extern __shared__ __attribute__((aligned(16))) char lds[];
	double2 *xs = reinterpret_cast<double2 *>(lds);
	char *tile0 = lds + G::kXsBytes;
	const int nch = a.nch;
	const int tile_bytes = nch * S;
	const unsigned tile0_lds = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) tile0;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));   // scalar: the two roles never share a branch mask
	const unsigned voff = (unsigned) lane * 16u;
	const unsigned long long ybase = (unsigned long long) a.Y_old;
	const unsigned long long ybytes = (unsigned long long) a.ldy * 8ull;

	for (int it = blockIdx.x; it < a.nrows; it += gridDim.x) {
		const int r = a.rowlist ? a.rowlist[it] : it;
		const int beg = a.ptr[r], end = a.ptr[r + 1];
		// long rows are what the launch ends on: both of their waves run at raised priority (instruction issue is
		// arbitrated by priority, then age) beside the waves of short rows on the same SIMDs
		const bool long_row = a.prio_len > 0 && end - beg >= a.prio_len;
		if (wave < NL) {
			// ---------------- loader(s): chunk c -> tile c & 1, then "landed" = barrier c
			if (long_row)
				__builtin_amdgcn_s_setprio(3);
			else
				__builtin_amdgcn_s_setprio(1);   // its few instructions gate the compute wave
			int nx_idx = 0;
			if (beg + lane < min(end, beg + nch)) nx_idx = a.idx[beg + lane];
			int buf = 0;
			for (int c = beg; c < end; c += nch, buf ^= 1) {
				const int cnt = min(nch, end - c);
				int my_idx = nx_idx;
				asm volatile("" : "+v"(my_idx));   // hipcc's wait for this load stays here, outside the loops below
				nx_idx = 0;
				if (c + nch + lane < min(end, c + 2 * nch)) nx_idx = a.idx[c + nch + lane];
				const unsigned long long rowaddr = ybase + (unsigned long long) (unsigned) my_idx * ybytes;
				const int alo = (int) (unsigned) rowaddr, ahi = (int) (unsigned) (rowaddr >> 32);
				const unsigned tb = tile0_lds + (unsigned) (buf * tile_bytes);
				int n = 4 * wave;   // loader w takes the groups of four rows w, w + NL, ...
				for (; n + 4 <= cnt; n += 4 * NL) {
					unsigned long long b[4];
#pragma unroll
					for (int u = 0; u < 4; ++u)
						b[u] = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(ahi, n + u) << 32) |
						       (unsigned long long) (unsigned) __builtin_amdgcn_readlane(alo, n + u);
					if (lane < P) {
#pragma unroll
						for (int u = 0; u < 4; ++u)
							asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0"
							             :
							             : "s"(b[u]), "s"(tb + (unsigned) ((n + u) * S)), "v"(voff)
							             : "memory");
					}
				}
				for (int m = n; m < min(n + 4, cnt); ++m) {   // the last, partial group of four belongs to the loader it falls to
					const unsigned long long b = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(ahi, m) << 32) |
					                             (unsigned long long) (unsigned) __builtin_amdgcn_readlane(alo, m);
					if (lane < P)
						asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0"
						             :
						             : "s"(b), "s"(tb + (unsigned) (m * S)), "v"(voff)
						             : "memory");
				}
				// landed (this also retires the index load of the next chunk, issued in front of the transfers)
				asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
			}
			asm volatile("s_barrier" ::: "memory");   // row end: the compute wave has consumed the last tile
		} else {
			// ---------------- compute: phases A and B of chunk c after barrier c
			if (long_row)
				__builtin_amdgcn_s_setprio(3);
			else
				__builtin_amdgcn_s_setprio(0);
			const double2 *__restrict__ xrow2 = reinterpret_cast<const double2 *>(a.X_old + (size_t) r * a.ldx);
			double2 acc = make_double2(0.0, 0.0);
			if (lane < P) {
				const double2 v = xrow2[lane];
				xs[lane] = v;
				if (a.seed) acc = v;
			}
			double nx_val = 0.0;
			if (beg + lane < min(end, beg + nch)) nx_val = a.val[beg + lane];
			const unsigned boff = (unsigned) (lane < P ? lane : 0) * 16u;   // lanes beyond the row re-read piece 0: no branch masks in phase B
			int buf = 0;
			for (int c = beg; c < end; c += nch, buf ^= 1) {
				const int cnt = min(nch, end - c);
				double my_val = nx_val;
				nx_val = 0.0;
				if (c + nch + lane < min(end, c + 2 * nch)) nx_val = a.val[c + nch + lane];
				asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(my_val)::"memory");   // chunk c has landed
				const char *tile = tile0 + buf * tile_bytes;
				const double2 *t2 = reinterpret_cast<const double2 *>(tile + (lane < nch ? lane : 0) * S);
				const double dot = phase_a_dot_ahead<KT / 2, 8>(t2, xs);
				const double e = a.c2 * (my_val - dot);
				const char *tb = tile + boff;
				int n = 0;
				for (; n + 16 <= cnt; n += 16) {
					double2 t[16];
#pragma unroll
					for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const double2 *>(tb + (n + u) * S);
					asm volatile("" ::: "memory");
#pragma unroll
					for (int u = 0; u < 16; ++u) {
						const double en = readlane_f64(e, n + u);
						acc.x = acc.x + en * t[u].x;
						acc.y = acc.y + en * t[u].y;
					}
				}
				for (; n < cnt; ++n) {
					const double en = readlane_f64(e, n);
					const double2 t = *reinterpret_cast<const double2 *>(tb + n * S);
					acc.x = acc.x + en * t.x;
					acc.y = acc.y + en * t.y;
				}
			}
			if (lane < P) reinterpret_cast<double2 *>(a.X_new + (size_t) r * a.ldx)[lane] = acc;
			// row end: every read of the last tile and of xs is complete before the loader refills / xs is rewritten
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		}
	}
}

// Three waves per row: loader, phase A, phase B -- the pair's compute wave split in two.  A pair walks a row at
// max(issue + landing, A + B) per chunk and is bound by A + B (clocks of a lone long row, tools/stamps.py: per 16 entries
// issue 1430 / landing 370 / phase A 1770 / phase B 1100 cycles; phase A costs the same for 32 entries, lane = entry); with
// the phases on separate waves, one chunk apart, the walk costs max(issue + landing, A, B).  Lockstep pipeline, ONE barrier
// per step s: the loader fills tile s % 3 with chunk s, the A wave forms the errors of chunk s-1 (tile (s-1) % 3) and parks
// them in ebuf[(s-1) & 1], the B wave accumulates chunk s-2 (tile (s-2) % 3, errors from ebuf[s & 1]).  The barrier that ends
// step s says: chunk s has landed, the errors of chunk s-1 are in LDS, tile (s-2) % 3 = (s+1) % 3 is consumed.  The
// arithmetic is that of the pair's compute wave, in the same order => same bits.
// Measured (tools/r3_trio_ab.sh, profiles/r03/trio_ab.txt; experiments build, MF_SWEEP_TRIO=1): the side whose time IS its
// longest row gains -- cfg3 power-law users 0.138 -> 0.121 ms -- but every throughput-bound side loses to the third tile
// (fewer workgroups per CU): cfg3 power-law items 0.115 -> 0.151, Netflix-shape items 9.97 -> 10.48 ms, cfg4 items 11.84 ->
// 12.68.  And the users' gain was the item side's ordered sums no longer overlapping the user sweep: with trios on the user
// side ALONE (MF_SWEEP_TRIO_U=1) it is 0.1447 against 0.1378 ms.  Not chosen by any rule; experiments build only.
template <int KT>
__global__ void __launch_bounds__(3 * kWave) sweep_trio_kernel(SweepArgs a)
{
	using G = DmaGeom<KT>;
	static_assert(G::kPasses == 1 && (G::kPieces | 1) > 32, "one LDS-DMA instruction per gathered row");
	constexpr int P = G::kPieces, S = G::kStride;
	extern __shared__ __attribute__((aligned(16))) char lds[];
	double2 *xs = reinterpret_cast<double2 *>(lds);
	double *ebuf = reinterpret_cast<double *>(lds + G::kXsBytes);   // 2 x 64 errors
	char *tile0 = lds + G::kXsBytes + 1024;
	const int nch = a.nch;
	const int tile_bytes = nch * S;
	const unsigned tile0_lds = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) tile0;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));   // scalar: the roles never share a branch mask
	const unsigned voff = (unsigned) lane * 16u;
	const unsigned long long ybase = (unsigned long long) a.Y_old;
	const unsigned long long ybytes = (unsigned long long) a.ldy * 8ull;

	for (int it = blockIdx.x; it < a.nrows; it += gridDim.x) {
		const int r = a.rowlist ? a.rowlist[it] : it;
		const int beg = a.ptr[r], end = a.ptr[r + 1];
		const int nchunks = (end - beg + nch - 1) / nch;
		const bool long_row = a.prio_len > 0 && end - beg >= a.prio_len;
		if (long_row)
			__builtin_amdgcn_s_setprio(3);
		else if (wave == 0)
			__builtin_amdgcn_s_setprio(1);   // its few instructions gate the two other waves
		else
			__builtin_amdgcn_s_setprio(0);
		if (wave == 0) {
			// ---------------- loader: chunk s -> tile s % 3
			int nx_idx = 0;
			if (beg + lane < min(end, beg + nch)) nx_idx = a.idx[beg + lane];
			int ti = 0;
			for (int s = 0; s < nchunks + 2; ++s) {
				if (s < nchunks) {
					const int c = beg + s * nch;
					const int cnt = min(nch, end - c);
					int my_idx = nx_idx;
					asm volatile("" : "+v"(my_idx));   // hipcc's wait for this load stays here, outside the loops below
					nx_idx = 0;
					if (c + nch + lane < min(end, c + 2 * nch)) nx_idx = a.idx[c + nch + lane];
					const unsigned long long rowaddr = ybase + (unsigned long long) (unsigned) my_idx * ybytes;
					const int alo = (int) (unsigned) rowaddr, ahi = (int) (unsigned) (rowaddr >> 32);
					const unsigned tb = tile0_lds + (unsigned) (ti * tile_bytes);
					int n = 0;
					for (; n + 4 <= cnt; n += 4) {
						unsigned long long b[4];
#pragma unroll
						for (int u = 0; u < 4; ++u)
							b[u] = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(ahi, n + u) << 32) |
							       (unsigned long long) (unsigned) __builtin_amdgcn_readlane(alo, n + u);
						if (lane < P) {
#pragma unroll
							for (int u = 0; u < 4; ++u)
								asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0"
								             :
								             : "s"(b[u]), "s"(tb + (unsigned) ((n + u) * S)), "v"(voff)
								             : "memory");
						}
					}
					for (; n < cnt; ++n) {
						const unsigned long long b = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(ahi, n) << 32) |
						                             (unsigned long long) (unsigned) __builtin_amdgcn_readlane(alo, n);
						if (lane < P)
							asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0"
							             :
							             : "s"(b), "s"(tb + (unsigned) (n * S)), "v"(voff)
							             : "memory");
					}
				}
				// landed (this also retires the index load of the next chunk, issued in front of the transfers)
				asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
				ti = ti == 2 ? 0 : ti + 1;
			}
		} else if (wave == 1) {
			// ---------------- phase A of chunk s - 1 at step s: the errors, parked in ebuf[(s - 1) & 1]
			const double2 *__restrict__ xrow2 = reinterpret_cast<const double2 *>(a.X_old + (size_t) r * a.ldx);
			if (lane < P) xs[lane] = xrow2[lane];
			double nx_val = 0.0;
			if (beg + lane < min(end, beg + nch)) nx_val = a.val[beg + lane];
			int ti = 0;   // tile of chunk s - 1
			for (int s = 0; s < nchunks + 2; ++s) {
				if (s >= 1 && s <= nchunks) {
					const int c = beg + (s - 1) * nch;
					const double my_val = nx_val;
					nx_val = 0.0;
					if (c + nch + lane < min(end, c + 2 * nch)) nx_val = a.val[c + nch + lane];
					const char *tile = tile0 + ti * tile_bytes;
					const double2 *t2 = reinterpret_cast<const double2 *>(tile + (lane < nch ? lane : 0) * S);
					const double dot = phase_a_dot_ahead<KT / 2, 8>(t2, xs);
					ebuf[((s - 1) & 1) * kWave + lane] = a.c2 * (my_val - dot);
					ti = ti == 2 ? 0 : ti + 1;
				}
				asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(nx_val)::"memory");
			}
		} else {
			// ---------------- phase B of chunk s - 2 at step s
			const double2 *__restrict__ xrow2 = reinterpret_cast<const double2 *>(a.X_old + (size_t) r * a.ldx);
			double2 acc = make_double2(0.0, 0.0);
			if (lane < P && a.seed) acc = xrow2[lane];
			const unsigned boff = (unsigned) (lane < P ? lane : 0) * 16u;   // lanes beyond the row re-read piece 0: no branch masks
			int ti = 0;   // tile of chunk s - 2
			for (int s = 0; s < nchunks + 2; ++s) {
				if (s >= 2) {
					const int c = beg + (s - 2) * nch;
					const int cnt = min(nch, end - c);
					const double e = ebuf[(s & 1) * kWave + lane];
					const char *tb = tile0 + ti * tile_bytes + boff;
					int n = 0;
					for (; n + 16 <= cnt; n += 16) {
						double2 t[16];
#pragma unroll
						for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const double2 *>(tb + (n + u) * S);
						asm volatile("" ::: "memory");
#pragma unroll
						for (int u = 0; u < 16; ++u) {
							const double en = readlane_f64(e, n + u);
							acc.x = acc.x + en * t[u].x;
							acc.y = acc.y + en * t[u].y;
						}
					}
					for (; n < cnt; ++n) {
						const double en = readlane_f64(e, n);
						const double2 t = *reinterpret_cast<const double2 *>(tb + n * S);
						acc.x = acc.x + en * t.x;
						acc.y = acc.y + en * t.y;
					}
					ti = ti == 2 ? 0 : ti + 1;
				}
				asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(acc.x), "+v"(acc.y)::"memory");
			}
			if (lane < P) reinterpret_cast<double2 *>(a.X_new + (size_t) r * a.ldx)[lane] = acc;
		}
	}
}

// Ordered sum of the scaled rows of one extreme row: X_new[r][k] = (...((seed + p_0[k]) + p_1[k]) + ...), the
// serial accumulation order.  One wave per (row, 8-column slice).  The slice is contiguous over the entries (64 B
// each), so ONE LDS-DMA instruction brings a block of 16 consecutive entries (1 KiB) into a slot of an LDS ring and the
// wave keeps D-1 blocks in flight ahead of the block it is adding: that hides the ~2 us read latency behind the only
// true critical path, the chain of dependent adds.  The DMA and its s_waitcnt are inline asm with hand-counted vmcnt
// (hipcc would otherwise wait vmcnt(0) before every LDS read that may alias a pending LDS-DMA); no prefetch registers
// exist, so nothing can be sunk or spilled.
//
// ONE LDS read per block: the DMA's per-lane source address permutes the block so that the lane-linear LDS image is
// [piece][entry] -- DPP row p (16 lanes) holds piece p of entries 0..15 --, one ds_read_b128 hands every lane ITS
// entry's two doubles, and the running sum, replicated over the 16 lanes of a row, takes entry e by
//   v_fmac_f64_dpp acc, v, 1.0  row_newbcast:e          acc <- v[lane e of my row] * 1.0 + acc
// (gfx90a+ DPP on FP64: row_newbcast only, and only on VOP1/VOP2 encodings -- v_add_f64 is VOP3, v_fmac_f64 is VOP2).
// The product by 1.0 is exact, so the single rounding is that of the add: the same bits as acc + v.  (The first form had
// every lane read all 16 entries: 16 ds_read_b128 of 1 KiB per block and wave, 512 LDS cycles per block with four waves
// on a CU -- 217 ns per block for the longest row of the cfg3 power-law shape against 150 now; tools/micro/osum_probe.
// That form is still here as the DPP = false instantiation, selected at plan creation by MF_OS_DPP=0.)
//
// Hand-written region, invariants by construction (tests/test_isa.py checks them on the built code object):
//   * an LDS read whose result hipcc cannot see pending (inline asm) is WAITED FOR INSIDE THE SAME asm STATEMENT, and its
//     destination is an early-clobber output: no register holds a pending LDS return across statements, so no compiler
//     copy, spill or reuse can observe one (cdna_hip_programming.md 5.7 item 1, form (i));
//   * the steady state keeps the read of the next block in flight under the chain of the current one -- inside ONE
//     statement, which ends with the wait.
// Why: round 2 shipped, for a while, a two-register software pipeline -- the read of block b+1 issued from an asm statement
// before the chain of block b, `s_waitcnt lgkmcnt(0)` in another asm statement at the top of the next step -- that gave one
// wrong sum per ~1e8 blocks, only beside another kernel on a busy chip, and was blamed on the hardware (DPP beside a
// returning LDS read).  It was a SOFTWARE bug, settled in round 3 from the builds kept in tools/micro/alt and with the
// diagnostic build below (tools/os_diag.py): the loop's odd tail step ended in `cur = nxt;` in front of `landed(cur)`, and
// hipcc -- for which an asm output is written when the statement ends -- emitted that copy as two v_mov_b64 of the PENDING
// quad in front of the wait (libmatfact_hip_base.so: ds_read_b128 v[24:27] ... v_mov_b64 v[6:7], v[24:25] / v[8:9],
// v[26:27] ... s_waitcnt lgkmcnt(0), in all three depth classes).  Whenever the read took longer than the chain in between
// -- ~300 cycles, i.e. only on a CU whose LDS is busy -- the copy caught the register before the data.  The order that
// "fixed" it (read after the chain) was clean only because hipcc happened to coalesce the two quads there.  Re-built this
// round, the old order fails 93 of 1500 lockstep iterations; with one lgkmcnt(0) in front of the copy, none of 9.6e8 blocks.
//
// Depth classes: a CU's miss bandwidth (~29 GB/s) is shared by its resident waves in proportion to what each keeps in
// flight, so a wave streaming the longest row must hold more than the waves of the merely long rows beside it, or it
// crawls at a quarter of the CU while they finish early.  D = blocks in flight + 1 is picked per row from its length
// relative to the longest one (R-1, R/2 or R/4 of the R ring slots).
struct OrderedSumArgs {
	int nrows, K, seed, nslices;
	int ldx;                              // row pitch of X in doubles
	int max_cnt;                          // entries of the longest row of the launch
	const int *__restrict__ row;          // extreme row ids
	const long long *__restrict__ sbeg;   // first scratch entry of the row
	const int *__restrict__ cnt;          // entries of the row
	const double *__restrict__ scratch;   // [slice][entry][kSliceCols], each slice padded by one block
	size_t scratch_entries;
	const double *__restrict__ X_old;
	double *__restrict__ X_new;
	unsigned long long *stamps;           // probe builds only (tools/micro/osum_probe.hip): 4 clock stamps per task, else null
};

constexpr int kRing = 32;   // LDS ring slots of 1 KiB (a power of two; kRing <= 63 = the largest vmcnt)
constexpr size_t kOrderedSumLds = (size_t) (kRing + 1) * 1024;   // ring + the seed's slot

typedef double v2d __attribute__((ext_vector_type(2)));

#define MF_FMAC_BCAST(E)                                                                                              \
	asm volatile("v_fmac_f64_dpp %0, %2, %4 row_newbcast:" #E " row_mask:0xf bank_mask:0xf\n\t"                        \
	             "v_fmac_f64_dpp %1, %3, %4 row_newbcast:" #E " row_mask:0xf bank_mask:0xf"                            \
	             : "+v"(ax), "+v"(ay)                                                                                  \
	             : "v"(v.x), "v"(v.y), "v"(one))

// One (row, slice) task with D-1 blocks in flight.  `src` = this lane's 16 bytes of block 0 (piece lane >> 4 of entry
// lane & 15), `my` = its 16 bytes of ring slot 0, `seed_ptr` = its two columns of X_old (null: start from zero).
// DPP = false: the plain form -- every lane reads the 16 entries of its piece (LDS broadcast reads) and adds them with
// v_add_f64; same DMA image, same order, same bits.
template <int D, bool DPP>
__device__ __forceinline__ void ordered_sum_task(const char *src, unsigned ring_base, unsigned my, int cnt,
                                                 const double *seed_ptr, double &ax, double &ay, double one,
                                                 unsigned long long &t_issued)
{
	static_assert(D >= 4 && D <= kRing, "depth");
	constexpr int EB = kBlockEntries;
	const int nblk = (cnt + EB - 1) / EB;
	const unsigned piece_base = (my & ~1023u) + ((my & 1023u) >> 8 << 8);   // plain form: piece p of entry e sits at 256 p + 16 e
	auto slot_of = [&](int b) { return (unsigned) (b & (kRing - 1)) * 1024u; };
	auto issue = [&](int b) {
		const char *g = src + (size_t) b * 1024;
		const unsigned m0 = __builtin_amdgcn_readfirstlane(ring_base + slot_of(b));
		asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0) : "memory");   // m0 is a reserved register: hipcc re-loads it before each of its own uses
	};
	// Block b out of the ring into registers, complete when the statement ends (early-clobber output, the wait inside).
	// ISSUE: the DMA of block `nb` goes out between the read and its wait, so a part of the read's latency is hidden.
	// WAIT = how many newer DMAs may still be outstanding for block b to have landed (-1: the caller has waited).
	auto fetch = [&](int b, v2d &v, auto wait_tag, int nb) {
		constexpr int WAIT = decltype(wait_tag)::value;
		const unsigned addr = my + slot_of(b);
		if constexpr (WAIT >= 0) {   // steady state: nb = b + D - 1 is always a block of the row
			const char *g = src + (size_t) nb * 1024;
			const unsigned m0 = __builtin_amdgcn_readfirstlane(ring_base + slot_of(nb));
			asm volatile("s_waitcnt vmcnt(%4)\n\t"
			             "ds_read_b128 %0, %1\n\t"
			             "s_mov_b32 m0, %3\n\t"
			             "s_nop 0\n\t"
			             "global_load_lds_dwordx4 %2, off\n\t"
			             "s_waitcnt lgkmcnt(0)"
			             : "=&v"(v)
			             : "v"(addr), "v"(g), "s"(m0), "n"(WAIT)
			             : "memory");
		} else {
			(void) nb;
			asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
		}
	};
	auto add_block = [&](int b, int n, auto wait_tag, int nb) {   // entries 0..n-1 of block b onto (ax, ay), in order
		if constexpr (DPP) {
			v2d v;
			fetch(b, v, wait_tag, nb);
			if (n == EB) {
				MF_FMAC_BCAST(0); MF_FMAC_BCAST(1); MF_FMAC_BCAST(2); MF_FMAC_BCAST(3);
				MF_FMAC_BCAST(4); MF_FMAC_BCAST(5); MF_FMAC_BCAST(6); MF_FMAC_BCAST(7);
				MF_FMAC_BCAST(8); MF_FMAC_BCAST(9); MF_FMAC_BCAST(10); MF_FMAC_BCAST(11);
				MF_FMAC_BCAST(12); MF_FMAC_BCAST(13); MF_FMAC_BCAST(14); MF_FMAC_BCAST(15);
			} else {   // the last, partial block of a row
				if (n > 0) MF_FMAC_BCAST(0);
				if (n > 1) MF_FMAC_BCAST(1);
				if (n > 2) MF_FMAC_BCAST(2);
				if (n > 3) MF_FMAC_BCAST(3);
				if (n > 4) MF_FMAC_BCAST(4);
				if (n > 5) MF_FMAC_BCAST(5);
				if (n > 6) MF_FMAC_BCAST(6);
				if (n > 7) MF_FMAC_BCAST(7);
				if (n > 8) MF_FMAC_BCAST(8);
				if (n > 9) MF_FMAC_BCAST(9);
				if (n > 10) MF_FMAC_BCAST(10);
				if (n > 11) MF_FMAC_BCAST(11);
				if (n > 12) MF_FMAC_BCAST(12);
				if (n > 13) MF_FMAC_BCAST(13);
				if (n > 14) MF_FMAC_BCAST(14);
			}
		} else {
			constexpr int WAIT = decltype(wait_tag)::value;
			if constexpr (WAIT >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
			const unsigned addr = piece_base + slot_of(b);
			v2d w[EB];
			// 16 reads and their wait in one statement (the ring is written by LDS-DMA hipcc knows nothing of)
			asm volatile("ds_read_b128 %0, %16\n\tds_read_b128 %1, %16 offset:16\n\tds_read_b128 %2, %16 offset:32\n\t"
			             "ds_read_b128 %3, %16 offset:48\n\tds_read_b128 %4, %16 offset:64\n\tds_read_b128 %5, %16 offset:80\n\t"
			             "ds_read_b128 %6, %16 offset:96\n\tds_read_b128 %7, %16 offset:112\n\tds_read_b128 %8, %16 offset:128\n\t"
			             "ds_read_b128 %9, %16 offset:144\n\tds_read_b128 %10, %16 offset:160\n\tds_read_b128 %11, %16 offset:176\n\t"
			             "ds_read_b128 %12, %16 offset:192\n\tds_read_b128 %13, %16 offset:208\n\tds_read_b128 %14, %16 offset:224\n\t"
			             "ds_read_b128 %15, %16 offset:240\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]),
			               "=&v"(w[8]), "=&v"(w[9]), "=&v"(w[10]), "=&v"(w[11]), "=&v"(w[12]), "=&v"(w[13]), "=&v"(w[14]),
			               "=&v"(w[15])
			             : "v"(addr)
			             : "memory");
			if (nb >= 0) issue(nb);
#pragma unroll
			for (int e = 0; e < EB; ++e)
				if (e < n) {
					ax = ax + w[e].x;
					ay = ay + w[e].y;
				}
		}
	};
	// Hand-counted region.  The seed travels like a block -- an LDS-DMA transfer into a slot of its own, issued BEFORE
	// the blocks: LDS-DMA transfers land in order, so it has landed whenever block 0 has, and its round trip runs beside
	// theirs instead of in front of them.  (An ordinary load into a register issued before the transfers is NOT ordered
	// with them: on a busy chip the first add was seen to read the register before the load had returned.)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing older in flight (the store of a previous task)
	if (seed_ptr) {
		const unsigned m0 = __builtin_amdgcn_readfirstlane(ring_base + (unsigned) kRing * 1024u);
		asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(seed_ptr), "s"(m0) : "memory");
	}
	auto read_seed = [&]() {   // caller: the seed's transfer has landed
		v2d sv = {0.0, 0.0};
		if (seed_ptr) {
			const unsigned addr = my + (unsigned) kRing * 1024u;
			asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sv) : "v"(addr) : "memory");
		}
		ax = sv.x;
		ay = sv.y;
	};
	const int ahead = min(nblk, D - 1);
	for (int b = 0; b < ahead; ++b) issue(b);
	t_issued = __builtin_amdgcn_s_memrealtime();   // (probe) no store inside the hand-counted region
	int b = 0;
	// Steady state, per block b: wait until b has landed -- blocks up to b+D-2 are issued, so at most the D-2 newer ones
	// may be outstanding --, read it, issue block b+D-1 into the slot of block b-1 (read and added one step ago; with
	// D-1 < kRing an older one), then the 16 dependent adds, the true critical path.
	if (nblk > D - 1) {   // at least one block is still to be issued
		asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 2) : "memory");   // block 0, and the seed in front of it
		read_seed();
		if constexpr (DPP) {
			// Software-pipelined by one block, each step ONE asm statement: wait until block blk+1 has landed (blocks up to
			// blk+D-2 are issued: at most D-3 newer ones outstanding), start its LDS read into `nxt`, run the chain of block
			// blk out of `cur` -- complete since the previous statement ended --, issue block blk+D-1 into the slot of block
			// blk-1, and WAIT for the read before the statement ends.  The read's latency hides under the chain, and no
			// register is pending across statements: whatever hipcc does with `nxt` afterwards (the copy at the end of an
			// odd run of steps below) it does to landed data.
#define MF_CHAIN2(E)                                                                                                     \
	"v_fmac_f64_dpp %[ax], %[cx], %[one] row_newbcast:" #E " row_mask:0xf bank_mask:0xf\n\t"                            \
	"v_fmac_f64_dpp %[ay], %[cy], %[one] row_newbcast:" #E " row_mask:0xf bank_mask:0xf\n\t"
			auto pstep = [&](int blk, const v2d &cur, v2d &nxt) {
				const unsigned addr = my + slot_of(blk + 1);
				const char *g = src + (size_t) (blk + D - 1) * 1024;
				const unsigned m0 = __builtin_amdgcn_readfirstlane(ring_base + slot_of(blk + D - 1));
				asm volatile("s_waitcnt vmcnt(%[w])\n\t"
				             "ds_read_b128 %[nxt], %[addr]\n\t"
				             MF_CHAIN2(0) MF_CHAIN2(1) MF_CHAIN2(2) MF_CHAIN2(3) MF_CHAIN2(4) MF_CHAIN2(5) MF_CHAIN2(6) MF_CHAIN2(7)
				             MF_CHAIN2(8) MF_CHAIN2(9) MF_CHAIN2(10) MF_CHAIN2(11) MF_CHAIN2(12) MF_CHAIN2(13) MF_CHAIN2(14) MF_CHAIN2(15)
				             "s_mov_b32 m0, %[m0]\n\t"
				             "s_nop 0\n\t"
				             "global_load_lds_dwordx4 %[g], off\n\t"
				             "s_waitcnt lgkmcnt(0)"
				             : [ax] "+v"(ax), [ay] "+v"(ay), [nxt] "=&v"(nxt)
				             : [cx] "v"(cur.x), [cy] "v"(cur.y), [one] "v"(one), [addr] "v"(addr), [g] "v"(g), [m0] "s"(m0), [w] "n"(D - 3)
				             : "memory");
			};
#undef MF_CHAIN2
			v2d r0, r1;
			fetch(0, r0, std::integral_constant<int, -1>{}, -1);
			const int last = nblk - D;   // the last block whose step still has a block to issue
			for (; b + 1 <= last; b += 2) {   // two steps per trip: the two registers swap roles
				pstep(b, r0, r1);
				pstep(b + 1, r1, r0);
			}
			if (b <= last) {
				pstep(b, r0, r1);
				r0 = r1;   // a copy of LANDED data
				++b;
			}
			// r0 holds block b = nblk - D + 1, a full one, not yet added; every block is issued
			{
				const v2d v = r0;
				MF_FMAC_BCAST(0); MF_FMAC_BCAST(1); MF_FMAC_BCAST(2); MF_FMAC_BCAST(3);
				MF_FMAC_BCAST(4); MF_FMAC_BCAST(5); MF_FMAC_BCAST(6); MF_FMAC_BCAST(7);
				MF_FMAC_BCAST(8); MF_FMAC_BCAST(9); MF_FMAC_BCAST(10); MF_FMAC_BCAST(11);
				MF_FMAC_BCAST(12); MF_FMAC_BCAST(13); MF_FMAC_BCAST(14); MF_FMAC_BCAST(15);
			}
			++b;
		} else
			for (; b + (D - 1) < nblk; ++b) add_block(b, EB, std::integral_constant<int, D - 2>{}, b + D - 1);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (b == 0) read_seed();
	for (; b < nblk; ++b) add_block(b, min(EB, cnt - EB * b), std::integral_constant<int, -1>{}, -1);
}
#ifdef MF_OS_DIAG
// Diagnostic builds only (make csrc/libmatfact_hip_osdiag<level>.so; tools/os_diag.py): ROUND 2'S LOOP -- the LDS read of
// block b+1 issued before the v_fmac_f64_dpp chain of block b and waited for after it, two registers swapping roles, the
// odd tail step ending in `cur = nxt` -- with two checks after every chain, written to tell the candidate causes apart:
//   bit 0  the register the chain consumed differs from what the block's LDS slot holds NOW (re-read after the chain):
//          the read returned before the transfer had landed, or returned something else than the slot's bytes;
//   bit 1  the chain's result differs from the same 16 adds formed without DPP from the SAME register (lane e of the
//          row fetched by ds_bpermute): the DPP chain mis-executed.
// Neither bit set on a launch whose result is wrong: the slot itself held wrong bytes (the scratch as this wave's
// transfers saw it).  mf_os_diag: [0] blocks checked, [1] records, then 8 words per record.
// What the levels showed (tenth-scale Netflix shape, MF_SWEEP_LONG=3000, 1500 lockstep iterations each): level 1 (the old
// loop, no checks) 93 wrong iterations; level 4 (the transfer issued before the chain) 131; levels 2, 3 and 5 (a check --
// hence an `s_waitcnt lgkmcnt(0)` -- after the chain, in front of or behind the transfer) none, no record.  The checks
// cured what they were looking for: the wait they add sits in front of the compiler's copy of the pending quad (the
// `cur = nxt` of the tail step, see the ISA of level 1: v_mov_b64 v[8:9], v[26:27] / v[10:11], v[28:29] ahead of the
// lgkmcnt(0)).  Cause = that copy; neither DPP nor the transfers.
__device__ unsigned long long mf_os_diag[2 + 8 * 32];

template <int D>
__device__ __forceinline__ void ordered_sum_task_diag(const char *src, unsigned ring_base, unsigned my, int cnt,
                                                      const double *seed_ptr, double &ax, double &ay, double one, int task)
{
	constexpr int EB = kBlockEntries;
	const int nblk = (cnt + EB - 1) / EB;
	const int lane = threadIdx.x;
	auto slot_of = [&](int b) { return (unsigned) (b & (kRing - 1)) * 1024u; };
	auto issue = [&](int b) {
		const char *g = src + (size_t) b * 1024;
		const unsigned m0 = __builtin_amdgcn_readfirstlane(ring_base + slot_of(b));
		asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0) : "memory");
	};
	auto read_block = [&](int b, v2d &v) {
		const unsigned addr = my + slot_of(b);
		asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
	};
	auto landed = [&](v2d &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory"); };
	auto add16 = [&](const v2d &v) {
		MF_FMAC_BCAST(0); MF_FMAC_BCAST(1); MF_FMAC_BCAST(2); MF_FMAC_BCAST(3);
		MF_FMAC_BCAST(4); MF_FMAC_BCAST(5); MF_FMAC_BCAST(6); MF_FMAC_BCAST(7);
		MF_FMAC_BCAST(8); MF_FMAC_BCAST(9); MF_FMAC_BCAST(10); MF_FMAC_BCAST(11);
		MF_FMAC_BCAST(12); MF_FMAC_BCAST(13); MF_FMAC_BCAST(14); MF_FMAC_BCAST(15);
	};
	auto checked_add16 = [&](int blk, const v2d &v) {
		// MF_OS_DIAG = 1: the old order alone (does it still fail beside this round's kernels?), 2: + the re-read check,
		// 3: + the DPP-free recomputation (32 ds_bpermute per block: it changes the timing the most)
		const double bx = ax, by = ay;
		add16(v);
		double px = ax, py = ay;
		v2d chk = v;
#if MF_OS_DIAG >= 3
		// (bit 1) the same sixteen adds without DPP, from the same register
		px = bx;
		py = by;
#pragma unroll
		for (int e = 0; e < 16; ++e) {
			px = px + __shfl(v.x, (lane & ~15) + e);
			py = py + __shfl(v.y, (lane & ~15) + e);
		}
#else
		(void) bx;
		(void) by;
#endif
#if MF_OS_DIAG >= 2
		// (bit 0) the slot again, now that the chain is over (it is refilled one step later at the earliest)
		{
			const unsigned addr = my + slot_of(blk);
			asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(chk) : "v"(addr) : "memory");
		}
#endif
		int flags = 0;
		if (__double_as_longlong(chk.x) != __double_as_longlong(v.x) || __double_as_longlong(chk.y) != __double_as_longlong(v.y)) flags |= 1;
		if (__double_as_longlong(px) != __double_as_longlong(ax) || __double_as_longlong(py) != __double_as_longlong(ay)) flags |= 2;
#if MF_OS_DIAG >= 2
		if (lane == 0) atomicAdd(&mf_os_diag[0], 1ull);
#endif
		if (flags) {
			const unsigned long long slot = atomicAdd(&mf_os_diag[1], 1ull);
			if (slot < 32) {
				unsigned long long *r = mf_os_diag + 2 + 8 * slot;
				r[0] = (unsigned long long) task;
				r[1] = (unsigned long long) blk | ((unsigned long long) nblk << 32);
				r[2] = (unsigned long long) lane | ((unsigned long long) flags << 32) | ((unsigned long long) D << 40);
				r[3] = (unsigned long long) __double_as_longlong(v.x);
				r[4] = (unsigned long long) __double_as_longlong(chk.x);
				r[5] = (unsigned long long) __double_as_longlong(ax);
				r[6] = (unsigned long long) __double_as_longlong(px);
				r[7] = (unsigned long long) __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
			}
		}
	};
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (seed_ptr) {
		const unsigned m0 = __builtin_amdgcn_readfirstlane(ring_base + (unsigned) kRing * 1024u);
		asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(seed_ptr), "s"(m0) : "memory");
	}
	auto read_seed = [&]() {
		v2d sv = {0.0, 0.0};
		if (seed_ptr) {
			const unsigned addr = my + (unsigned) kRing * 1024u;
			asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sv) : "v"(addr) : "memory");
		}
		ax = sv.x;
		ay = sv.y;
	};
	const int ahead = min(nblk, D - 1);
	for (int b = 0; b < ahead; ++b) issue(b);
	int b = 0;
	if (nblk > D - 1) {
		v2d cur, nxt;
		asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 2) : "memory");
		read_seed();
		read_block(0, cur);
		auto step = [&](int blk, v2d &have, v2d &want) {   // round 2's failing order
			landed(have);
			asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 3) : "memory");
			read_block(blk + 1, want);   // in flight under the chain below
#if MF_OS_DIAG == 4
			issue(blk + D - 1);          // 4: the transfer issued BEFORE the chain instead of after it
			add16(have);
#elif MF_OS_DIAG == 5
			// 5: the failing sequence untouched -- read, chain, transfer -- and the checks only AFTER the transfer is out
			const double bx = ax, by = ay;
			add16(have);
			issue(blk + D - 1);
			{
				double px = bx, py = by;
#pragma unroll
				for (int e = 0; e < 16; ++e) {
					px = px + __shfl(have.x, (lane & ~15) + e);
					py = py + __shfl(have.y, (lane & ~15) + e);
				}
				v2d chk;
				const unsigned addr = my + slot_of(blk);
				asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(chk) : "v"(addr) : "memory");
				int flags = 0;
				if (__double_as_longlong(chk.x) != __double_as_longlong(have.x) || __double_as_longlong(chk.y) != __double_as_longlong(have.y)) flags |= 1;
				if (__double_as_longlong(px) != __double_as_longlong(ax) || __double_as_longlong(py) != __double_as_longlong(ay)) flags |= 2;
				if (lane == 0) atomicAdd(&mf_os_diag[0], 1ull);
				if (flags) {
					const unsigned long long slot = atomicAdd(&mf_os_diag[1], 1ull);
					if (slot < 32) {
						unsigned long long *r = mf_os_diag + 2 + 8 * slot;
						r[0] = (unsigned long long) task;
						r[1] = (unsigned long long) blk | ((unsigned long long) nblk << 32);
						r[2] = (unsigned long long) lane | ((unsigned long long) flags << 32) | ((unsigned long long) D << 40);
						r[3] = (unsigned long long) __double_as_longlong(have.x);
						r[4] = (unsigned long long) __double_as_longlong(chk.x);
						r[5] = (unsigned long long) __double_as_longlong(ax);
						r[6] = (unsigned long long) __double_as_longlong(px);
						r[7] = (unsigned long long) __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
					}
				}
			}
#else
			checked_add16(blk, have);
			issue(blk + D - 1);
#endif
		};
		for (; b + D < nblk; b += 2) {
			step(b, cur, nxt);
			step(b + 1, nxt, cur);
		}
		if (b + (D - 1) < nblk) {
			step(b, cur, nxt);
			cur = nxt;
			++b;
		}
		landed(cur);
		add16(cur);
		++b;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	} else {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		read_seed();
	}
	for (; b < nblk; ++b) {
		v2d v;
		asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(my + slot_of(b)) : "memory");
		const int n = min(EB, cnt - EB * b);
		if (n > 0) MF_FMAC_BCAST(0);
		if (n > 1) MF_FMAC_BCAST(1);
		if (n > 2) MF_FMAC_BCAST(2);
		if (n > 3) MF_FMAC_BCAST(3);
		if (n > 4) MF_FMAC_BCAST(4);
		if (n > 5) MF_FMAC_BCAST(5);
		if (n > 6) MF_FMAC_BCAST(6);
		if (n > 7) MF_FMAC_BCAST(7);
		if (n > 8) MF_FMAC_BCAST(8);
		if (n > 9) MF_FMAC_BCAST(9);
		if (n > 10) MF_FMAC_BCAST(10);
		if (n > 11) MF_FMAC_BCAST(11);
		if (n > 12) MF_FMAC_BCAST(12);
		if (n > 13) MF_FMAC_BCAST(13);
		if (n > 14) MF_FMAC_BCAST(14);
		if (n > 15) MF_FMAC_BCAST(15);
	}
}
#endif
#undef MF_FMAC_BCAST

template <bool DPP>
__global__ void __launch_bounds__(kWave) ordered_sum_kernel(OrderedSumArgs a)
{
	static_assert(kSliceCols == 8, "row p of the wave = piece p of the slice: four pieces");
	extern __shared__ __attribute__((aligned(1024))) char ring[];   // kOrderedSumLds bytes: the ring, then the seed's slot
	const int lane = threadIdx.x, K = a.K;
	const unsigned ring_base = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) ring;
	const int piece = lane >> 4, ent = lane & 15;
	const unsigned my = ring_base + 16u * (unsigned) lane;
	const int total = a.nrows * a.nslices;
	double one = 1.0;
	asm volatile("" : "+v"(one));   // a register operand (VOP2 src1), not a literal
	for (int it = blockIdx.x; it < total; it += gridDim.x) {
		const int li = it / a.nslices, slice = it % a.nslices;
		const int r = a.row[li], cnt = a.cnt[li];
		unsigned long long *stamp = (a.stamps && lane == 0) ? a.stamps + 4 * (size_t) it : nullptr;
		const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
		unsigned long long t_issued = 0;
		const int k0 = slice * kSliceCols + 2 * piece;   // this row-of-lanes' two columns
		const bool live = k0 < K;                        // K is even: k0 + 1 < K too
		// block b of the row in this slice: 16 entries of 64 B; this lane fetches piece `piece` of entry `ent`
		const char *src = reinterpret_cast<const char *>(
		                      a.scratch + (((size_t) slice * a.scratch_entries + (size_t) a.sbeg[li]) << kSliceShift)) +
		                  64 * ent + 16 * piece;
		// every lane a valid address (the dead lanes of the last slice fetch column 0; they never store)
		const double *seed_ptr = a.seed ? a.X_old + (size_t) r * a.ldx + (live ? k0 : 0) : nullptr;
		double ax = 0.0, ay = 0.0;
#ifdef MF_OS_DIAG
		if constexpr (DPP) {
			if (4 * (long long) cnt >= 2 * (long long) a.max_cnt)
				ordered_sum_task_diag<kRing>(src, ring_base, my, cnt, seed_ptr, ax, ay, one, it);
			else if (4 * (long long) cnt >= (long long) a.max_cnt)
				ordered_sum_task_diag<kRing / 2>(src, ring_base, my, cnt, seed_ptr, ax, ay, one, it);
			else
				ordered_sum_task_diag<kRing / 4>(src, ring_base, my, cnt, seed_ptr, ax, ay, one, it);
		} else
#endif
		// in flight: all of the ring for the longest rows, a half or a quarter of it for the shorter ones
		if (4 * (long long) cnt >= 2 * (long long) a.max_cnt)
			ordered_sum_task<kRing, DPP>(src, ring_base, my, cnt, seed_ptr, ax, ay, one, t_issued);
		else if (4 * (long long) cnt >= (long long) a.max_cnt)
			ordered_sum_task<kRing / 2, DPP>(src, ring_base, my, cnt, seed_ptr, ax, ay, one, t_issued);
		else
			ordered_sum_task<kRing / 4, DPP>(src, ring_base, my, cnt, seed_ptr, ax, ay, one, t_issued);
		if (live && ent == 0) *reinterpret_cast<double2 *>(a.X_new + (size_t) r * a.ldx + k0) = make_double2(ax, ay);
		if (stamp) {
			stamp[0] = t_start;
			stamp[1] = t_issued;
			stamp[2] = __builtin_amdgcn_s_memrealtime();
			stamp[3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Resident form for TOY instances (the reference's inst0/1/2: a handful of rows, up to 1e5 iterations): everything -- both generations of L and R and the CSR/CSC entries -- fits the LDS of ONE
// workgroup, so the whole iteration loop runs inside one launch with a workgroup barrier per iteration
// instead of two dependent kernel launches (~3.4 us each even when replayed from a graph).  Thread t owns
// factor row t (users first, then items) and walks its entries in file order: the same owner-computes
// arithmetic and order as the sweep kernels, so the result is the same bits.
// ------------------------------------------------------------------------------------------------
struct ResidentArgs {
	int users, items, K, iters;
	int ldl, ldr;                               // row pitch of L and of R in global memory (doubles)
	double c2;                                  // alpha * 2
	const int *__restrict__ csr_ptr;            // users + 1
	const int *__restrict__ csr_idx;            // item ids
	const double *__restrict__ csr_val;
	const int *__restrict__ csc_ptr;            // items + 1
	const int *__restrict__ csc_idx;            // local user ids
	const double *__restrict__ csc_val;
	const double *__restrict__ L_in;            // current generation
	const double *__restrict__ R_in;
	double *__restrict__ L_out;                 // where the generation after `iters` iterations belongs
	double *__restrict__ R_out;
	int nnz;
};

constexpr size_t resident_lds_bytes(int users, int items, int K, long long nnz)
{
	return (size_t) 2 * (size_t) (users + items) * K * 8 + (size_t) nnz * 2 * 12 + (size_t) (users + items + 2) * 4 + 64;
}

// KMAX > 0: K <= KMAX and the owner keeps its old row and its accumulators in registers (loops fully unrolled, the
// K loads of a gathered row issued together: one LDS latency per entry instead of one per element); KMAX == 0: any K.
// The K <= 32 variant keeps three 32-double arrays (192 VGPRs) per thread: that fits the register file only with at
// most 256 threads per workgroup (512 VGPRs per lane and SIMD / one wave per SIMD), so it is bounded -- and chosen --
// for users + items <= 256; under a 1024-thread bound it spilled to scratch.
constexpr int resident_max_threads(int kmax) { return kmax == 32 ? 256 : 1024; }
template <int KMAX>
__global__ void __launch_bounds__(resident_max_threads(KMAX)) sweep_resident_kernel(ResidentArgs a)
{
	extern __shared__ __attribute__((aligned(16))) char rlds[];
	const int U = a.users, I = a.items, K = a.K, nnz = a.nnz;
	const int nf = (U + I) * K;                 // doubles per generation: L rows, then R rows
	double *gen0 = reinterpret_cast<double *>(rlds), *gen1 = gen0 + nf;
	double *val = gen1 + nf;                    // [csr values | csc values]
	int *idx = reinterpret_cast<int *>(val + 2 * (size_t) nnz);   // [csr idx | csc idx]
	int *ptr = idx + 2 * (size_t) nnz;          // [csr_ptr (U+1) | csc_ptr (I+1)]
	const int t = threadIdx.x, nt = blockDim.x;
	for (int x = t; x < U * K; x += nt) gen0[x] = a.L_in[(size_t) (x / K) * a.ldl + x % K];
	for (int x = t; x < I * K; x += nt) gen0[U * K + x] = a.R_in[(size_t) (x / K) * a.ldr + x % K];
	for (int x = t; x < nnz; x += nt) {
		val[x] = a.csr_val[x];
		val[nnz + x] = a.csc_val[x];
		idx[x] = a.csr_idx[x];
		idx[nnz + x] = a.csc_idx[x];
	}
	for (int x = t; x <= U; x += nt) ptr[x] = a.csr_ptr[x];
	for (int x = t; x <= I; x += nt) ptr[U + 1 + x] = a.csc_ptr[x];
	__syncthreads();

	// thread t < U: user row t against R (CSR); U <= t < U+I: item row t-U against L (CSC)
	const bool user = t < U, owner = t < U + I;
	const int r = user ? t : t - U;
	const int beg = owner ? (user ? ptr[r] : nnz + ptr[U + 1 + r]) : 0;
	const int end = owner ? (user ? ptr[r + 1] : nnz + ptr[U + 1 + r + 1]) : 0;
	const int xoff = (user ? r : U + r) * K;    // my row inside a generation
	const int ybase = user ? U * K : 0;         // the other factor inside a generation
	double *cur = gen0, *nxt = gen1;
	for (int it = 0; it < a.iters; ++it) {
		if (owner && KMAX > 0) {
			constexpr int KM = KMAX > 0 ? KMAX : 1;
			double xr[KM], acc[KM];
#pragma unroll
			for (int k = 0; k < KM; ++k) {
				xr[k] = k < K ? cur[xoff + k] : 0.0;
				acc[k] = xr[k];
			}
			for (int n = beg; n < end; ++n) {
				const double *y = cur + ybase + idx[n] * K;
				double yr[KM];
#pragma unroll
				for (int k = 0; k < KM; ++k) yr[k] = k < K ? y[k] : 0.0;
				double dot = 0.0;
#pragma unroll
				for (int k = 0; k < KM; ++k)
					if (k < K) dot = dot + xr[k] * yr[k];
				const double e = a.c2 * (val[n] - dot);
#pragma unroll
				for (int k = 0; k < KM; ++k)
					if (k < K) acc[k] = acc[k] + e * yr[k];
			}
#pragma unroll
			for (int k = 0; k < KM; ++k)
				if (k < K) nxt[xoff + k] = acc[k];
		} else if (owner) {
			const double *x = cur + xoff;
			double *xn = nxt + xoff;
			for (int k = 0; k < K; ++k) xn[k] = x[k];
			for (int n = beg; n < end; ++n) {
				const double *y = cur + ybase + idx[n] * K;
				double dot = 0.0;
				for (int k = 0; k < K; ++k) dot = dot + x[k] * y[k];
				const double e = a.c2 * (val[n] - dot);
				for (int k = 0; k < K; ++k) xn[k] = xn[k] + e * y[k];
			}
		}
		__syncthreads();
		double *sw = cur;
		cur = nxt;
		nxt = sw;
	}
	for (int x = t; x < U * K; x += nt) a.L_out[(size_t) (x / K) * a.ldl + x % K] = cur[x];
	for (int x = t; x < I * K; x += nt) a.R_out[(size_t) (x / K) * a.ldr + x % K] = cur[U * K + x];
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
	constexpr int P = G::kPieces, NP = G::kPasses, S = G::kStride;
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
		const double2 *__restrict__ xrow2 = reinterpret_cast<const double2 *>(a.X_old + (size_t) r * a.ldx);
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
						unsigned long long base = ybase + (unsigned long long) (unsigned) j * (unsigned long long) a.ldy * 8ull;
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
			double2 *__restrict__ out2 = reinterpret_cast<double2 *>(a.X_new + (size_t) r * a.ldx);
#pragma unroll
			for (int p = 0; p < NP; ++p) {
				const int q = lane + kWave * p;
				if (q < P) out2[q] = acc[p];
			}
		}
		__syncthreads();   // xs is rewritten for the next row
	}
}

}  // namespace mf
