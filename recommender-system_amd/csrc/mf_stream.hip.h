// mf_stream.hip.h -- the "errors + streams" form of one iteration for instances whose factors live in L2 / Infinity
// Cache (the reference's own samples, MovieLens-sized data): latency-bound, not bandwidth-bound.
//
// matFact.c:41-53 computes ONE error per entry, e_n = (alpha*2)*(a_n - dot(Ls[i], Rs[j])), and uses it for both
// updates.  dot is the same bits whichever side forms it (the products x[k]*y[k] commute, the sum runs over k in the
// same order), so an iteration splits into
//   E  entry-parallel: e_n for every entry -- the ERRORS mode of sweep_dma_kernel over <= 64-entry SEGMENTS of the
//      CSR rows (one wave per segment, thousands of them: no wave walks a long row), stored in CSR and CSC order;
//   S  row-parallel, BOTH factors in one launch: X_new[r] = (...((X_old[r] + e_0*y_0) + e_1*y_1) + ...) in file order
//      -- stream_kernel below: no dot products left, only the chain of dependent adds the serial order prescribes.
// Two launches per iteration like the two sweeps, but the longest row costs one add per entry (~4-7 ns) instead of a
// gather -> dots -> accumulate round trip per 16-entry chunk (~150 ns per entry when few rows leave nothing to hide
// the latency behind).  Same rounded products, same order of adds: results stay bit-identical to matFact.c.
//
// stream_kernel: one wave per row.  The row's entries are cut into chunks of NCH; per chunk a META transfer (the
// chunk's indices and errors) and a GATHER (its NCH rows of Y) go global -> LDS by LDS-DMA into rings of D+2 and D+1
// slots, issued D chunks ahead of the chunk being added up.  Every transfer is inline asm with hand-counted
// `s_waitcnt vmcnt(N)` (hipcc would drain the ring with vmcnt(0) in front of every LDS read that may alias a pending
// LDS-DMA); all ordinary loads are retired before the counted region starts, stores only happen after it.
#pragma once
#include "mf_common.hip.h"
#include "mf_sweep.hip.h"

namespace mf {

struct StreamSide {
	const int *__restrict__ ptr;        // row pointers of this side (CSC for the items, CSR for the users)
	const int *__restrict__ idx;        // row of Y per entry
	const double *__restrict__ err;     // e_n per entry, in this side's entry order
	const double *__restrict__ X_old;
	const double *__restrict__ Y_old;
	double *__restrict__ X_new;
};

struct StreamArgs {
	int ntasks;
	int K;
	const int *__restrict__ tasks;      // (side << 30) | row, longest rows first
	StreamSide side[2];                 // 0: items (X = R, Y = L), 1: users (X = L, Y = R)
};

constexpr int kStreamDepth = 3;                     // chunks in flight ahead of the one being added
constexpr int kStreamTileSlots = kStreamDepth + 1;
constexpr int kStreamMetaSlots = kStreamDepth + 2;
constexpr int kStreamMetaBytes = 256 + 512;         // 64 indices + 64 errors per slot

// chunk size and DMA instruction counts per chunk (compile-time: the vmcnt immediates depend on them)
template <int KT, int NPASS>
struct StreamGeom {
	static constexpr int kPieces = KT / 2;                                   // 0 for run-time K
	static constexpr int kPs = kPieces | 1;
	static constexpr bool kMultiRow = KT > 0 && kPs <= 32;                  // several rows per DMA instruction
	static constexpr int kRpi = kMultiRow ? kWave / kPs : 1;                 // rows per instruction
	static constexpr int kNch = kMultiRow ? (kRpi >= 4 ? 64 : 32) : (16 / NPASS > 0 ? 16 / NPASS : 1);
	static constexpr int kGather = kMultiRow ? (kNch + kRpi - 1) / kRpi : kNch * NPASS;
	static constexpr int kMeta = kNch > 32 ? 3 : 2;
	static_assert(kStreamDepth * (kGather + kMeta) <= 63, "vmcnt is a 6-bit counter");
	static_assert(kNch <= 64, "one lane per entry of a chunk");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one LDS-DMA instruction: every active lane moves BYTES (4 or 16) from its own global address to
// lds_dst + 16|4 * lane (wave-uniform base in M0, written in the same statement that uses it)
__device__ __forceinline__ void dma16(const void *g, unsigned lds_dst)
{
	asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g),
	             "s"(__builtin_amdgcn_readfirstlane(lds_dst))
	             : "memory");
}
__device__ __forceinline__ void dma4(const void *g, unsigned lds_dst)
{
	asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g),
	             "s"(__builtin_amdgcn_readfirstlane(lds_dst))
	             : "memory");
}

inline size_t stream_lds_bytes(int K, int nch)
{
	return (size_t) kStreamMetaSlots * kStreamMetaBytes + (size_t) kStreamTileSlots * nch * 16 * ((K / 2) | 1);
}

template <int KT, int NPASS>
__global__ void __launch_bounds__(kWave) stream_kernel(StreamArgs a)
{
	using G = StreamGeom<KT, NPASS>;
	constexpr int NCH = G::kNch, NG = G::kGather, NM = G::kMeta, D = kStreamDepth, NP = NPASS;
	const int K = KT > 0 ? KT : a.K;
	const int P = K >> 1;
	const int S = 16 * (P | 1);
	extern __shared__ __attribute__((aligned(16))) char lds[];
	const unsigned lds_base = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) lds;
	const int tile_bytes = NCH * S;
	const unsigned tiles_base = lds_base + kStreamMetaSlots * kStreamMetaBytes;
	char *tiles = lds + kStreamMetaSlots * kStreamMetaBytes;
	const int lane = threadIdx.x;

	for (int t = blockIdx.x; t < a.ntasks; t += gridDim.x) {
		const int task = __builtin_amdgcn_readfirstlane(a.tasks[t]);
		const StreamSide sd = a.side[task >> 30];
		const int r = task & ((1 << 30) - 1);
		const int beg = sd.ptr[r], end = sd.ptr[r + 1];
		const int nc = (end - beg + NCH - 1) / NCH;
		const unsigned long long ybase = (unsigned long long) sd.Y_old;

		double2 acc[NP];
		{
			const double2 *__restrict__ xrow2 = reinterpret_cast<const double2 *>(sd.X_old + (size_t) r * K);
#pragma unroll
			for (int p = 0; p < NP; ++p) {
				const int q = lane + kWave * p;
				acc[p] = q < P ? xrow2[q] : make_double2(0.0, 0.0);
			}
		}
		// every ordinary load above has landed before the hand-counted region starts
		asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

		// META(c): 64 indices and 64 (NCH <= 32: 32) errors starting at the chunk's first entry -> meta slot c % (D+2).
		// Reads up to 63 entries past the chunk: the arrays carry 64 entries of slack.
		auto issue_meta = [&](int c) {
			const size_t first = (size_t) beg + (size_t) c * NCH;
			const unsigned slot = lds_base + (unsigned) (c % kStreamMetaSlots) * kStreamMetaBytes;
			dma4(sd.idx + first + lane, slot);
			const int *e32 = reinterpret_cast<const int *>(sd.err + first);
			dma4(e32 + lane, slot + 256);
			if (NM == 3) dma4(e32 + 64 + lane, slot + 512);
		};
		// GATHER(c): the chunk's NCH rows of Y -> tile slot c % (D+1); META(c) must have landed.  Always NG
		// instructions (the vmcnt arithmetic needs a fixed count): entries past the end of the row re-gather its last row.
		auto issue_gather = [&](int c) {
			const int cnt = min(NCH, end - (beg + c * NCH));
			const int my_idx = *reinterpret_cast<const int *>(lds + (c % kStreamMetaSlots) * kStreamMetaBytes +
			                                                  4 * min(lane, cnt - 1));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my_idx is here; LDS reads of the slot refilled below are done
			const unsigned tbase = tiles_base + (unsigned) (c % kStreamTileSlots) * (unsigned) tile_bytes;
			if constexpr (G::kMultiRow) {
				constexpr int PP = G::kPieces, PS = G::kPs, RPI = G::kRpi;
				const int rr = lane / PS, piece = lane - rr * PS;
#pragma unroll
				for (int n0 = 0; n0 < NCH; n0 += RPI) {
					const int n = n0 + rr;
					const int j = __shfl(my_idx, n < NCH ? n : 0);
					const char *src = reinterpret_cast<const char *>(ybase) + (size_t) (unsigned) j * (size_t) (KT * 8) + 16 * piece;
					if (rr < RPI && piece < PP && n < NCH) dma16(src, tbase + (unsigned) (n0 * S));
				}
			} else {
#pragma unroll 4
				for (int n = 0; n < NCH; ++n) {
					const int j = __builtin_amdgcn_readlane(my_idx, n);
					const char *row = reinterpret_cast<const char *>(ybase) + (size_t) (unsigned) j * (size_t) (K * 8) + 16 * lane;
#pragma unroll
					for (int p = 0; p < NP; ++p) {
						if (lane + kWave * p < P)
							dma16(row + 1024 * p, tbase + (unsigned) (n * S + 1024 * p));
						else if (P <= kWave * p && lane == 0)
							dma16(row, tbase + (unsigned) (n * S));   // a pass with no piece left (run-time K): keep the count, re-copy piece 0
					}
				}
			}
		};
		// ADD(c): acc[k] = acc[k] + e_n * y_n[k] for the chunk's entries in order (the serial accumulation order)
		auto add_chunk = [&](int c) {
			const int cnt = min(NCH, end - (beg + c * NCH));
			const double e = *reinterpret_cast<const double *>(lds + (c % kStreamMetaSlots) * kStreamMetaBytes + 256 +
			                                                   8 * (lane < NCH ? lane : 0));
			const char *tb = tiles + (c % kStreamTileSlots) * tile_bytes + 16 * lane;
			int n = 0;
			for (; n + 4 <= cnt; n += 4) {
				double2 tt[4][NP];
				double en[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					en[u] = readlane_f64(e, n + u);
#pragma unroll
					for (int p = 0; p < NP; ++p)
						tt[u][p] = (lane + kWave * p < P) ? *reinterpret_cast<const double2 *>(tb + (n + u) * S + 1024 * p)
						                                  : make_double2(0.0, 0.0);
				}
#pragma unroll
				for (int u = 0; u < 4; ++u)
#pragma unroll
					for (int p = 0; p < NP; ++p) {
						acc[p].x = acc[p].x + en[u] * tt[u][p].x;
						acc[p].y = acc[p].y + en[u] * tt[u][p].y;
					}
			}
			for (; n < cnt; ++n) {
				const double en = readlane_f64(e, n);
#pragma unroll
				for (int p = 0; p < NP; ++p)
					if (lane + kWave * p < P) {
						const double2 tv = *reinterpret_cast<const double2 *>(tb + n * S + 1024 * p);
						acc[p].x = acc[p].x + en * tv.x;
						acc[p].y = acc[p].y + en * tv.y;
					}
			}
		};

		if (nc <= D + 1) {
			// short row: everything in flight at once, one latency for the indices and one for the rows
			for (int c = 0; c < nc; ++c) issue_meta(c);
			wait_vmcnt<0>();
			for (int c = 0; c < nc; ++c) issue_gather(c);
			wait_vmcnt<0>();
			for (int c = 0; c < nc; ++c) add_chunk(c);
		} else {
			// step s issues META(s), then GATHER(s-1), then adds chunk s-1-D; the immediates count the transfers
			// issued AFTER the one waited for (transfers complete in issue order)
			issue_meta(0);
			issue_meta(1);
			wait_vmcnt<NM>();
			issue_gather(0);
			for (int s = 2; s <= D; ++s) {
				issue_meta(s);
				wait_vmcnt<NG + NM>();
				issue_gather(s - 1);
			}
			for (int s = D + 1; s < nc; ++s) {
				issue_meta(s);
				wait_vmcnt<NG + NM>();
				issue_gather(s - 1);
				wait_vmcnt<D *(NG + NM)>();
				add_chunk(s - 1 - D);
			}
			wait_vmcnt<NG>();
			issue_gather(nc - 1);
			wait_vmcnt<D * NG + (D - 1) * NM>();
			add_chunk(nc - 1 - D);
			static_assert(D == 3, "the drain below is written out for three chunks in flight");
			wait_vmcnt<2 * NG + NM>();
			add_chunk(nc - 3);
			wait_vmcnt<NG>();
			add_chunk(nc - 2);
			wait_vmcnt<0>();
			add_chunk(nc - 1);
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rings are reused by the next row
		double2 *__restrict__ out2 = reinterpret_cast<double2 *>(sd.X_new + (size_t) r * K);
#pragma unroll
		for (int p = 0; p < NP; ++p) {
			const int q = lane + kWave * p;
			if (q < P) out2[q] = acc[p];
		}
	}
}

}  // namespace mf
