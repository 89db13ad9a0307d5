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
// stream_kernel: one wave per (row, slice of <= 16 columns).  A single wave pulls ~25 GB/s through LDS-DMA and retires
// one dependent v_add_f64 per ~4 ns, so a 128-byte slice of a row per entry keeps the gather just ahead of the chain;
// the slices of a row run on different CUs at the same time.  The row's entries are cut into chunks of 64; per chunk a
// META transfer (the chunk's indices and errors) and a GATHER (its 64 slices of rows of Y, eight rows per instruction)
// go global -> LDS by LDS-DMA into rings, META kStreamMetaAhead chunks ahead of its GATHER (the indices must be in
// registers before the gather can be issued: that latency is otherwise paid on every step), GATHER kStreamDepth chunks
// ahead of the chunk being added up.  Every transfer is inline asm and every wait a hand-counted `s_waitcnt vmcnt(N)`
// (hipcc would drain the ring with vmcnt(0) in front of every LDS read that may alias a pending LDS-DMA); transfers
// complete in issue order, so N = the number of transfers issued after the one waited for.  All ordinary loads are
// retired before the counted region starts.
// A wave does not own ONE row: the (row, slice)s are cut into chunks of at most 64 entries (a chunk never crosses a
// row), the chunk list is dealt to ~768 persistent waves in contiguous, cost-balanced ranges (a row is never split
// between waves), and a wave streams its range through the rings without ever draining them: the first chunk of a row
// brings the row's seed X_old along in its META, the last one stores X_new.  Short rows therefore cost their bytes,
// not three memory latencies each.
#pragma once
#include "mf_common.hip.h"
#include "mf_sweep.hip.h"

namespace mf {

// One entry of a side for the streams launch: the row of Y it gathers (fixed at plan time) and its error e_n (written
// by the errors launch every iteration), 16 bytes, so that ONE LDS-DMA instruction brings a chunk's 64 records.
struct StreamRec {
	int idx;
	int pad;
	double err;
};

struct StreamSide {
	const StreamRec *__restrict__ rec;  // per entry, in this side's entry order (CSC for items, CSR for users)
	const double *__restrict__ X_old;
	const double *__restrict__ Y_old;
	double *__restrict__ X_new;
};

// One chunk of the streams launch: entries [first, first + cnt) of a side's entry arrays, all of one (row, slice).
struct StreamChunk {
	int first;
	int cnt_flags;   // cnt (0..64; 0 only for a row without entries) | kStreamFirst | kStreamLast
	int trip;        // (side << 30) | (slice << 24) | row
	int pad;
};
constexpr int kStreamFirst = 1 << 8, kStreamLast = 1 << 9;

struct StreamArgs {
	int nwaves;
	int K;
	int sp;                                      // 16-byte pieces (column pairs) per slice, <= kStreamSlicePieces
	int dbg;                                     // timing experiments only (MF_ES_DBG): 1 no stores, 2 no adds, 4 no gathers, 8 drain
	unsigned *stamps;                            // MF_ES_DBG & 64: clock at 4 points of the first 16 steps of every wave
	const int *__restrict__ wave_beg;            // nwaves + 1 offsets into chunks
	const StreamChunk *__restrict__ chunks;
	StreamSide side[2];                          // 0: items (X = R, Y = L), 1: users (X = L, Y = R)
};

constexpr int kStreamChunk = 64;                    // entries per chunk: one lane per entry of the META transfer
constexpr int kStreamSlicePieces = 8;               // 16-byte pieces per (entry, slice): 128-byte tile rows
constexpr int kStreamDepth = 4;                     // GATHERs in flight ahead of the chunk being added
constexpr int kStreamMetaAhead = 5;                 // METAs in flight ahead of their GATHER
constexpr int kStreamTileSlots = kStreamDepth + 1;
constexpr int kStreamMetaSlots = kStreamMetaAhead + kStreamDepth + 2;
constexpr int kStreamMetaBytes = 1024 + 128;        // 64 records + the row's seed slice
constexpr int kStreamTileBytes = kStreamChunk * kStreamSlicePieces * 16;
constexpr int kStreamGather = kStreamChunk / (kWave / kStreamSlicePieces);   // DMA instructions per full GATHER (8 rows each)
constexpr int kStreamMeta = 2;                      // DMA instructions per META at most
constexpr size_t kStreamLdsBytes = (size_t) kStreamMetaSlots * kStreamMetaBytes + (size_t) kStreamTileSlots * kStreamTileBytes;
constexpr int kStreamWavesPerCu = 3;                // what the LDS footprint allows
static_assert(kStreamDepth * (kStreamGather + kStreamMeta) + kStreamMetaAhead * kStreamMeta <= 63, "vmcnt is a 6-bit counter");
static_assert(kStreamMetaSlots <= 16, "the transfer-count history keeps 16 entries");

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in 0..63.  The instruction only takes an immediate, and a C switch
// becomes a tree of a dozen scalar branches on this target (0.7 us per wait, measured): jump into a table of
// {s_waitcnt vmcnt(i); s_branch end} pairs instead.  s_getpc_b64 returns the address of the instruction behind it;
// the five 4-byte instructions up to and including s_setpc_b64 put the table at +20.
#define MF_W1(i) "s_waitcnt vmcnt(" #i ")\n\ts_branch 1f\n\t"
#define MF_W8(a, b, c, d, e, f, g, h) MF_W1(a) MF_W1(b) MF_W1(c) MF_W1(d) MF_W1(e) MF_W1(f) MF_W1(g) MF_W1(h)
__device__ __forceinline__ void wait_vmcnt_n(int n)
{
	asm volatile("s_getpc_b64 s[96:97]\n\t"
	             "s_lshl_b32 s95, %0, 3\n\t"
	             "s_add_u32 s95, s95, 20\n\t"
	             "s_add_u32 s96, s96, s95\n\t"
	             "s_addc_u32 s97, s97, 0\n\t"
	             "s_setpc_b64 s[96:97]\n\t"
	             MF_W8(0, 1, 2, 3, 4, 5, 6, 7) MF_W8(8, 9, 10, 11, 12, 13, 14, 15)
	             MF_W8(16, 17, 18, 19, 20, 21, 22, 23) MF_W8(24, 25, 26, 27, 28, 29, 30, 31)
	             MF_W8(32, 33, 34, 35, 36, 37, 38, 39) MF_W8(40, 41, 42, 43, 44, 45, 46, 47)
	             MF_W8(48, 49, 50, 51, 52, 53, 54, 55) MF_W8(56, 57, 58, 59, 60, 61, 62, 63)
	             "1:\n\t"
	             :
	             : "s"(__builtin_amdgcn_readfirstlane(n))
	             : "s95", "s96", "s97", "scc", "memory");
}
#undef MF_W8
#undef MF_W1

// one LDS-DMA instruction: every active lane moves 16 (4) bytes from its own global address to
// lds_dst + 16 (4) * lane (wave-uniform base in M0, written in the same statement that uses it)
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

typedef int mf_int4 __attribute__((ext_vector_type(4)));

// A chunk descriptor through the scalar cache.  It must not be an ordinary load: hipcc would fetch it with a vector
// load and put `s_waitcnt vmcnt(0)` in front of its use, draining the rings on every step.  Load and wait sit in ONE
// statement (hipcc takes an asm output for valid where the statement ends and may copy it at once); the descriptors
// are 16 bytes each and read in order, so three of four loads hit the scalar cache.
__device__ __forceinline__ mf_int4 sload16(const void *p)
{
	mf_int4 r;
	asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
	return r;
}

__global__ void __launch_bounds__(kWave) stream_kernel(StreamArgs a)
{
	constexpr int D = kStreamDepth, A = kStreamMetaAhead;
	const int K = a.K, P = K >> 1, sp = a.sp;
	extern __shared__ __attribute__((aligned(16))) char lds[];
	const unsigned lds_base = (unsigned) (unsigned long long) (__attribute__((address_space(3))) char *) lds;
	const unsigned tiles_base = lds_base + kStreamMetaSlots * kStreamMetaBytes;
	char *tiles = lds + kStreamMetaSlots * kStreamMetaBytes;
	const int lane = threadIdx.x;
	const int rr = lane >> 3, piece = lane & 7;   // GATHER: lane -> (row of the instruction, piece of the slice)
	const int cb = a.wave_beg[blockIdx.x], n = a.wave_beg[blockIdx.x + 1] - cb;
	const StreamChunk *__restrict__ chunks = a.chunks + cb;
	// both sides' arrays in scalar registers, selected per chunk
	const StreamRec *const rec0 = a.side[0].rec, *const rec1 = a.side[1].rec;
	const double *const xo0 = a.side[0].X_old, *const xo1 = a.side[1].X_old;
	const double *const yo0 = a.side[0].Y_old, *const yo1 = a.side[1].Y_old;
	double *const xn0 = a.side[0].X_new, *const xn1 = a.side[1].X_new;

	// Transfers issued so far, and that count right after the META / GATHER of the last 16 chunks, with the chunks'
	// descriptors beside them (lane = chunk & 15; read back with v_readlane): the wait for a transfer allows as many
	// outstanding ones as were issued after it.
	int tot = 0;
	int hist_meta = 0, hist_gather = 0, hist_first = 0, hist_cf = 0, hist_trip = 0;

	struct Where {   // what a chunk's descriptor says, all wave-uniform
		int first, cnt, flags, side, row, piece0, np;
	};
	auto decode = [&](int first, int cf, int trip) {
		Where w;
		w.first = first;
		w.cnt = cf & 255;
		w.flags = cf;
		w.side = (trip >> 30) & 1;
		w.row = trip & ((1 << 24) - 1);
		w.piece0 = ((trip >> 24) & 63) * sp;
		w.np = min(sp, P - w.piece0);
		return w;
	};
	auto where = [&](int i) {
		return decode(__builtin_amdgcn_readlane(hist_first, i & 15), __builtin_amdgcn_readlane(hist_cf, i & 15),
		              __builtin_amdgcn_readlane(hist_trip, i & 15));
	};
	// META(i): the chunk's 64 records (reads up to 63 entries past the chunk: the arrays carry 64 entries
	// of slack) and, for the first chunk of a row, the row's seed slice -> meta slot i % kStreamMetaSlots
	auto issue_meta = [&](int i, mf_int4 d) {
		const Where w = decode(d.x, d.y, d.z);
		const bool me = lane == (i & 15);
		hist_first = me ? d.x : hist_first;
		hist_cf = me ? d.y : hist_cf;
		hist_trip = me ? d.z : hist_trip;
		const unsigned dst = lds_base + (unsigned) (i % kStreamMetaSlots) * kStreamMetaBytes;
		if (w.cnt > 0) {
			dma16((w.side ? rec1 : rec0) + (size_t) w.first + lane, dst);
			tot += 1;
		}
		if (w.flags & kStreamFirst) {
			const int *x32 = reinterpret_cast<const int *>((w.side ? xo1 : xo0) + (size_t) w.row * K + 2 * w.piece0);
			if (lane < 4 * w.np) dma4(x32 + lane, dst + 1024);
			tot += 1;
		}
		hist_meta = me ? tot : hist_meta;
	};
	// GATHER(i): the slices of the chunk's rows of Y -> tile slot i % kStreamTileSlots, eight rows per instruction;
	// META(i) must have landed
	auto issue_gather = [&](int i) {
		const Where w = where(i);
		if (w.cnt > 0 && !(a.dbg & 4)) {
			const char *ybyte = reinterpret_cast<const char *>(w.side ? yo1 : yo0) + 16 * w.piece0 + 16 * piece;
			const int my_idx = *reinterpret_cast<const int *>(lds + (i % kStreamMetaSlots) * kStreamMetaBytes + 16 * min(lane, w.cnt - 1));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my_idx is here; LDS reads of the slot refilled below are done
			const unsigned tbase = tiles_base + (unsigned) (i % kStreamTileSlots) * kStreamTileBytes;
			const int ninstr = (w.cnt + 7) >> 3;
#pragma unroll
			for (int q = 0; q < kStreamGather; ++q) {
				if (q >= ninstr) break;
				const int j = __shfl(my_idx, 8 * q + rr);
				if (piece < w.np) dma16(ybyte + (size_t) (unsigned) j * (size_t) (K * 8), tbase + (unsigned) (q * 1024));
			}
			tot += ninstr;
		}
		hist_gather = lane == (i & 15) ? tot : hist_gather;
	};

	double acc = 0.0;
	unsigned stamp = 0;   // lane 4*step + point
	auto mark = [&](int step, int point) {
		if ((a.dbg & 64) && step < 16) {
			const unsigned now = (unsigned) __builtin_amdgcn_s_memtime();
			stamp = lane == 4 * step + point ? now : stamp;
		}
	};
	mark(0, 0);
	// every ordinary load above has landed before the hand-counted region starts
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
	for (int i = 0; i < min(A + 1, n); ++i) issue_meta(i, sload16(chunks + i));
	mark(0, 1);
	// step s issues META(s+A), then GATHER(s-1), then adds chunk s-1-D
	for (int s = 1; s <= n + D; ++s) {
		mark(s, 0);
		if (s + A < n) issue_meta(s + A, sload16(chunks + ((a.dbg & 32) ? 0 : s + A)));
		mark(s, 1);
		if (s <= n) {
			if (a.dbg & 16) wait_vmcnt<0>(); else
			wait_vmcnt_n(min(63, tot - __builtin_amdgcn_readlane(hist_meta, (s - 1) & 15)));
			mark(s, 2);
			issue_gather(s - 1);
		}
		mark(s, 3);
		const int c = s - 1 - D;
		if (c < 0) continue;
		if (a.dbg & 16) wait_vmcnt<0>(); else
		wait_vmcnt_n((a.dbg & 8) ? 0 : min(63, tot - __builtin_amdgcn_readlane(hist_gather, c & 15)));
		// ADD(c): acc = acc + e_n * y_n[col] for the chunk's entries in order -- the serial accumulation order.  Lane =
		// column of the slice; the errors are read from LDS at a wave-uniform address (broadcast).
		const Where w = where(c);
		const int ncol = 2 * w.np;
		const char *mb = lds + (c % kStreamMetaSlots) * kStreamMetaBytes;
		const int col = lane < ncol ? lane : 0;
		if (w.flags & kStreamFirst) acc = *reinterpret_cast<const double *>(mb + 1024 + 8 * col);
		const char *eb = mb + 8;   // record e: {idx, pad, err}
		const char *tb = tiles + (c % kStreamTileSlots) * kStreamTileBytes + 8 * col;
		int e = 0;
		if (a.dbg & 2) e = w.cnt;
		for (; e + 8 <= w.cnt; e += 8) {
			double ev[8], tv[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) ev[u] = *reinterpret_cast<const double *>(eb + 16 * (e + u));
#pragma unroll
			for (int u = 0; u < 8; ++u) tv[u] = *reinterpret_cast<const double *>(tb + (e + u) * 128);
#pragma unroll
			for (int u = 0; u < 8; ++u) acc = acc + ev[u] * tv[u];
		}
		for (; e < w.cnt; ++e) {
			const double en = *reinterpret_cast<const double *>(eb + 16 * e);
			acc = acc + en * *reinterpret_cast<const double *>(tb + e * 128);
		}
		if ((w.flags & kStreamLast) && lane < ncol && !(a.dbg & 1))
			(w.side ? xn1 : xn0)[(size_t) w.row * K + 2 * w.piece0 + lane] = acc;
	}
	if (a.dbg & 64) {
		mark(15, 3 * (n + D >= 15 ? 0 : 1) + 0);
		a.stamps[(size_t) blockIdx.x * 64 + lane] = lane == 63 ? (unsigned) __builtin_amdgcn_s_memtime() : stamp;
	}
}

}  // namespace mf
