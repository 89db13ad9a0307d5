// mf_resident.hip.h -- the streams launch for SMALL factor matrices (MovieLens-100k: 943 x 30 and 1682 x 30): a column
// slice of the whole of Y lives in the LDS of a workgroup, so the ordered accumulation gathers nothing from memory.
//
// Gathering every entry's row of Y from memory is what bounds the sweeps and any other streams launch on such an
// instance (a CU lands about one 1-KiB LDS-DMA transfer per 100-180 cycles; instML100k gathers 51 MB per iteration).
// But the same few hundred KB of Y are gathered over and over: every row of R is used by ~60 users.  Here a
// workgroup (one per CU) first copies an SW-column slice of ALL rows of Y into LDS -- coalesced, once -- and then its
// waves add up slices of X:  X_new[r][c] = (...((X_old[r][c] + e_0*Y[j_0][c]) + e_1*Y[j_1][c]) + ...) with the y values
// read from LDS by index.  The only streamed data are the {idx, e_n} records (16 bytes per entry).
//
// A wave owns a run of CONSECUTIVE rows of its side: their records are one contiguous stretch of the record array and
// their row pointers and seeds X_old are contiguous too, so everything the wave needs is addressed from the workgroup's
// descriptor alone -- two memory latencies from launch to the first add, however many rows follow.  The stretch is
// streamed in chunks of 64 records (prefetched ahead in four registers; two chunks per fill of the step pipeline), 64 / SW
// entries per step: lane (g, c) forms the
// product of entry step + g for column c and parks it in LDS; then every lane adds the step's products for its column
// in entry order, closing a row (store X_new, take the next row's seed) wherever its last entry falls.  Same rounded
// products, same order of adds as matFact.c:41-53: bit-identical.  The chain of dependent adds (10 cycles each) is the
// critical path; the products of the next entries are formed beside it.
#pragma once
#include "mf_common.hip.h"
#include <type_traits>
#include "mf_stream.hip.h"
#ifndef MF_ES_ABLATE
#define MF_ES_ABLATE 0   // diagnostic builds only (results wrong, timing valid): 1 no read-back, 2 no y gather, 3 no add chain
#endif

namespace mf {

constexpr int kResidentWaves = 8;
constexpr int kResidentThreads = kResidentWaves * kWave;
constexpr int kResidentWaveLds = 4608;    // per wave: 128 + 3 x 32 records (padding of three steps at SW = 2) + 2 x 64 products
constexpr int kResidentRows = 63;          // rows a wave owns at most: its row pointers sit in one register
constexpr int kResidentCopyPieces = 20;   // 16-byte pieces of the Y slice a thread copies per round (20 x 512 x 16 B = 160 KB)

struct SliceWg {
	int side;                            // 0: items (X = R, Y = L), 1: users (X = L, Y = R)
	int slice;                           // columns [slice * SW, slice * SW + SW)
	int row_beg[kResidentWaves + 1];     // wave w owns rows [row_beg[w], row_beg[w+1]) of the side
	int ent_beg[kResidentWaves + 1];     // = ptr[row_beg[w]]: its records [ent_beg[w], ent_beg[w+1])
};

struct SliceArgs {
	int K;
	const SliceWg *__restrict__ wg;
	const int *__restrict__ ptr[2];   // row pointers per side (CSC for items, CSR for users)
	StreamSide side[2];
	int yrows[2];                     // rows of Y per side
	int ldx[2], ldy[2];               // row pitch of X and of Y per side, in doubles (>= K)
};

#ifdef MF_STAMPS
// diagnostic build only (tools/es_stamps.py): per workgroup and wave -- [0] s_memrealtime at entry (100 MHz), [1] shader
// cycles until the slice is in LDS (after the barrier), [2] shader cycles until the wave's last row is stored, [3]
// s_memrealtime at exit, [4] entries of the wave, [5] rows of the wave, [6] shader cycles in pipeline fills, [7] in runs of
// steps inside a row, [8] those steps, [9] cycles in steps with row bookkeeping, [10] those steps, [11] fills
__device__ unsigned long long mf_es_stamp_buf[512 * kResidentWaves * 12];
#endif

template <int SW>
__global__ void __launch_bounds__(kResidentThreads) stream_resident_kernel(SliceArgs a)
{
#ifdef MF_STAMPS
	const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime(), st_clk0 = __builtin_amdgcn_s_memtime();
#endif
	static_assert(SW == 8 || SW == 4 || SW == 2, "slice width in columns");
	constexpr int G = kWave / SW;          // entries per step = lane groups per wave
	extern __shared__ __attribute__((aligned(16))) char lds[];
	const SliceWg &me = a.wg[blockIdx.x];
	const int K = a.K;
	const int side = me.side;
	const StreamSide sd = a.side[side];
	const int *__restrict__ ptr = a.ptr[side];
	const int yrows = a.yrows[side];
	const size_t ldx = (size_t) a.ldx[side], ldy = (size_t) a.ldy[side];
	const int col0 = me.slice * SW, ncol = min(SW, K - col0);
	const double *ys = reinterpret_cast<const double *>(lds);
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	char *wbuf = lds + (size_t) yrows * SW * 8 + (size_t) wave * kResidentWaveLds;
	StreamRec *recbuf = reinterpret_cast<StreamRec *>(wbuf);
	double *prod = reinterpret_cast<double *>(wbuf + kResidentWaveLds - 1024);   // two steps' products
	const int g = lane / SW, c = lane - g * SW;
	const int cc = c < ncol ? c : 0;

	const int rb = me.row_beg[wave], re = me.row_beg[wave + 1];
	const int eb = me.ent_beg[wave], ee = me.ent_beg[wave + 1];
	// ---- everything this wave needs is addressed from the descriptor: issue it all before the slice copy
	// records of chunk [pos, pos + 64), one per lane; reads up to 63 entries past the stretch (64 entries of slack)
	// (always a load, never a branch: past the end of the stretch it re-reads the last chunk's address)
	const int last_chunk = ee > eb ? eb + ((ee - eb - 1) / 64) * 64 : eb;
	auto ld = [&](int pos) { return sd.rec[(size_t) min(pos, last_chunk) + lane]; };
	StreamRec r0 = ld(eb), r1 = ld(eb + 64), r2 = ld(eb + 128), r3 = ld(eb + 192);
	// row pointers of the wave's rows (at most kResidentRows of them): lane l holds ptr[rb + l]
	const int pv = ptr[min(rb + lane, re)];
	// seeds of all its rows: register b, lane (g, c) holds X_old[rb + b * G + g][col0 + c].  Nothing but the record
	// prefetch and the X_new stores touches memory once the stream runs: a load inside it would make hipcc drain the
	// prefetched chunks (s_waitcnt vmcnt(0) at every join of its branch).
	constexpr int NSEED = kWave / G;
	double sv[NSEED];
#pragma unroll
	for (int b = 0; b < NSEED; ++b)
		sv[b] = rb + b * G + g < re ? sd.X_old[(size_t) (rb + b * G + g) * ldx + col0 + cc] : 0.0;

	// ---- the slice of every row of Y -> LDS (16-byte pieces, SW/2 per row).  All of a thread's loads are issued
	// before the first LDS write (one memory latency for the whole copy, not one per piece): at most kPieces per thread.
	{
		constexpr int PP = SW / 2;
		const int total = yrows * PP;
		double2 *ys2 = reinterpret_cast<double2 *>(lds);
		for (int base = 0; base < total; base += kResidentThreads * kResidentCopyPieces) {
			double2 v[kResidentCopyPieces];
#pragma unroll
			for (int k = 0; k < kResidentCopyPieces; ++k) {
				const int i = base + k * kResidentThreads + tid;
				const int r = i / PP, pc = i - r * PP;
				v[k] = make_double2(0.0, 0.0);
				if (i < total && 2 * pc < ncol) v[k] = *reinterpret_cast<const double2 *>(sd.Y_old + (size_t) r * ldy + col0 + 2 * pc);
			}
#pragma unroll
			for (int k = 0; k < kResidentCopyPieces; ++k) {
				const int i = base + k * kResidentThreads + tid;
				if (i < total) ys2[i] = v[k];
			}
		}
	}
	recbuf[128 + lane] = StreamRec{0, 0, 0.0};   // padding behind the chunk pair (read three steps ahead, never used)
	if (lane < 32) recbuf[192 + lane] = StreamRec{0, 0, 0.0};
	__syncthreads();
#ifdef MF_STAMPS
	unsigned long long st_fill = 0, st_in = 0, st_nin = 0, st_edge = 0, st_nedge = 0, st_nfill = 0;
	const unsigned long long st_clk1 = __builtin_amdgcn_s_memtime();
	auto stamp_out = [&]() {
		if (lane == 0 && blockIdx.x < 512) {
			unsigned long long *o = mf_es_stamp_buf + ((size_t) blockIdx.x * kResidentWaves + wave) * 12;
			o[6] = st_fill; o[7] = st_in; o[8] = st_nin; o[9] = st_edge; o[10] = st_nedge; o[11] = st_nfill;
			o[0] = st_real0;
			o[1] = st_clk1 - st_clk0;
			o[2] = __builtin_amdgcn_s_memtime() - st_clk0;
			o[3] = __builtin_amdgcn_s_memrealtime();
			o[4] = (unsigned long long) (ee - eb);
			o[5] = (unsigned long long) (re - rb);
		}
	};
	if (rb >= re) stamp_out();
#endif
	if (rb >= re) return;

	// ---- the stream.  cur = the row being added up, row_end = its last entry + 1, acc = its running sum.
	int cur = rb;
	auto ptr_of = [&](int q) { return __builtin_amdgcn_readlane(pv, q - rb); };   // ptr[q], rb <= q <= re
	auto seed_of = [&](int q) {    // X_old[q][col0 + c] in every lane group
		const int b = (q - rb) / G, gg = (q - rb) - b * G;
		double x = sv[0];
#pragma unroll
		for (int k = 1; k < NSEED; ++k) x = b == k ? sv[k] : x;
		return __shfl(x, gg * SW + cc);
	};
	double acc = seed_of(rb);
	int row_end = ptr_of(cur + 1);
	// A finished row is stored from an asm statement: a store hipcc knows of inside the stream makes it drain the
	// prefetched chunks at every chunk (vmcnt(0) at the loop head).  Unknown to its bookkeeping the store only makes
	// its counted waits stricter (memory operations retire in issue order).
	auto close_rows_at = [&](int pos) {   // rows whose last entry is pos - 1 (and empty rows behind them) are complete
		while (pos == row_end && cur < re) {
			if (g == 0 && c < ncol) {
				double *dst = sd.X_new + (size_t) cur * ldx + col0 + c;
				asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(acc) : "memory");
			}
			++cur;
			if (cur < re) {
				acc = seed_of(cur);
				row_end = ptr_of(cur + 1);
			}
		}
	};
	close_rows_at(eb);   // leading rows without entries
	// One chunk of <= 64 records, G entries per step, software-pipelined over the steps: while the products of step s
	// are being added (the chain of dependent adds), the records of step s+2 and the y values of step s+1 are already
	// on their way from LDS and the products of step s+1 are written to the other half of `prod` and read back.  A wave's LDS
	// accesses execute in program order, so the write of a step's products precedes their reads without a barrier.
	// Records past the end of the chunk are stored with idx = 0 (a valid row of the slice) and recbuf carries 3 G
	// records of padding, so the steps need neither clamps nor validity selects; runs of steps that lie inside the
	// current row go through a loop without any row bookkeeping.
	// (TWO chunks per call: the pipeline is filled once per 128 records -- five dependent LDS round trips, ~700 cycles,
	// which at one chunk per call were a third of a long run's time: tools/es_stamps.py)
	auto process = [&](StreamRec chunk, StreamRec chunk_b, int c0) {
		const int cnt = min(128, ee - c0), nsteps = (cnt + G - 1) / G;
#ifdef MF_STAMPS
		const unsigned long long st_p0 = __builtin_amdgcn_s_memtime();
#endif
		if (lane >= cnt) chunk.idx = 0;
		if (lane + 64 >= cnt) chunk_b.idx = 0;
		recbuf[lane] = chunk;
		recbuf[64 + lane] = chunk_b;
		__builtin_amdgcn_wave_barrier();
		StreamRec my1 = recbuf[g];                                   // step 0
		prod[lane] = my1.err * ys[(size_t) my1.idx * SW + c];
		my1 = recbuf[G + g];                                         // step 1
		double y1 = ys[(size_t) my1.idx * SW + c];
		StreamRec my2 = recbuf[2 * G + g];                           // step 2
		// the products of the step being added are in REGISTERS (pr[PAR]), read back from LDS during the step before:
		// the chain of dependent adds never waits for LDS.  Two register sets swap roles from step to step.  Every
		// LDS result is consumed one full step after its read was issued: records three steps ahead, y values two,
		// products one.
		double pr[2][G];
#pragma unroll
		for (int u = 0; u < G; ++u) pr[0][u] = prod[u * SW + c];
		int sidx = 0;
#ifdef MF_STAMPS
		asm volatile("" : "+v"(pr[0][0]));
		st_fill += __builtin_amdgcn_s_memtime() - st_p0;
		++st_nfill;
#endif
		auto step = [&](auto par_c, bool inside) {
			constexpr int PAR = decltype(par_c)::value;
			const int s0 = sidx * G;
			const StreamRec my3 = recbuf[s0 + 3 * G + g];            // records of step sidx + 3 (padding past the chunk)
#if MF_ES_ABLATE == 2
			const double y2 = (double) my2.idx;                      // ablation: no gather of y from the LDS slice
#else
			const double y2 = ys[(size_t) my2.idx * SW + c];         // y of step sidx + 2
#endif
			prod[(PAR ^ 1) * kWave + lane] = my1.err * y1;           // products of step sidx + 1 ...
			// ... back into registers (every lane group reads them: masking the read-back to the one group that stores
			// the row was measured 6 % slower -- the LDS cost of an instruction does not shrink with its active lanes)
#if MF_ES_ABLATE == 1
#pragma unroll
			for (int u = 0; u < G; ++u) pr[PAR ^ 1][u] = my1.err * (double) u;   // ablation: no read-back of the products
#else
#pragma unroll
			for (int u = 0; u < G; ++u) pr[PAR ^ 1][u] = prod[(PAR ^ 1) * kWave + u * SW + c];
#endif
			if (inside) {
#if MF_ES_ABLATE == 3
				acc = acc + ((pr[PAR][0] + pr[PAR][1]) + (pr[PAR][G - 2] + pr[PAR][G - 1]));   // ablation: no chain of G dependent adds
#else
#pragma unroll
				for (int u = 0; u < G; ++u) acc = acc + pr[PAR][u];
#endif
			} else {
				const int m = min(G, cnt - s0), pos0 = c0 + s0;
#pragma unroll
				for (int u = 0; u < G; ++u)
					if (u < m) {
						acc = acc + pr[PAR][u];
						close_rows_at(pos0 + u + 1);
					}
			}
			my1 = my2;
			y1 = y2;
			my2 = my3;
			__builtin_amdgcn_wave_barrier();
			++sidx;
		};
		using P0 = std::integral_constant<int, 0>;
		using P1 = std::integral_constant<int, 1>;
		while (sidx < nsteps) {
			// full steps from here that end before the current row does: no row bookkeeping in their loop
			int nf = min((cnt - sidx * G) / G, (row_end - 1 - (c0 + sidx * G)) / G);
#ifdef MF_STAMPS
			const unsigned long long st_r0 = __builtin_amdgcn_s_memtime();
			const int st_s0 = sidx;
#endif
			if ((sidx & 1) && nf > 0) {
				step(P1{}, true);
				--nf;
			}
			if (!(sidx & 1))
				for (; nf >= 2; nf -= 2) {
					step(P0{}, true);
					step(P1{}, true);
				}
			if (nf > 0) step(P0{}, true);
#ifdef MF_STAMPS
			const unsigned long long st_r1 = __builtin_amdgcn_s_memtime();
			st_in += st_r1 - st_r0;
			st_nin += (unsigned long long) (sidx - st_s0);
#endif
			if (sidx < nsteps) {
				if (sidx & 1)
					step(P1{}, false);
				else
					step(P0{}, false);
#ifdef MF_STAMPS
				st_edge += __builtin_amdgcn_s_memtime() - st_r1;
				++st_nedge;
#endif
			}
		}
	};
	// four chunks per trip, two per call, each register pair refilled right after its chunks are consumed: no register is ever
	// copied, so a pair is waited for with exactly two younger loads in flight (unconditional loads: hipcc counts them)
	for (int c0 = eb; c0 < ee; c0 += 256) {
		process(r0, r1, c0);
		r0 = ld(c0 + 256);
		r1 = ld(c0 + 320);
		if (c0 + 128 < ee) process(r2, r3, c0 + 128);
		r2 = ld(c0 + 384);
		r3 = ld(c0 + 448);
	}
	// trailing rows without entries (and the last row when the stretch is empty)
	row_end = ee;
	close_rows_at(ee);
#ifdef MF_STAMPS
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	stamp_out();
#endif
}

}  // namespace mf
