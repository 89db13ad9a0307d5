// mf_stream.hip.h -- the "errors + streams" form of one iteration for instances whose factors are small enough to be
// latency-bound, not bandwidth-bound (the reference's own samples, MovieLens-100k).
//
// matFact.c:41-53 computes ONE error per entry, e_n = (alpha*2)*(a_n - dot(Ls[i], Rs[j])), and uses it for both
// updates.  dot is the same bits whichever side forms it (the products x[k]*y[k] commute, the sum runs over k in the
// same order), so an iteration splits into
//   E  entry-parallel: e_n for every entry -- the ERRORS mode of sweep_dma_kernel over <= 64-entry SEGMENTS of the
//      CSR rows (one wave per segment, thousands of them: no wave walks a long row), stored into the records of both
//      sides (CSR order and CSC order);
//   S  row-parallel, BOTH factors in one launch: X_new[r] = (...((X_old[r] + e_0*y_0) + e_1*y_1) + ...) in file order
//      -- stream_resident_kernel (mf_resident.hip.h): no dot products left, only the chain of dependent adds the
//      serial order prescribes.
// Two launches per iteration like the two sweeps, but the longest row costs one dependent add per entry instead of a
// gather -> dots -> accumulate round trip per 16-entry chunk (~150 ns per entry when few rows leave nothing to hide
// the latency behind).  Same rounded products, same order of adds: results stay bit-identical to matFact.c.
//
// (A streams launch that pulled every entry's slice of a Y row through LDS-DMA rings -- persistent waves, hand-counted
// vmcnt -- was built and measured first: a CU lands only one 1-KiB LDS-DMA transfer per 100-180 cycles whatever the
// rings hold in flight, so it was slower than the two sweeps on the cfg3 shape (0.70 vs 0.33 ms) and only level with
// the resident form on instML100k; it is not in the tree.)
#pragma once
#include "mf_common.hip.h"

namespace mf {

// One entry of a side for the streams launch: the row of Y it gathers (fixed at plan time) and its error e_n (written
// by the errors launch every iteration), 16 bytes, contiguous per row.
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

}  // namespace mf
