// mf_collective.hip.h -- in-process exchange of the replicated factor between shards.
#pragma once
#include "mf_common.hip.h"

namespace mf {
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

// All N loads of an element pair are issued before the first add (the loop over the shards is unrolled to kMaxShards with
// a uniform predicate): a peer load over xGMI takes microseconds, N of them in a row would serialise the launch.
__global__ void __launch_bounds__(256) peer_allreduce_kernel(PeerReduceArgs a)
{
	const size_t stride = (size_t) gridDim.x * 256 * 2;
	const int n = a.nshards;
	for (size_t e = a.begin + ((size_t) blockIdx.x * 256 + threadIdx.x) * 2; e < a.end; e += stride) {
		if (e + 1 < a.end) {
			double2 w[kMaxShards];
#pragma unroll
			for (int h = 0; h < kMaxShards; ++h)
				if (h < n) w[h] = *reinterpret_cast<const double2 *>(a.buf[h] + e);
			double2 v = w[0];
#pragma unroll
			for (int h = 1; h < kMaxShards; ++h)
				if (h < n) {   // shard order: a fixed summation order, so the result is reproducible
					v.x = v.x + w[h].x;
					v.y = v.y + w[h].y;
				}
#pragma unroll
			for (int h = 0; h < kMaxShards; ++h)
				if (h < n) *reinterpret_cast<double2 *>(a.buf[h] + e) = v;
		} else {
			double v = a.buf[0][e];
			for (int h = 1; h < n; ++h) v = v + a.buf[h][e];
			for (int h = 0; h < n; ++h) a.buf[h][e] = v;
		}
	}
}

}  // namespace mf
