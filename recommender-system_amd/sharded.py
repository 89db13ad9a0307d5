"""One-process-per-GPU driver of the row-sharded iteration (the MPI decomposition of matFact-mpi.c on RCCL).

Decomposition (matFact-mpi.c:155-214 with the 8x1 grid create_balanced_grid picks for users >> items,
mpiutil.c:54-88): users are cut into `world` contiguous blocks; rank g holds its block of L (private: the
row communicator has size 1) and a full replica of R.  Per iteration
    item sweep : R_next = (rank == 0 ? R_cur : 0) + sum over LOCAL entries      matFact-mpi.c:187, :190-205
    all-reduce : R_next <- SUM over ranks (RCCL over xGMI; `MPI_Iallreduce`)    matFact-mpi.c:208
    user sweep : L_next from the frozen R_cur, overlapped with the collective   (L needs no communication)
    flip
The recommendation step needs no collective: every rank scores its own users against its R replica
(the reference reduces partial maxima over the row communicator, matFact-mpi.c:98, because its R is split).

The arithmetic runs in whatever `plan` is given: the HIP plan (capi.Plan) in production; tests pass an
oracle-backed stand-in so that the protocol is exercised with gloo on CPU.  This module never touches oracle/.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(users, world, row_ptr=None, by_entries=True):
    """User block boundaries begin[0..world] (by entry count when a CSR row pointer is available)."""
    from . import capi
    return capi.partition_users(users, world, row_ptr if by_entries else None)


class ShardedFactorization:
    """Runs iterations of one shard and keeps the two R generations in torch tensors for the collective."""

    def __init__(self, plan, r_buffers, rank, world, group=None, overlap=True, stream=None, force_collective=False):
        """`stream`: a NON-default torch.cuda.Stream for GPU runs.  The plan's kernels and the collective must be
        ordered on the same stream; the default stream's handle is 0, which mf_plan_set_stream reads as "use the
        plan's own stream", so it cannot be used here.  None only for CPU stand-in plans (tests)."""
        self.plan, self.rank, self.world, self.group, self.overlap = plan, rank, world, group, overlap
        self.r = list(r_buffers)          # two tensors, items x K, same device as the plan
        self._ptr = {int(t.data_ptr()): t for t in self.r}
        self.force_collective = force_collective
        self.stream = stream
        if stream is not None:
            if int(stream.cuda_stream) == 0:
                raise ValueError("ShardedFactorization needs a non-default CUDA stream")
            plan.set_stream(int(stream.cuda_stream))
        elif self.r[0].is_cuda:
            raise ValueError("GPU buffers need an explicit stream")

    def _next_tensor(self):
        return self._ptr[int(self.plan.items_next_ptr())]

    def current_items(self):
        return self._ptr[int(self.plan.items_current_ptr())]

    def step(self):
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._step()
        else:
            self._step()

    def _step(self):
        p = self.plan
        if self.world == 1 and not self.force_collective:
            p.sweep_items(True)
            p.sweep_users()
            p.flip()
            return
        p.sweep_items(self.rank == 0)
        nxt = self._next_tensor()
        if self.overlap:
            work = dist.all_reduce(nxt, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            p.sweep_users()
            work.wait()
        else:
            dist.all_reduce(nxt, op=dist.ReduceOp.SUM, group=self.group)
            p.sweep_users()
        p.flip()

    def run(self, iters):
        for _ in range(iters):
            self.step()

    def gather_recommendations(self, users_total, begin):
        """Every rank scores its own users; rank 0 receives the concatenation (MPI_Gatherv, matFact-mpi.c:135)."""
        mine = torch.from_numpy(np.ascontiguousarray(self.plan.recommend(), dtype=np.int32))
        if self.world == 1:
            return mine.numpy()
        parts = [None] * self.world
        dist.all_gather_object(parts, mine.numpy(), group=self.group)
        out = np.empty(users_total, np.int32)
        for g in range(self.world):
            out[begin[g]:begin[g + 1]] = parts[g]
        return out
