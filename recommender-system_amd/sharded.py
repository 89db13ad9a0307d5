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


def all_gather_arrays(arr, group=None):
    """All-gather of one numpy array per rank (same dtype, different lengths along axis 0) with tensor collectives
    only -- lengths first, then the byte images padded to the longest -- so nothing is pickled and, on RCCL, the
    payload moves GPU to GPU (MPI_Gatherv, matFact-mpi.c:135).  Returns the list of arrays in rank order."""
    arr = np.ascontiguousarray(arr)
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    raw = torch.from_numpy(arr.reshape(-1).view(np.uint8).copy())
    n = torch.tensor([raw.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(t.item()) for t in sizes]
    width = max(max(sizes), 1)
    mine = torch.zeros(width, dtype=torch.uint8, device=dev)
    mine[:raw.numel()] = raw.to(dev)
    parts = [torch.empty(width, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    tail = arr.shape[1:]
    return [parts[g][:sizes[g]].cpu().numpy().view(arr.dtype).reshape((-1,) + tail) for g in range(world)]


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
            p.iterate(1)      # the library's own iteration (item sweep, user sweep, flip): what the CLI calls
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
        parts = all_gather_arrays(mine.numpy(), group=self.group)
        out = np.empty(users_total, np.int32)
        for g in range(self.world):
            out[begin[g]:begin[g + 1]] = parts[g]
        return out


# ------------------------------------------------------------------------------------------------------------
# 2-D process grid: the full decomposition of matFact-mpi.c:155-214.  Rank (gr, gc) of a
# rows x cols grid (create_balanced_grid, mpiutil.c:54-88 -> capi.balanced_grid; rank = gr*cols + gc, the
# row-major order of MPI_Cart_create) owns the TILE user block gr x item block gc: its entries, a copy of the
# L block (shared by the ranks of grid row gr) and of the R block (shared by the ranks of grid column gc).
# Per iteration (matFact-mpi.c:185-209)
#     item sweep : R_next = (gr == 0 ? R_cur : 0) + sum over the tile's entries     :187, :190-205
#     all-reduce : R_next over the grid COLUMN (ranks with the same gc)             :208  (col_comm)
#     user sweep : L_next = (gc == 0 ? L_cur : 0) + sum over the tile's entries     :188
#     all-reduce : L_next over the grid ROW (ranks with the same gr)                :207  (row_comm)
#     flip
# With cols == 1 this is the row-sharded scheme above; with rows == 1 its mirror image (item-heavy inputs).
# Recommendation: every tile scans its own item block (mf_plan_recommend_scored), the partial scan states are
# combined left to right over the item blocks -- the MPI_Reduce(max_cmp) of matFact-mpi.c:98, with the serial
# program's tie and NaN rules (matFact.c:13-23) instead of the MPI variant's.
# ------------------------------------------------------------------------------------------------------------
def grid_coords(rank, grid):
    return rank // grid[1], rank % grid[1]


def block_bounds(n, parts):
    """BLOCK_LOW boundaries (mpiutil.h:8) of n indices over `parts` blocks."""
    return np.array([(p * n) // parts for p in range(parts + 1)], np.int64)


def make_grid_groups(grid, rank, **new_group_kwargs):
    """(row_group, col_group) of this rank; every rank must call it (dist.new_group is collective).
    row_group joins the ranks of one grid row (they share an L block), col_group those of one grid column."""
    rows, cols = grid
    row_group = col_group = None
    for gr in range(rows):
        ranks = [gr * cols + c for c in range(cols)]
        g = dist.new_group(ranks, **new_group_kwargs) if cols > 1 else None
        if rank in ranks:
            row_group = g
    for gc in range(cols):
        ranks = [r * cols + gc for r in range(rows)]
        g = dist.new_group(ranks, **new_group_kwargs) if rows > 1 else None
        if rank in ranks:
            col_group = g
    return row_group, col_group


def merge_candidates(left, right):
    """Combine the partial scan states (capi.CANDIDATE_DTYPE records, GLOBAL item ids) of two item ranges,
    `left` holding the lower item ids: what the sequential scan of matFact.c:13-23 would hold after both."""
    out = left.copy()
    no_first = left["first"] < 0
    out["first"][no_first] = right["first"][no_first]
    out["first_nan"][no_first] = right["first_nan"][no_first]
    take = (right["best"] >= 0) & ((left["best"] < 0) | (right["score"] > left["score"]))
    out["best"][take] = right["best"][take]
    out["score"][take] = right["score"][take]
    return out


def certify_filters(filters, norm, rmax, margin):
    """Certification of the matrix-core pass over the item blocks of one grid row (mf_filter records per block, in
    item order, arg already GLOBAL).  Returns (answer, certain): for certain users `answer` is the reference's
    arg-max (-1: nothing unrated); the others must be re-scored exactly.  A score differs from the reference's by
    less than half the margin, so the winning block's best must beat its own runner-up AND every other block's best
    by margin * ||L[i]|| * max ||R[j]||; any non-finite score, or a non-finite norm, certifies nobody."""
    best = np.stack([f["best"] for f in filters])          # blocks x users
    w = np.argmax(best, axis=0)
    cols = np.arange(best.shape[1])
    b1 = best[w, cols]
    others = best.copy()
    others[w, cols] = -np.inf
    runner = np.maximum(np.stack([f["second"] for f in filters])[w, cols], others.max(axis=0))
    arg = np.stack([f["arg"] for f in filters])[w, cols]
    bad = np.stack([f["nonfinite"] for f in filters]).any(axis=0)
    thr = margin * norm * rmax + 1e-300
    with np.errstate(invalid="ignore"):
        gap_ok = (b1 - runner) > thr                        # False for NaN thresholds and for -inf - -inf
    none = np.stack([f["arg"] for f in filters]).max(axis=0) < 0
    certain = ~bad & (none | ((arg >= 0) & gap_ok))
    return np.where(none, -1, arg).astype(np.int32), certain


def finish_candidates(c):
    """best[i] of print_output: -1 without an unrated item; the first unrated item when its score is NaN
    (nothing compares greater than NaN); else the arg-max over the non-NaN scores."""
    return np.where(c["first"] < 0, -1, np.where(c["first_nan"] != 0, c["first"], c["best"])).astype(np.int32)


class GridFactorization:
    """Iterations of one tile of the rows x cols grid; both factor generations live in torch tensors."""

    def __init__(self, plan, l_buffers, r_buffers, rank, grid, row_group=None, col_group=None, overlap=True,
                 stream=None):
        self.plan, self.rank, self.grid, self.overlap = plan, rank, tuple(grid), overlap
        self.gr, self.gc = grid_coords(rank, grid)
        self.row_group, self.col_group = row_group, col_group
        self.l, self.r = list(l_buffers), list(r_buffers)
        self._ptr = {int(t.data_ptr()): t for t in self.l + self.r}
        self.stream = stream
        if stream is not None:
            if int(stream.cuda_stream) == 0:
                raise ValueError("GridFactorization needs a non-default CUDA stream")
            plan.set_stream(int(stream.cuda_stream))
        elif self.r[0].is_cuda:
            raise ValueError("GPU buffers need an explicit stream")
        if self.grid[1] > 1 and row_group is None or self.grid[0] > 1 and col_group is None:
            raise ValueError("a %dx%d grid needs its row and column process groups" % self.grid)

    def current_items(self):
        return self._ptr[int(self.plan.items_current_ptr())]

    def current_users(self):
        return self._ptr[int(self.plan.users_current_ptr())]

    def step(self):
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._step()
        else:
            self._step()

    def _step(self):
        p = self.plan
        rows, cols = self.grid
        pending = []
        p.sweep_items(self.gr == 0)
        if rows > 1:
            nxt = self._ptr[int(p.items_next_ptr())]
            pending.append(dist.all_reduce(nxt, op=dist.ReduceOp.SUM, group=self.col_group, async_op=True))
            if not self.overlap:
                pending.pop().wait()
        p.sweep_users(self.gc == 0)
        if cols > 1:
            nxt = self._ptr[int(p.users_next_ptr())]
            pending.append(dist.all_reduce(nxt, op=dist.ReduceOp.SUM, group=self.row_group, async_op=True))
        for w in pending:
            w.wait()
        p.flip()

    def run(self, iters):
        for _ in range(iters):
            self.step()

    def gather_recommendations(self, users_total, user_begin, item_begin, use_filter=True):
        """Every tile runs the matrix-core pass over its item block; a user is settled when the winning block's
        best beats every other candidate of the grid row by the rounding margin (certify_filters).  The others are
        re-scored exactly in every tile of the row and the partial scan states merged in item order -- the
        MPI_Reduce(max_cmp) of matFact-mpi.c:98 with the serial program's tie and NaN rules.  Every rank returns
        the full list (one int per user)."""
        rows, cols = self.grid
        world = rows * cols
        j0 = int(item_begin[self.gc])
        margin = None
        if use_filter and hasattr(self.plan, "recommend_filter"):
            from . import capi
            filt, norm, rmax = self.plan.recommend_filter()
            filt["arg"][filt["arg"] >= 0] += j0
            margin = capi.recommend_margin(self.plan.feats)
            mine = (filt, norm, rmax)
        else:
            mine = None
        parts = [mine]
        if world > 1 and mine is not None:
            # three tensor all-gathers (filter records, ||L[i]||, max ||R[j]||) instead of pickled objects
            fs = all_gather_arrays(filt)
            ns = all_gather_arrays(norm)
            rs = all_gather_arrays(np.array([rmax], np.float64))
            parts = [(fs[g], ns[g], float(rs[g][0])) for g in range(world)]
        elif world > 1:
            parts = [None] * world
        out = np.full(users_total, -1, np.int32)
        uncertain = {}                      # grid row -> local user ids that need the exact pass
        for gr in range(rows):
            n = int(user_begin[gr + 1] - user_begin[gr])
            if mine is None:
                uncertain[gr] = np.arange(n, dtype=np.int32)
                continue
            blocks = [parts[gr * cols + gc] for gc in range(cols)]
            # np.max propagates a NaN norm (Python's max would drop it unless it came first): thr is then NaN and
            # nobody is certified
            ans, certain = certify_filters([b[0] for b in blocks], blocks[0][1],
                                           float(np.max(np.array([b[2] for b in blocks], np.float64))), margin)
            out[user_begin[gr]:user_begin[gr + 1]] = ans
            uncertain[gr] = np.flatnonzero(~certain).astype(np.int32)
        self.last_uncertain = int(sum(len(v) for v in uncertain.values()))
        # exact pass for the uncertain users of MY grid row, in my item block; merged over the row in item order
        todo = uncertain[self.gr]
        if hasattr(self.plan, "recommend_scored_users"):
            cand = self.plan.recommend_scored_users(todo)
        else:
            cand = self.plan.recommend_scored()[todo]
        for f in ("best", "first"):
            cand[f][cand[f] >= 0] += j0
        cparts = [cand]
        if world > 1:
            cparts = all_gather_arrays(cand)
        for gr in range(rows):
            if len(uncertain[gr]) == 0:
                continue
            acc = cparts[gr * cols]
            for gc in range(1, cols):
                acc = merge_candidates(acc, cparts[gr * cols + gc])
            out[user_begin[gr] + uncertain[gr]] = finish_candidates(acc)
        return out
