"""The one-process-per-GPU protocol (recommender-system_amd/sharded.py) with world_size 2 on CPU over gloo.
The arithmetic of each shard is supplied by an oracle-backed stand-in plan (tests may use the oracle); what is
under test is the host logic: partition, seed-on-root rule, all-reduce of the item factor, flip, gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, random_instance


class OraclePlan:
    """Same surface as capi.Plan, computed by oracle.shard_step into CPU torch buffers."""

    def __init__(self, O, inst, begin, rank, L_block, R, r_bufs):
        self.O, self.inst, self.u0, self.uc = O, inst, int(begin[rank]), int(begin[rank + 1] - begin[rank])
        sel = (inst.row >= begin[rank]) & (inst.row < begin[rank + 1])
        self.row, self.col, self.val = (np.ascontiguousarray(a[sel]) for a in (inst.row, inst.col, inst.val))
        self.L = [L_block.copy(), np.empty_like(L_block)]
        self.r = r_bufs
        self.r[0].copy_(torch.from_numpy(R))
        self.cur = 0
        self._pending_L = None

    def _step(self, seed):
        return self.O.shard_step(self.u0, self.uc, self.inst.items, self.inst.feats, self.row, self.col, self.val,
                                 self.inst.alpha, self.L[self.cur], self.r[self.cur].numpy().copy(), seed)

    def sweep_items(self, seed_from_old=True):
        Ln, Ra = self._step(seed_from_old)
        self._pending_L = Ln
        self.r[self.cur ^ 1].copy_(torch.from_numpy(Ra))

    def sweep_users(self):
        self.L[self.cur ^ 1] = self._pending_L

    def items_next_ptr(self):
        return self.r[self.cur ^ 1].data_ptr()

    def items_current_ptr(self):
        return self.r[self.cur].data_ptr()

    def flip(self):
        self.cur ^= 1

    def recommend(self):
        sub = self.O.Instance(self.inst.iters, self.inst.alpha, self.inst.feats, self.uc, self.inst.items,
                              self.row - self.u0, self.col, self.val)
        return self.O.recommend(sub, self.L[self.cur], self.r[self.cur].numpy().copy())


def _worker(rank, world, port, d, iters, overlap, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import recommender_system_amd as rs
    from oracle import oracle as O
    inst = O.Instance(**d)
    ptr = np.concatenate([[0], np.cumsum(np.bincount(inst.row, minlength=inst.users))]).astype(np.int64)
    begin = rs.sharded.shard_bounds(inst.users, world, ptr)
    L, R = rs.capi.init_factors(inst.users, inst.items, inst.feats)
    r_bufs = [torch.empty(inst.items, inst.feats, dtype=torch.float64) for _ in range(2)]
    plan = OraclePlan(O, inst, begin, rank, np.ascontiguousarray(L[begin[rank]:begin[rank + 1]]), R, r_bufs)
    run = rs.sharded.ShardedFactorization(plan, r_bufs, rank, world, overlap=overlap)
    run.run(iters)
    best = run.gather_recommendations(inst.users, begin)
    Lparts = [None] * world
    dist.all_gather_object(Lparts, plan.L[plan.cur])
    if rank == 0:
        np.savez(out_path, L=np.concatenate(Lparts), R=run.current_items().numpy(), best=best, begin=begin)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world, overlap", [(2, True), (2, False), (8, True)])
def test_sharded_run_matches_serial(orc, tmp_path, world, overlap):
    """world 2, and world 8 -- the size of the driver's scaling run (matFact-mpi.c's 8x1 grid for cfg4): eight ranks, user
    blocks balanced by entries (one of them may be a single user), seed-on-root, eight-way all-reduce, gather of the lists."""
    import recommender_system_amd as rs  # noqa: F401  (registers rs.sharded)
    import importlib
    importlib.import_module("recommender_system_amd.sharded")
    d = random_instance(21, 37, 23, 6, density=0.3, iters=9, alpha=0.004, empty_rows=(4,), full_rows=(7,))
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), d, d["iters"], overlap, out), nprocs=world, join=True)
    got = np.load(out)
    inst = orc.Instance(**d)
    L, R = orc.init_factors(inst.users, inst.items, inst.feats)
    orc.factorize(inst, L, R)
    # sharding re-associates the sums into R (the partial sums of the shards are added): north-star tolerance 1e-5 relative
    assert np.allclose(got["L"], L, rtol=1e-9, atol=1e-13)
    assert np.allclose(got["R"], R, rtol=1e-9, atol=1e-13)
    assert np.array_equal(got["best"], orc.recommend(inst, L, R))
    assert got["begin"][0] == 0 and got["begin"][-1] == inst.users


# ------------------------------------------------------------------------------------------------- 2-D grid
def scan_state(scores, rated, j0=0):
    """The partial state of print_output's scan (matFact.c:13-23) over one item range, in plain Python."""
    best, score, first, fnan = -1, 0.0, -1, 0
    for j, s in enumerate(scores):
        if rated[j]:
            continue
        if first < 0:
            first, fnan = j0 + j, int(s != s)
        if s == s and (best < 0 or s > score):
            best, score = j0 + j, s
    return score, best, first, fnan, 0


class OracleTilePlan:
    """capi.Plan's surface for one tile of the grid, computed by oracle.tile_step into CPU torch buffers."""

    def __init__(self, O, cand_dtype, inst, ub, ib, gr, gc, L_block, R_block, l_bufs, r_bufs):
        self.O, self.inst, self.cand_dtype = O, inst, cand_dtype
        self.u0, self.uc = int(ub[gr]), int(ub[gr + 1] - ub[gr])
        self.j0, self.ic = int(ib[gc]), int(ib[gc + 1] - ib[gc])
        sel = ((inst.row >= ub[gr]) & (inst.row < ub[gr + 1]) & (inst.col >= ib[gc]) & (inst.col < ib[gc + 1]))
        self.row, self.col, self.val = (np.ascontiguousarray(a[sel]) for a in (inst.row, inst.col, inst.val))
        self.l, self.r = l_bufs, r_bufs
        self.l[0].copy_(torch.from_numpy(L_block))
        self.r[0].copy_(torch.from_numpy(R_block))
        self.cur = 0
        self._aux = None

    def _tile(self, l_root, r_root):
        return self.O.tile_step(self.u0, self.uc, self.j0, self.ic, self.inst.feats, self.row, self.col, self.val,
                                self.inst.alpha, self.l[self.cur].numpy().copy(), self.r[self.cur].numpy().copy(),
                                l_root, r_root)

    def sweep_items(self, seed_from_old=True):
        self.r[self.cur ^ 1].copy_(torch.from_numpy(self._tile(True, seed_from_old)[1]))

    def sweep_users(self, seed_from_old=True):
        self.l[self.cur ^ 1].copy_(torch.from_numpy(self._tile(seed_from_old, True)[0]))

    def items_next_ptr(self):
        return self.r[self.cur ^ 1].data_ptr()

    def items_current_ptr(self):
        return self.r[self.cur].data_ptr()

    def users_next_ptr(self):
        return self.l[self.cur ^ 1].data_ptr()

    def users_current_ptr(self):
        return self.l[self.cur].data_ptr()

    def flip(self):
        self.cur ^= 1

    def recommend_scored(self):
        L, R = self.l[self.cur].numpy(), self.r[self.cur].numpy().copy()
        out = np.zeros(self.uc, self.cand_dtype)
        for i in range(self.uc):
            rated = np.zeros(self.ic, bool)
            rated[self.col[self.row == self.u0 + i] - self.j0] = True
            out[i] = scan_state(self.O.predict_row(L[i], R), rated)
        return out


def _grid_worker(rank, world, port, d, grid, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import recommender_system_amd as rs
    from oracle import oracle as O
    sh = rs.sharded
    inst = O.Instance(**d)
    gr, gc = sh.grid_coords(rank, grid)
    ptr = np.concatenate([[0], np.cumsum(np.bincount(inst.row, minlength=inst.users))]).astype(np.int64)
    ub = sh.shard_bounds(inst.users, grid[0], ptr)
    ib = sh.block_bounds(inst.items, grid[1])
    L, R = rs.capi.init_factors(inst.users, inst.items, inst.feats)
    l_bufs = [torch.empty(int(ub[gr + 1] - ub[gr]), inst.feats, dtype=torch.float64) for _ in range(2)]
    r_bufs = [torch.empty(int(ib[gc + 1] - ib[gc]), inst.feats, dtype=torch.float64) for _ in range(2)]
    plan = OracleTilePlan(O, rs.capi.CANDIDATE_DTYPE, inst, ub, ib, gr, gc,
                          np.ascontiguousarray(L[ub[gr]:ub[gr + 1]]), np.ascontiguousarray(R[ib[gc]:ib[gc + 1]]),
                          l_bufs, r_bufs)
    row_group, col_group = sh.make_grid_groups(grid, rank)
    run = sh.GridFactorization(plan, l_bufs, r_bufs, rank, grid, row_group, col_group)
    run.run(inst.iters)
    best = run.gather_recommendations(inst.users, ub, ib)
    parts = [None] * world
    dist.all_gather_object(parts, (run.current_users().numpy(), run.current_items().numpy()))
    if rank == 0:
        Lf = np.concatenate([parts[r * grid[1]][0] for r in range(grid[0])])
        Rf = np.concatenate([parts[c][1] for c in range(grid[1])])
        # every rank of a grid row must hold the same L block, every rank of a grid column the same R block
        same = all(np.array_equal(parts[r * grid[1] + c][0], parts[r * grid[1]][0]) and
                   np.array_equal(parts[r * grid[1] + c][1], parts[c][1])
                   for r in range(grid[0]) for c in range(grid[1]))
        np.savez(out_path, L=Lf, R=Rf, best=best, same=same)
    dist.destroy_process_group()


@pytest.mark.parametrize("grid", [(2, 2), (1, 2), (3, 1)])
def test_grid_run_matches_serial(orc, tmp_path, grid):
    """matFact-mpi.c:185-209 on a rows x cols grid over gloo: factors to re-association accuracy, same top-1."""
    importlib = __import__("importlib")
    importlib.import_module("recommender_system_amd.sharded")
    d = random_instance(23, 26, 31, 5, density=0.35, iters=7, alpha=0.004, empty_rows=(3,), full_rows=(9,))
    out = str(tmp_path / "grid.npz")
    world = grid[0] * grid[1]
    mp.spawn(_grid_worker, args=(world, _free_port(), d, grid, out), nprocs=world, join=True)
    got = np.load(out)
    inst = orc.Instance(**d)
    L, R = orc.init_factors(inst.users, inst.items, inst.feats)
    orc.factorize(inst, L, R)
    assert bool(got["same"])
    assert np.allclose(got["L"], L, rtol=1e-9, atol=1e-13)
    assert np.allclose(got["R"], R, rtol=1e-9, atol=1e-13)
    assert np.array_equal(got["best"], orc.recommend(inst, L, R))


def test_candidate_merge_equals_the_sequential_scan():
    """Cutting the item range anywhere and merging the partial states gives the scan over the whole range:
    ties keep the lower index, NaN never wins unless it is the first unrated item, -inf and all-rated blocks."""
    import recommender_system_amd as rs
    sh = __import__("importlib").import_module("recommender_system_amd.sharded")
    rng = np.random.default_rng(5)
    pool = np.array([np.nan, -np.inf, np.inf, 0.0, -0.0, 1.5, 1.5, 2.0, -3.0])
    for trial in range(300):
        n = int(rng.integers(1, 13))
        scores = rng.choice(pool, size=n)
        rated = rng.random(n) < 0.35
        if trial % 17 == 0:
            rated[:] = True
        whole = np.zeros(1, rs.capi.CANDIDATE_DTYPE)
        whole[0] = scan_state(scores, rated)
        cuts = sorted(set(rng.integers(0, n + 1, size=int(rng.integers(1, 4))).tolist()) | {0, n})
        acc = None
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = np.zeros(1, rs.capi.CANDIDATE_DTYPE)
            part[0] = scan_state(scores[a:b], rated[a:b], j0=a)
            acc = part if acc is None else sh.merge_candidates(acc, part)
        assert sh.finish_candidates(acc)[0] == sh.finish_candidates(whole)[0], (scores, rated, cuts)
        # and the serial program's own rule, restated directly
        want, cur = -1, None
        for j in range(n):
            if rated[j]:
                continue
            if cur is None or scores[j] > cur:
                want, cur = j, scores[j]
        assert sh.finish_candidates(acc)[0] == want


def test_balanced_grid_rule():
    """create_balanced_grid (mpiutil.c:54-88), checked by hand against its text (it needs MPI to compile):
    most square factorisation stretched towards the matrix's aspect ratio, long side along the long dimension."""
    import recommender_system_amd as rs
    g = rs.capi.balanced_grid
    assert g(1000000, 100000, 8) == (8, 1)      # cfg4: ratio 10 -> 4x2 -> 8x1
    assert g(200000, 200000, 8) == (4, 2)       # cfg5: square matrix keeps MPI_Dims_create's 4x2
    assert g(400, 50000, 8) == (1, 8)           # inst400-50000: item-heavy -> swapped
    assert g(943, 1682, 4) == (2, 2)            # ML100k: ratio 1
    assert g(100, 1000, 12) == (3, 4)           # ratio 10, limit 10: 4x3 cannot stretch to 12 -> swapped
    assert g(3000, 1000, 8) == (4, 2)           # ratio 3: 8 > limit 3, stays 4x2
    assert g(100, 100, 7) == (7, 1) and g(5, 5, 1) == (1, 1)
    for n in range(1, 40):
        r, c = g(1000, 30, n)
        assert r * c == n and r >= c


def test_filter_certification_over_item_blocks():
    """certify_filters: a user is settled only when the winning block's best beats its own runner-up and every
    other block's best by the margin; ties across blocks, near-ties, NaN/inf scores, NaN norms and fully rated
    users must all come back uncertain or exactly right."""
    import recommender_system_amd as rs
    sh = __import__("importlib").import_module("recommender_system_amd.sharded")
    rng = np.random.default_rng(11)
    users, items, cuts = 200, 40, [0, 13, 14, 40]
    S = rng.standard_normal((users, items))
    rated = rng.random((users, items)) < 0.3
    rated[3] = True                                  # nothing unrated at all
    rated[4, :14] = True                             # whole blocks rated
    S[5, 20] = S[5, 2] = S[5].max() + 1.0            # exact tie across blocks
    S[6, 30] = S[6].max() + 2.0
    S[6, 1] = S[6, 30] - 1e-13                       # near-tie below the margin
    S[7, 9] = np.nan
    S[8, 25] = np.inf
    rated[5, [2, 20]] = rated[6, [1, 30]] = rated[7, 9] = rated[8, 25] = False
    norm = np.ones(users)
    norm[9] = np.nan
    filters = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        f = np.zeros(users, rs.capi.FILTER_DTYPE)
        for i in range(users):
            v = [(S[i, j], j) for j in range(a, b) if not rated[i, j]]
            fin = sorted([x for x in v if np.isfinite(x[0])], key=lambda x: (-x[0], x[1]))
            f[i] = (fin[0][0] if fin else -np.inf, fin[1][0] if len(fin) > 1 else -np.inf,
                    fin[0][1] if fin else -1, int(any(not np.isfinite(x[0]) for x in v)))
        filters.append(f)
    ans, certain = sh.certify_filters(filters, norm, 1.0, 1e-9)
    for i in range(users):
        want = scan_state(S[i], rated[i])
        want = -1 if want[2] < 0 else (want[2] if want[3] else want[1])
        if certain[i]:
            assert ans[i] == want, i
    assert certain[3] and ans[3] == -1
    assert not certain[5] and not certain[6] and not certain[7] and not certain[8] and not certain[9]
    assert certain.sum() > users * 0.9


def test_certify_filters_nan_rmax_in_a_later_block_certifies_nobody():
    """the largest max||R[j]|| of a grid row must PROPAGATE a NaN wherever it sits (Python's max drops a
    NaN unless it comes first); with a NaN threshold nobody who has a candidate is certified."""
    import recommender_system_amd as rs
    sh = __import__("importlib").import_module("recommender_system_amd.sharded")
    users = 5
    filters = []
    for b in range(3):
        f = np.zeros(users, rs.capi.FILTER_DTYPE)
        f["best"], f["second"], f["arg"] = 10.0 * (b + 1), 0.0, 7 * b
        filters.append(f)
    rmaxes = [1.0, float("nan"), 2.0]
    assert max(rmaxes) == 2.0                                    # the trap
    rmax = float(np.max(np.array(rmaxes)))
    assert rmax != rmax
    ans, certain = sh.certify_filters(filters, np.ones(users), rmax, 1e-9)
    assert not certain.any()
    ans, certain = sh.certify_filters(filters, np.ones(users), 2.0, 1e-9)
    assert certain.all() and (ans == 14).all()


def _gather_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import recommender_system_amd as rs
    mine = np.zeros(3 * rank, rs.capi.CANDIDATE_DTYPE)          # rank 0 contributes an EMPTY array
    mine["best"] = np.arange(3 * rank) + 100 * rank
    mine["score"] = rank + 0.5
    parts = rs.sharded.all_gather_arrays(mine)
    mat = rs.sharded.all_gather_arrays(np.full((rank + 1, 4), float(rank)))
    ok = all(p.dtype == rs.capi.CANDIDATE_DTYPE and len(p) == 3 * g and (p["score"] == g + 0.5).all() and
             (p["best"] == np.arange(3 * g) + 100 * g).all() for g, p in enumerate(parts))
    ok = ok and all(m.shape == (g + 1, 4) and (m == g).all() for g, m in enumerate(mat))
    if rank == world - 1:
        np.save(out_path, np.array([ok]))
    dist.destroy_process_group()


def test_all_gather_arrays_ragged_structured(tmp_path):
    """The tensor-collective replacement of all_gather_object: ragged lengths (one rank empty), record dtypes and
    2-D arrays come back in rank order."""
    out = str(tmp_path / "ok.npy")
    mp.spawn(_gather_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    assert bool(np.load(out)[0])


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (child torch.distributed.run,
    before any GPU call).  Without a GPU here each rank stops at the no-device check -- the message names the rank
    and the world size, which proves the launcher ran; on the GPU box the same command is run by the gpu tests."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0", "--config", "twin", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=600, env=env)
    import recommender_system_amd as rs
    if rs.capi.device_count() > 0:
        assert r.returncode == 0, r.stderr[-2000:]
        out = __import__("json").loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    else:
        assert r.returncode != 0
        assert "rank 0/2 needs an MI355X" in r.stderr and "rank 1/2 needs an MI355X" in r.stderr, r.stderr[-2000:]
