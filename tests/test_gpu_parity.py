"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs, against the committed golden fixtures (the reference's own outputs), and --
at sizes the oracle cannot finish -- through exact spot checks and size-independent properties.

Bar: factor matrices BIT-EXACT (the kernels keep the serial summation order; north-star tolerance is 1e-5
relative, we hold 0), recommendations bit-identical."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, golden_in, random_instance, to_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(capi):
    if capi.device_count() < 1:
        pytest.fail("GPU tests need an MI355X; mf_backend_device_count() = %d" % capi.device_count())


@pytest.fixture(params=["auto", "sweeps", "sweeps-db"], autouse=True)
def iter_mode(request, monkeypatch):
    """Every test runs three times: with the iteration form the plan picks by itself (errors + streams for cache-resident
    factors, mf_stream.hip.h), with the two sweeps forced (the single-buffered single-wave kernel, or whatever the plan's
    schedule puts beside it) and with the two sweeps in the intra-wave double-buffered form (MF_SWEEP_DB=1: a kernel the plan
    never picks by itself, kept selectable) -- so that all of them stay pinned on the oracle."""
    if request.param != "auto":
        monkeypatch.setenv("MF_ITER_MODE", "sweeps")
    if request.param == "sweeps-db":
        monkeypatch.setenv("MF_SWEEP_DB", "1")
    return request.param


def _inst(capi, d):
    return capi.Instance(d["iters"], d["alpha"], d["feats"], d["users"], d["items"], d["row"], d["col"], d["val"])


def _oracle_run(orc, d, iters=None):
    inst = orc.Instance(**d)
    L, R = orc.init_factors(inst.users, inst.items, inst.feats)
    orc.factorize(inst, L, R, iters=iters)
    return L, R, orc.recommend(inst, L, R)


# ------------------------------------------------------------------ golden fixtures (reference's own outputs)
@pytest.mark.parametrize("name", ["inst0", "inst1", "inst2", "inst30-40-10-2-10", "inst1000-1000-100-2-30",
                                  "instML100k"])
def test_golden_full_run(capi, name):
    inst = capi.parse_file(golden_in(name))
    L, R = capi.init_factors(inst.users, inst.items, inst.feats)
    best = capi.backend_run(inst, L, R)
    snap = np.load(os.path.join(GOLDEN, name + ".factors.npz"))
    assert np.array_equal(L, snap["L_full"]), "L differs from the reference's factors"
    assert np.array_equal(R, snap["R_full"]), "R differs from the reference's factors"
    out = "".join("%d\n" % b for b in best if b >= 0)
    assert out == open(os.path.join(GOLDEN, name + ".out")).read()


@pytest.mark.parametrize("name", ["inst0", "inst2", "inst30-40-10-2-10"])
def test_golden_snapshots(capi, name):
    inst = capi.parse_file(golden_in(name))
    snap = np.load(os.path.join(GOLDEN, name + ".factors.npz"))
    for it in (1, 2, 10):
        L, R = capi.init_factors(inst.users, inst.items, inst.feats)
        capi.backend_factorize(inst, L, R, iters=it)
        assert np.array_equal(L, snap["L_%d" % it]) and np.array_equal(R, snap["R_%d" % it]), it


@pytest.mark.parametrize("name", ["inst0", "inst30-40-10-2-10", "instML100k"])
def test_cli_stdout_is_byte_identical_to_out(capi, name, tmp_path):
    path = golden_in(name)
    if path.endswith(".gz"):
        import gzip
        raw = gzip.open(path, "rb").read()
        path = str(tmp_path / (name + ".in"))
        open(path, "wb").write(raw)
    r = subprocess.run([capi.CLI_PATH, path], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(GOLDEN, name + ".out"), "rb").read()


# ------------------------------------------------------------------ seeded random instances vs the oracle
SHAPES = [
    # users, items, K
    (1, 1, 1), (5, 7, 2), (17, 9, 3), (40, 33, 7), (30, 40, 10), (64, 64, 20), (90, 70, 30), (33, 65, 31),
    (70, 50, 50), (50, 80, 64), (45, 45, 65), (120, 90, 100), (60, 70, 128), (40, 40, 130), (80, 60, 256),
    (24, 20, 300), (12, 10, 1000),
]


@pytest.mark.parametrize("impl", ["dma", "reg"])
@pytest.mark.parametrize("shape", SHAPES)
def test_random_instance_bit_exact(capi, orc, shape, impl, monkeypatch):
    monkeypatch.setenv("MF_SWEEP_IMPL", impl)
    u, i, k = shape
    d = random_instance(1000 + u + 7 * i + 13 * k, u, i, k, density=0.35, iters=4, alpha=0.002,
                        empty_rows=(0,) if u > 3 else (), full_rows=(2,) if u > 3 else (), float_ratings=True)
    L, R = capi.init_factors(u, i, k)
    best = capi.backend_run(_inst(capi, d), L, R)
    Lo, Ro, bo = _oracle_run(orc, d)
    assert np.array_equal(L, Lo) and np.array_equal(R, Ro)
    assert np.array_equal(best, bo)


def test_long_rows_cross_chunks(capi, orc):
    """Rows and columns with several 64-entry chunks, and a chunk size forced small."""
    d = random_instance(77, 300, 260, 100, density=0.9, iters=2, alpha=1e-4)
    L, R = capi.init_factors(300, 260, 100)
    best = capi.backend_run(_inst(capi, d), L, R)
    Lo, Ro, bo = _oracle_run(orc, d)
    assert np.array_equal(L, Lo) and np.array_equal(R, Ro) and np.array_equal(best, bo)
    for nch in ("1", "7", "33"):
        os.environ["MF_SWEEP_NCH"] = nch
        try:
            L2, R2 = capi.init_factors(300, 260, 100)
            capi.backend_factorize(_inst(capi, d), L2, R2)
        finally:
            del os.environ["MF_SWEEP_NCH"]
        assert np.array_equal(L2, Lo) and np.array_equal(R2, Ro), nch


def test_empty_and_degenerate_inputs(capi, orc):
    # no entries at all: factors unchanged, every user gets the arg-max over all items
    d = dict(iters=3, alpha=0.01, feats=4, users=6, items=5, row=np.zeros(0, np.int32), col=np.zeros(0, np.int32),
             val=np.zeros(0))
    L, R = capi.init_factors(6, 5, 4)
    L0, R0 = L.copy(), R.copy()
    best = capi.backend_run(_inst(capi, d), L, R)
    assert np.array_equal(L, L0) and np.array_equal(R, R0)
    assert np.array_equal(best, _oracle_run(orc, d)[2])
    # zero iterations: only the recommendation step
    d = random_instance(3, 20, 30, 5, iters=0)
    L, R = capi.init_factors(20, 30, 5)
    best = capi.backend_run(_inst(capi, d), L, R)
    assert np.array_equal(best, _oracle_run(orc, d)[2])
    # unsorted entries: the factorisation follows FILE order (stable bucketing), as the serial loop does
    d = random_instance(4, 25, 20, 6, iters=3, density=0.4)
    perm = np.random.default_rng(0).permutation(len(d["row"]))
    for key in ("row", "col", "val"):
        d[key] = np.ascontiguousarray(d[key][perm])
    L, R = capi.init_factors(25, 20, 6)
    capi.backend_factorize(_inst(capi, d), L, R)
    Lo, Ro = orc.init_factors(25, 20, 6)
    orc.factorize(orc.Instance(**d), Lo, Ro)
    assert np.array_equal(L, Lo) and np.array_equal(R, Ro)


def test_argument_errors(capi):
    d = random_instance(1, 5, 5, 3)
    d["col"][0] = 99
    L, R = capi.init_factors(5, 5, 3)
    with pytest.raises(capi.HipBackendError) as e:
        capi.backend_run(_inst(capi, d), L, R)
    assert e.value.status == capi.MF_ERR_ARGUMENT
    d = random_instance(1, 5, 5, 3)
    with pytest.raises(capi.HipBackendError) as e:
        capi.Plan(5, 5, 3, 0.1, d["row"], d["col"], d["val"], device=99)
    assert e.value.status == capi.MF_ERR_NO_DEVICE


def test_recommend_ties_masks_and_nan(capi, orc):
    u, i, k = 70, 150, 8
    rng = np.random.default_rng(5)
    L = rng.standard_normal((u, k))
    R = rng.standard_normal((i, k))
    R[40] = R[10]          # exact ties: the lower index must win
    R[149] = R[10]
    R[64] = R[3]
    d = random_instance(8, u, i, k, density=0.3, full_rows=(5, 66), empty_rows=(6,))
    L[9, :] = np.nan       # all scores NaN: the first unrated item is kept (strict '>' never fires)
    R[0, 0] = np.inf
    inst = _inst(capi, d)
    best = capi.backend_recommend(inst, L, R)
    assert np.array_equal(best, orc.recommend(orc.Instance(**d), L, R))
    assert best[5] == -1 and best[66] == -1


@pytest.mark.parametrize("k", [6, 30, 100])
def test_ordered_sums_every_depth_class_with_and_without_seed(capi, orc, monkeypatch, k):
    """The extreme-row path on its own terms: item rows of 20 ... 2600 entries (all three in-flight depth classes of
    ordered_sum_kernel, rows shorter than a ring, partial last blocks, a last slice that is only partly inside K) and
    user rows of a few dozen entries beside them, through the sharded level-2 API so that the item sweep runs both
    seeded from the old generation (root) and from zero (matFact-mpi.c:187) -- bit-exact against orc_shard_step."""
    U, I = 2600, 48
    rng = np.random.default_rng(1000 + k)
    lens = np.unique(np.concatenate([[U, U - 1, 2049, 1040, 1025, 777, 512, 511, 300, 129, 128, 127, 65, 33, 20],
                                     rng.integers(20, U, 12)]))[::-1]
    rows, cols = [], []
    for j, n in enumerate(lens[:I]):
        r = np.sort(rng.choice(U, int(n), replace=False))
        rows.append(r)
        cols.append(np.full(len(r), j))
    row = np.concatenate(rows).astype(np.int32)
    col = np.concatenate(cols).astype(np.int32)
    order = np.lexsort((col, row))                      # file order: by user, then item
    row, col = row[order], col[order]
    val = rng.integers(1, 6, len(row)).astype(np.float64)
    alpha = 2e-4 / k
    monkeypatch.setenv("MF_SWEEP_LONG", "24")           # (almost) every item row takes the products + ordered-sum path
    L, R = capi.init_factors(U, I, k)
    for root in (True, False):
        plan = capi.Plan(U, I, k, alpha, row, col, val)
        assert "long_rows=0/" not in plan.describe(), plan.describe()
        plan.upload(L, R)
        plan.sweep_items(seed_from_old=root)
        plan.sweep_users()
        plan.flip()
        Ln, Ra = plan.download()
        plan.close()
        Lo, Rao = orc.shard_step(0, U, I, k, row, col, val, alpha, L, R, root)
        assert np.array_equal(Ra, Rao), (k, root, np.where((Ra != Rao).any(axis=1))[0][:8])
        assert np.array_equal(Ln, Lo), (k, root)


@pytest.mark.parametrize("k", [100, 128, 64, 96])
@pytest.mark.parametrize("nch", [None, "16", "5", "64"])
def test_wave_pair_sweep_bit_exact(capi, orc, k, nch, monkeypatch):
    """sweep_pair_kernel (loader wave + compute wave per row, one barrier per chunk, two tiles): forced with MF_SWEEP_PAIR=1
    on a shape with empty rows, one-entry rows, rows of exactly one / two chunks +- 1 and a few long rows, seeded and
    unseeded item sweeps, several iterations -- bit-exact against the oracle.  K = 64 and 96 have no compile-time
    instance (the run-time-K single-wave form runs, the switch is ignored); 100 and 128 take the pair form."""
    U, I = 700, 90
    rng = np.random.default_rng(900 + k)
    lens = rng.integers(0, 12, U)
    lens[:40] = [0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 5, 4, 3, 1, 90, 89, 88, 47, 48, 49, 0, 0, 1, 1, 2, 2, 31, 33,
                 90, 90, 77, 76, 75, 20, 21, 22, 23, 24]
    row = np.repeat(np.arange(U, dtype=np.int32), lens)
    col = np.concatenate([np.sort(rng.choice(I, int(n), replace=False)) for n in lens if n] or [np.zeros(0, np.int64)]).astype(np.int32)
    val = rng.integers(1, 6, len(row)).astype(np.float64)
    monkeypatch.setenv("MF_SWEEP_PAIR", "1")
    monkeypatch.setenv("MF_ITER_MODE", "sweeps")
    monkeypatch.delenv("MF_SWEEP_DB", raising=False)    # (the fixture's double-buffered form would take precedence)
    if nch:
        monkeypatch.setenv("MF_SWEEP_NCH", nch)
    alpha = 1e-3 / k
    d = dict(iters=4, alpha=alpha, feats=k, users=U, items=I, row=row, col=col, val=val)
    plan = capi.Plan(U, I, k, alpha, row, col, val)
    desc = plan.describe()
    assert ("wave_pair=1/1" in desc) == (k in (100, 128)), desc
    L0, R0 = capi.init_factors(U, I, k)
    plan.upload(L0, R0)
    plan.iterate(4)
    L, R = plan.download()
    Lo, Ro = L0.copy(), R0.copy()
    orc.factorize(orc.Instance(**d), Lo, Ro)
    assert np.array_equal(L, Lo) and np.array_equal(R, Ro), (k, nch)
    # the sharded steps: a non-root shard's item sweep starts from zero (matFact-mpi.c:187)
    for root in (True, False):
        plan.upload(L0, R0)
        plan.sweep_items(seed_from_old=root)
        plan.sweep_users()
        plan.flip()
        Ln, Ra = plan.download()
        Lo1, Rao = orc.shard_step(0, U, I, k, row, col, val, alpha, L0, R0, root)
        assert np.array_equal(Ra, Rao) and np.array_equal(Ln, Lo1), (k, nch, root)
    plan.close()


def test_wave_pair_rule_and_extreme_rows_beside_it(capi, orc, monkeypatch):
    """The plan's own choice on a small skewed instance at K=100: the item side (one item rated by everybody) keeps the
    extreme-row path beside a wave-pair launch of the other rows, the user side (longest row walkable within the sweep's
    time) is not split at all -- and the factors are the oracle's bits."""
    U, I, K = 8000, 1500, 100   # 4.3e5 entries: a sweep of ~57 us of bytes (below 50 a tiny sweep takes one cooperative launch)
    rng = np.random.default_rng(4242)
    pop = (np.arange(I) + 1.0) ** -1.1
    pop /= pop.sum()
    lens = np.clip((rng.pareto(1.3, U) * 18 + 12).astype(np.int64), 4, 900)
    rows, cols = [], []
    for u in range(U):
        c = np.unique(np.concatenate([[0], rng.choice(I, int(lens[u]), replace=False, p=pop)]))
        rows.append(np.full(len(c), u, np.int32))
        cols.append(c.astype(np.int32))
    row, col = np.concatenate(rows), np.concatenate(cols)
    val = rng.integers(1, 6, len(row)).astype(np.float64)
    monkeypatch.setenv("MF_ITER_MODE", "sweeps")
    monkeypatch.delenv("MF_SWEEP_DB", raising=False)   # the plan's own choice (the fixture may have forced another form)
    d = dict(iters=3, alpha=2e-6, feats=K, users=U, items=I, row=row, col=col, val=val)
    plan = capi.Plan(U, I, K, d["alpha"], row, col, val)
    desc = plan.describe()
    assert "wave_pair=1/1" in desc and "long_rows=0/" not in desc and "/0 coop_nch" in desc, desc
    L0, R0 = capi.init_factors(U, I, K)
    plan.upload(L0, R0)
    plan.iterate(3)
    L, R = plan.download()
    plan.close()
    Lo, Ro = L0.copy(), R0.copy()
    orc.factorize(orc.Instance(**d), Lo, Ro)
    assert np.array_equal(L, Lo) and np.array_equal(R, Ro)


def test_switches_are_read_once_per_plan_and_experiments_are_not_in_the_shipped_library(capi, monkeypatch):
    """csrc/mf_config.hip.h: a plan's behaviour is fixed when it is created (a switch set afterwards changes nothing),
    mf_plan_describe names every documented switch that differs from its default, and the shipped library ignores the
    experiment-only ones (they exist in the -DMF_EXPERIMENTS build only)."""
    d = random_instance(3, 120, 90, 100, density=0.3)
    for k in ("MF_ITER_MODE", "MF_SWEEP_DB", "MF_SWEEP_NCH", "MF_SWEEP_SEG", "MF_SWEEP_PNCH", "MF_ES_NCH", "MF_SWEEP_FEW"):
        monkeypatch.delenv(k, raising=False)
    plain = capi.Plan(120, 90, 100, 0.01, d["row"], d["col"], d["val"])
    assert "config{" not in plain.describe(), plain.describe()
    monkeypatch.setenv("MF_SWEEP_NCH", "7")             # documented: honoured, by the NEXT plan only
    assert "config{" not in plain.describe() and " nch=16 " in plain.describe()
    later = capi.Plan(120, 90, 100, 0.01, d["row"], d["col"], d["val"])
    assert "config{MF_SWEEP_NCH=7}" in later.describe() and " nch=7 " in later.describe(), later.describe()
    monkeypatch.delenv("MF_SWEEP_NCH")
    for k, v in (("MF_SWEEP_SEG", "32"), ("MF_SWEEP_PNCH", "8"), ("MF_ES_NCH", "8"), ("MF_SWEEP_FEW", "1"), ("MF_SWEEP_PF", "0"),
                 ("MF_ES_ACTIVE", "4"), ("MF_ES_ROW_COST", "40"), ("MF_SWEEP_PAIR_I", "1"), ("MF_SWEEP_TRIO", "1"), ("MF_SWEEP_PF_ROWS", "5"), ("MF_RECOMMEND_WIDE", "1")):
        monkeypatch.setenv(k, v)                        # experiments: not compiled into the shipped library
    exp = capi.Plan(120, 90, 100, 0.01, d["row"], d["col"], d["val"])
    assert exp.describe() == plain.describe(), (exp.describe(), plain.describe())
    for p in (plain, later, exp):
        p.close()


def test_dispatch_order_of_a_large_skewed_sweep(capi, orc):
    """More than 32768 users, a few of them ten times longer than the average but below the extreme-row threshold:
    the user sweep runs from a row list with those rows first and the rest in index order; the item sweep (300 rows)
    keeps the plain order.  One iteration, bit-exact against the oracle."""
    U, I, K = 40000, 300, 100
    rng = np.random.default_rng(77)
    lens = rng.integers(20, 41, U)
    lens[rng.choice(U, 60, replace=False)] = rng.integers(250, 301, 60)
    row = np.repeat(np.arange(U, dtype=np.int32), lens)
    col = np.concatenate([np.sort(rng.choice(I, int(n), replace=False)) for n in lens]).astype(np.int32)
    val = rng.integers(1, 6, len(row)).astype(np.float64)
    d = dict(iters=1, alpha=1e-5, feats=K, users=U, items=I, row=row, col=col, val=val)
    plan = capi.Plan(U, I, K, d["alpha"], row, col, val)
    desc = plan.describe()
    L, R = capi.init_factors(U, I, K)
    plan.upload(L, R)
    plan.iterate(1)
    Lg, Rg = plan.download()
    plan.close()
    Lo, Ro = orc.init_factors(U, I, K)
    orc.factorize(orc.Instance(**d), Lo, Ro, iters=1)
    assert "long_rows=" in desc and np.array_equal(Lg, Lo) and np.array_equal(Rg, Ro), desc


def test_wave_pairs_on_a_large_side_of_long_skewed_rows(capi, orc):
    """The second case of pair_wanted (mf_build.hip.h): a side that is NOT small -- more than ~2 ms of bytes per sweep -- but
    made of long, skewed rows (600 items of ~27 000 entries, one of them rated by nearly every user) takes the wave-pair
    form, with that side's extreme-row threshold at 16e-6 nnz K; the users (133 entries on average) stay on single waves.
    One iteration of 1.6e7 entries, bit-exact against the oracle."""
    U, I, K = 120000, 600, 100
    rng = np.random.default_rng(2025)
    lens = rng.integers(20000, 33001, I)
    lens[7] = 118000                                    # the hot item: above 4x the mean and above the threshold
    lens[300] = 90000                                   # long, but walked by a pair (below 4x the mean)
    col = np.repeat(np.arange(I, dtype=np.int32), lens)
    row = np.concatenate([rng.choice(U, int(n), replace=False) for n in lens]).astype(np.int32)
    order = np.lexsort((col, row))
    row, col = np.ascontiguousarray(row[order]), np.ascontiguousarray(col[order])
    val = rng.integers(1, 6, len(row)).astype(np.float64)
    d = dict(iters=1, alpha=1e-6, feats=K, users=U, items=I, row=row, col=col, val=val)
    plan = capi.Plan(U, I, K, d["alpha"], row, col, val)
    desc = plan.describe()
    L, R = capi.init_factors(U, I, K)
    plan.upload(L, R)
    plan.iterate(1)
    Lg, Rg = plan.download()
    plan.close()
    assert "long_rows=1/0" in desc and ("wave_pair=1/0" in desc or "MF_SWEEP_DB=1" in desc), desc   # (the forced double-buffered form replaces the pairs)
    Lo, Ro = orc.init_factors(U, I, K)
    orc.factorize(orc.Instance(**d), Lo, Ro, iters=1)
    assert np.array_equal(Lg, Lo) and np.array_equal(Rg, Ro), desc


def test_plan_sharded_sweeps_match_oracle_shard_step(capi, orc):
    """matFact-mpi.c:187-208 semantics of the level-2 API: per-shard aux buffers, root seeds from old."""
    d = random_instance(31, 90, 60, 20, density=0.3, iters=1, alpha=0.003)
    L, R = capi.init_factors(90, 60, 20)
    cuts = [0, 25, 61, 90]
    R_sum = np.zeros_like(R)
    for g in range(3):
        sel = (d["row"] >= cuts[g]) & (d["row"] < cuts[g + 1])
        plan = capi.Plan(90, 60, 20, d["alpha"], d["row"][sel], d["col"][sel], d["val"][sel], user_begin=cuts[g],
                         user_count=cuts[g + 1] - cuts[g])
        Lb = np.ascontiguousarray(L[cuts[g]:cuts[g + 1]])
        plan.upload(Lb, R)
        plan.sweep_items(seed_from_old=(g == 0))
        plan.sweep_users()
        plan.flip()
        Ln, Ra = plan.download()
        Lo, Rao = orc.shard_step(cuts[g], cuts[g + 1] - cuts[g], 60, 20, d["row"][sel], d["col"][sel], d["val"][sel],
                                 d["alpha"], Lb, R, g == 0)
        assert np.array_equal(Ln, Lo) and np.array_equal(Ra, Rao)
        R_sum += Ra
        plan.close()
    Ls, Rs = L.copy(), R.copy()
    orc.factorize(orc.Instance(**d), Ls, Rs, iters=1)
    assert np.allclose(R_sum, Rs, rtol=1e-12, atol=1e-15)


# ------------------------------------------------------------------ full-size: exact spot checks + properties
def test_large_synthetic_spot_checks_and_properties(capi, orc):
    """2e5 x 2e4, K=100, ~2e7 entries (cfg4's shape at 1/5 scale; the serial oracle would need minutes per
    iteration).  Exact checks on sampled rows: a row of L_new depends only on that user's entries and R_old,
    a row of R_new only on that item's entries and L_old -- so the oracle, fed the filtered entries in file
    order, reproduces those rows bit for bit."""
    U, I, K, alpha = 200_000, 20_000, 100, 1e-4
    row, col, val = capi.synth_block(0xC0FFEE + 4, U, I, 50, 150)
    L0, R0 = capi.init_factors(U, I, K)
    plan = capi.Plan(U, I, K, alpha, row, col, val)
    plan.upload(L0, R0)
    plan.iterate(1)
    L1, R1 = plan.download()
    rng = np.random.default_rng(0)
    users = np.sort(rng.choice(U, 300, replace=False))
    sel = np.isin(row, users)
    Lo, _ = orc.shard_step(0, U, I, K, row[sel], col[sel], val[sel], alpha, L0, R0, True)
    assert np.array_equal(L1[users], Lo[users])
    items = np.sort(rng.choice(I, 40, replace=False))
    sel = np.isin(col, items)
    _, Ro = orc.shard_step(0, U, I, K, row[sel], col[sel], val[sel], alpha, L0, R0, True)
    assert np.array_equal(R1[items], Ro[items])
    # determinism: a second plan run is bit-identical
    plan2 = capi.Plan(U, I, K, alpha, row, col, val)
    plan2.upload(L0, R0)
    plan2.iterate(1)
    L1b, R1b = plan2.download()
    assert np.array_equal(L1, L1b) and np.array_equal(R1, R1b)
    # alpha = 0 is the identity (e = 0 * (a - dot) = 0 exactly)
    plan3 = capi.Plan(U, I, K, 0.0, row, col, val)
    plan3.upload(L0, R0)
    plan3.iterate(2)
    Lz, Rz = plan3.download()
    assert np.array_equal(Lz, L0) and np.array_equal(Rz, R0)
    plan3.close()
    plan2.close()
    # recommendations: exact check of sampled users against the oracle's row scorer + the mask rule
    best = plan.recommend()
    ptr = np.concatenate([[0], np.cumsum(np.bincount(row, minlength=U))])
    for uu in users[:40]:
        b = orc.predict_row(L1[uu], R1)
        rated = set(col[ptr[uu]:ptr[uu + 1]].tolist())
        exp = -1
        for j in range(I):
            if j not in rated and (exp == -1 or b[j] > b[exp]):
                exp = j
        assert best[uu] == exp
    plan.close()


def test_two_ranks_one_gpu_sharded_bench(capi):
    """The real multi-rank path (bench.py -> sharded.py -> HIP plan with caller-owned R buffers), two ranks
    sharing the one GPU of this box, gloo moving the CUDA tensors (RCCL refuses two ranks on one device).
    bench.py --check compares the sharded factors with a single-shard run: L and R within 1e-9 relative
    (only the 2-way re-association of the R sum differs; north-star tolerance 1e-5)."""
    import json
    import sys
    from conftest import ROOT
    # no launcher around it: bench.py starts its two ranks itself (child torch.distributed.run before any GPU call)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--config", "twin", "--backend", "gloo", "--check"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["rccl_ranks"] == 2 and out["collective_backend"] == "gloo"
    assert out["ms_per_step_rank_max"] >= out["ms_per_step_rank_min"] > 0
    assert out["recommend"]["recommended"] > 0 and out["check"]["recommend_differs"] <= 1
    assert out["check"]["L_max_rel"] < 1e-9 and out["check"]["R_max_rel"] < 1e-9, out["check"]


@pytest.mark.parametrize("impl", ["mfma", "mfma-one-workgroup-per-cu", "mfma-two-per-cu-any-k", "exact"])
def test_recommend_forms_agree_with_oracle(capi, orc, impl, monkeypatch):
    """MFMA filter + exact certification (64-user workgroups two per CU where K allows, and the 128-user form alone) vs
    the all-exact form vs the oracle, on shapes that cross every tile edge (64- and 128-user blocks, 64-item tile
    halves, K chunks of 8 / 16 / 20) and with planted exact ties."""
    monkeypatch.setenv("MF_RECOMMEND_IMPL", "exact" if impl == "exact" else "mfma")
    if impl == "mfma-one-workgroup-per-cu":
        monkeypatch.setenv("MF_RECOMMEND_HALF", "0")
    if impl == "mfma-two-per-cu-any-k":
        monkeypatch.setenv("MF_RECOMMEND_HALF", "all")   # the general form of recommend_mfma2_kernel (any even K <= 100)
    for seed, (u, i, k) in enumerate([(1, 1, 1), (129, 65, 17), (300, 200, 100), (257, 130, 33), (64, 1000, 4), (65, 129, 2),
                                      (200, 385, 30), (130, 700, 98), (193, 256, 6), (150, 300, 40), (70, 513, 60)]):
        rng = np.random.default_rng(40 + seed)
        L = rng.standard_normal((u, k))
        R = rng.standard_normal((i, k))
        if i > 70:
            R[i - 1] = R[0]          # exact duplicates: certified only by the exact pass, lowest index wins
            R[67] = R[3]
        d = random_instance(60 + seed, u, i, k, density=0.3, full_rows=(0,) if u > 1 else ())
        plan = capi.Plan(u, i, k, 0.0, d["row"], d["col"], d["val"])
        plan.upload(L, R)
        best = plan.recommend()
        info = plan.recommend_info()
        plan.close()
        assert np.array_equal(best, orc.recommend(orc.Instance(**d), L, R)), (impl, u, i, k)
        assert info == -1 if impl == "exact" else info >= 0


@pytest.mark.parametrize("k", [16, 20, 32, 40, 48, 60, 64, 66, 70, 72, 80, 90, 96, 98, 100, 104, 112, 128, 256])
def test_recommend_every_chunk_depth_and_staging_form(capi, orc, k, monkeypatch):
    """The MFMA pass picks its chunk depth (32 / 24 / 20), resident-L image and LDS-DMA staging from K: every
    combination, with partial last chunks (66, 70, 90, 98), against the oracle -- with and without the DMA form."""
    u, i = 300, 517
    rng = np.random.default_rng(100 + k)
    L = rng.standard_normal((u, k))
    R = rng.standard_normal((i, k))
    R[i - 1] = R[0]
    d = random_instance(200 + k, u, i, k, density=0.2, full_rows=(7,))
    want = orc.recommend(orc.Instance(**d), L, R)
    for bdma, half in (("1", "1"), ("1", "all"), ("1", "0"), ("0", "0")):   # the two-per-CU form exists with LDS-DMA staging only
        monkeypatch.setenv("MF_RECOMMEND_BDMA", bdma)
        monkeypatch.setenv("MF_RECOMMEND_HALF", half)
        plan = capi.Plan(u, i, k, 0.0, d["row"], d["col"], d["val"])
        plan.upload(L, R)
        best = plan.recommend()
        plan.close()
        assert np.array_equal(best, want), (k, bdma, half)


@pytest.mark.parametrize("half", ["1", "0"])
@pytest.mark.parametrize("split", ["0", "2", "3", "7", None])
def test_recommend_item_split_of_small_problems(capi, orc, split, half, monkeypatch):
    """A small recommendation (few 128-user blocks) splits the ITEMS over gridDim.y and merges the per-split top-2 reports
    (merge_splits_kernel): every split count -- none, the plan's own choice, counts that do not divide the tiles --
    against the oracle, with planted duplicates (certified only by the exact pass, lowest index wins), a fully rated user,
    items that are not a multiple of the tile and a user whose unrated items all lie in the last split."""
    u, i, k = 300, 1000, 100
    rng = np.random.default_rng(4100)
    L = rng.standard_normal((u, k))
    R = rng.standard_normal((i, k))
    R[i - 1] = R[0]
    R[700] = R[130]
    L[20] = 3.0 * R[130]                                 # user 20's top score is the duplicate pair 130 / 700: lowest index wins
    d = random_instance(4101, u, i, k, density=0.15, full_rows=(7,))
    mask = np.ones((u, i), bool)
    mask[11, :] = True
    mask[11, 990:] = False                               # user 11: only items 990.. are unrated
    keep = ~((d["row"] == 11))
    row = np.concatenate([d["row"][keep], np.full(990, 11, np.int32)])
    col = np.concatenate([d["col"][keep], np.arange(990, dtype=np.int32)])
    order = np.lexsort((col, row))
    row, col = np.ascontiguousarray(row[order]), np.ascontiguousarray(col[order])
    val = np.ones(len(row))
    monkeypatch.setenv("MF_RECOMMEND_HALF", half)
    if split is None:
        monkeypatch.delenv("MF_RECOMMEND_SPLIT", raising=False)
    else:
        monkeypatch.setenv("MF_RECOMMEND_SPLIT", split)
    inst = orc.Instance(1, 0.0, k, u, i, row, col, val)
    want = orc.recommend(inst, L, R)
    plan = capi.Plan(u, i, k, 0.0, row, col, val)
    plan.upload(L, R)
    best = plan.recommend()
    assert plan.recommend_info() >= 0                    # the matrix-core form ran (users whose top is a duplicate pair: exact pass)
    plan.close()
    assert np.array_equal(best, want), (split, np.where(best != want)[0][:8])
    assert want[7] == -1 and want[11] >= 990 and (want[20] == 130 or 130 in col[row == 20])


def test_mfma_certification_sends_near_ties_to_the_exact_pass(capi, orc):
    """Scores closer than the rounding bound must not be decided by the matrix cores."""
    u, i, k = 200, 300, 64
    rng = np.random.default_rng(3)
    L = rng.standard_normal((u, k))
    R = rng.standard_normal((i, k))
    R[200:] = R[:100] * (1 + 2.0 ** -52)     # 100 items one ulp-scale away from items 0..99
    d = random_instance(2, u, i, k, density=0.05)
    plan = capi.Plan(u, i, k, 0.0, d["row"], d["col"], d["val"])
    plan.upload(L, R)
    best = plan.recommend()
    sent = plan.recommend_info()
    plan.close()
    assert np.array_equal(best, orc.recommend(orc.Instance(**d), L, R))
    assert sent > 0
    # and a well-separated instance sends (almost) nobody
    R2 = rng.standard_normal((i, k))
    plan = capi.Plan(u, i, k, 0.0, d["row"], d["col"], d["val"])
    plan.upload(L, R2)
    best = plan.recommend()
    assert plan.recommend_info() <= 2
    plan.close()
    assert np.array_equal(best, orc.recommend(orc.Instance(**d), L, R2))


@pytest.mark.parametrize("name,iters", [("inst0", 5), ("inst1", 0), ("inst2", 0)])
def test_cli_mats_dump_reproduces_reference_mats_files(capi, name, iters, tmp_path):
    """samples/inst{0,1,2}.mats are the reference's own dumps of A, L, R, B at %f precision.  The CLI's
    MATFACT_MATS dump (every number from the GPU path) must reproduce them: byte for byte up to the `Final:`
    marker (initial state and the per-iteration blocks), and the Final block to 1.5e-6 -- the reference's OWN
    compiled code prints 2.019949 where inst0.mats has 2.019950 (the file predates the final revision), while
    our factors are bit-identical to that compiled code (test_golden_full_run)."""
    out = str(tmp_path / (name + ".mats"))
    env = dict(os.environ, MATFACT_MATS=out, MATFACT_MATS_ITERS=str(iters))
    r = subprocess.run([capi.CLI_PATH, golden_in(name)], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(GOLDEN, name + ".out"), "rb").read()
    got, ref = open(out).read(), open(os.path.join(GOLDEN, name + ".mats")).read()
    assert got[:got.index("Final:")] == ref[:ref.index("Final:")]
    gl, rl = got[got.index("Final:"):].split("\n"), ref[ref.index("Final:"):].split("\n")
    assert len(gl) == len(rl)
    for a, b in zip(gl, rl):
        if a[:1].isalpha() or not a:
            assert a == b
        else:
            assert np.allclose([float(x) for x in a.split()], [float(x) for x in b.split()], atol=1.5e-6, rtol=0)


def test_plan_predict_equals_oracle_rows(capi, orc):
    d = random_instance(12, 37, 29, 9, iters=2)
    L, R = capi.init_factors(37, 29, 9)
    plan = capi.Plan(37, 29, 9, d["alpha"], d["row"], d["col"], d["val"])
    plan.upload(L, R)
    B = plan.predict()
    plan.close()
    for i in (0, 17, 36):
        assert np.array_equal(B[i], orc.predict_row(L[i], R))


def test_single_process_multi_shard_run(capi, orc, tmp_path):
    """mf_backend_run_multi with several shards placed on the one GPU of this box: the same code path as one
    shard per GPU (per-shard plans, event-ordered peer reduce of the item factor, gather of the results)."""
    d = random_instance(91, 230, 140, 30, density=0.25, iters=12, alpha=0.002, empty_rows=(3, 100), full_rows=(8,))
    Lo, Ro, bo = _oracle_run(orc, d)
    for nd in (2, 3, 5):
        L, R = capi.init_factors(230, 140, 30)
        best = capi.backend_run_multi(_inst(capi, d), L, R, [0] * nd)
        # only the nd-way re-association of the sum into R differs from the serial order
        assert np.allclose(L, Lo, rtol=1e-9, atol=1e-13) and np.allclose(R, Ro, rtol=1e-9, atol=1e-13), nd
        assert np.array_equal(best, bo), nd
    # one shard == the single-GPU entry point, bit for bit
    L, R = capi.init_factors(230, 140, 30)
    best = capi.backend_run_multi(_inst(capi, d), L, R, [0])
    assert np.array_equal(L, Lo) and np.array_equal(R, Ro) and np.array_equal(best, bo)
    # reproducible: the peer reduce sums in a fixed shard order
    L1, R1 = capi.init_factors(230, 140, 30)
    L2, R2 = capi.init_factors(230, 140, 30)
    capi.backend_run_multi(_inst(capi, d), L1, R1, [0, 0, 0])
    capi.backend_run_multi(_inst(capi, d), L2, R2, [0, 0, 0])
    assert np.array_equal(L1, L2) and np.array_equal(R1, R2)
    # through the CLI
    p = tmp_path / "m.in"
    p.write_text(to_text(d))
    r = subprocess.run([capi.CLI_PATH, str(p)], capture_output=True, env=dict(os.environ, MATFACT_DEVICES="0,0,0,0"))
    assert r.returncode == 0, r.stderr
    assert r.stdout.decode() == "".join("%d\n" % b for b in bo if b >= 0)
    with pytest.raises(capi.HipBackendError) as e:
        capi.backend_run_multi(_inst(capi, d), L, R, [0, 7])
    assert e.value.status == capi.MF_ERR_NO_DEVICE


@pytest.mark.parametrize("k", [10, 20, 30, 50])
def test_row_cooperative_sweep_bit_exact(capi, orc, k, monkeypatch):
    """Few rows, some of them long (>= 128 entries): the 8-wave row-cooperative kernel (producers gather and
    scale, one accumulator wave adds in order) must reproduce the serial sums bit for bit, for every tile size."""
    rng = np.random.default_rng(500 + k)
    mask = rng.random((150, 700)) < 0.08
    mask[:10, :] = rng.random((10, 700)) < 0.9      # ten users with ~630 ratings (mean ~95)
    mask[:, :5] = True                               # five items rated by everybody (mean ~20)
    mask[5, :] = False
    row, col = np.nonzero(mask)
    d = dict(iters=3, alpha=2e-4, feats=k, users=150, items=700, row=row.astype(np.int32), col=col.astype(np.int32),
             val=(rng.random(len(row)) * 4 + 1))
    Lo, Ro, bo = _oracle_run(orc, d)
    for nch in (None, "5", "32"):
        if nch:
            monkeypatch.setenv("MF_SWEEP_NCH", nch)
        plan = capi.Plan(150, 700, k, d["alpha"], d["row"], d["col"], d["val"])
        assert "long_rows=" in plan.describe() and "long_rows=0/0" not in plan.describe()
        L, R = capi.init_factors(150, 700, k)
        plan.upload(L, R)
        plan.iterate(3)
        Lg, Rg = plan.download()
        desc = plan.describe()
        plan.close()
        bad_l, bad_r = np.where((Lg != Lo).any(axis=1))[0], np.where((Rg != Ro).any(axis=1))[0]
        assert len(bad_l) == 0 and len(bad_r) == 0, (k, nch, desc, "L rows", bad_l[:8], "R rows", bad_r[:8],
                                                     [np.where(Lg[x] != Lo[x])[0][:6] for x in bad_l[:3]],
                                                     [np.where(Rg[x] != Ro[x])[0][:6] for x in bad_r[:3]])


def test_multi_shard_item_heavy_instance_cuts_the_items(capi, orc):
    """items > users: the sharded run cuts the ITEMS (roles of L and R exchanged, mpiutil.c:54-88 /
    matFact-omp.c:44) -- results still those of the serial program (R rows exact, L summed across shards)."""
    d = random_instance(17, 24, 900, 20, density=0.3, iters=10, alpha=0.001, empty_rows=(3,))
    Lo, Ro, bo = _oracle_run(orc, d)
    for nd in (2, 4):
        L, R = capi.init_factors(24, 900, 20)
        best = capi.backend_run_multi(_inst(capi, d), L, R, [0] * nd)
        assert np.allclose(L, Lo, rtol=1e-9, atol=1e-13) and np.allclose(R, Ro, rtol=1e-9, atol=1e-13), nd
        assert np.array_equal(best, bo), nd


@pytest.mark.parametrize("k", [30, 100, 64])
def test_skewed_instance_long_rows_on_the_cooperative_kernel(capi, orc, k, monkeypatch):
    """>= 4096 rows with a few very long ones: the plan sends the long rows (and columns) to the row-cooperative
    kernel on a side stream and the rest to the single-wave kernel; the union must still be the serial result."""
    rng = np.random.default_rng(k)
    U, I = 5000, 4500
    pairs = set()
    for u in range(U):
        for j in rng.choice(I, 6, replace=False):
            pairs.add((u, int(j)))
    for u in (7, 2500, 4999):                      # three users with ~1500 ratings
        for j in rng.choice(I, 1500, replace=False):
            pairs.add((u, int(j)))
    for j in (3, 4000):                            # two items rated by ~2000 users
        for u in rng.choice(U, 2000, replace=False):
            pairs.add((int(u), j))
    pr = np.array(sorted(pairs), dtype=np.int32)
    d = dict(iters=2, alpha=1e-4, feats=k, users=U, items=I, row=np.ascontiguousarray(pr[:, 0]),
             col=np.ascontiguousarray(pr[:, 1]), val=rng.integers(1, 6, len(pr)).astype(np.float64))
    Lo, Ro, bo = _oracle_run(orc, d)
    for skew in ("1", "0"):
        monkeypatch.setenv("MF_SWEEP_SKEW", skew)
        L, R = capi.init_factors(U, I, k)
        best = capi.backend_run(_inst(capi, d), L, R)
        assert np.array_equal(L, Lo) and np.array_equal(R, Ro) and np.array_equal(best, bo), (k, skew)


def test_cli_checkpoint_and_resume_reproduce_the_uninterrupted_run(capi, tmp_path):
    """MATFACT_CHECKPOINT / MATFACT_RESUME: stop after a checkpoint, resume, get the same bytes."""
    src = golden_in("inst30-40-10-2-10")
    ck = str(tmp_path / "state.ckpt")
    gold = open(os.path.join(GOLDEN, "inst30-40-10-2-10.out"), "rb").read()
    # full run with checkpoints every 7000 iterations: last checkpoint on disk is at 14000 of 20000
    r = subprocess.run([capi.CLI_PATH, src], capture_output=True,
                       env=dict(os.environ, MATFACT_CHECKPOINT=ck, MATFACT_CHECKPOINT_EVERY="7000"))
    assert r.returncode == 0 and r.stdout == gold
    hdr = np.fromfile(ck, dtype=np.int32, count=6)
    assert bytes(np.fromfile(ck, dtype=np.uint8, count=7)) == b"MFCKPT1" and hdr[5] == 14000
    # resume from iteration 14000: the remaining 6000 iterations give the same recommendations
    r = subprocess.run([capi.CLI_PATH, src], capture_output=True, env=dict(os.environ, MATFACT_RESUME=ck))
    assert r.returncode == 0 and r.stdout == gold
    # and the resumed factors are bit-identical to the golden final factors (checked through a dump)
    out = str(tmp_path / "final.ckpt")
    r = subprocess.run([capi.CLI_PATH, src], capture_output=True,
                       env=dict(os.environ, MATFACT_RESUME=ck, MATFACT_CHECKPOINT=out, MATFACT_CHECKPOINT_EVERY="19999"))
    assert r.returncode == 0 and r.stdout == gold
    snap = np.load(os.path.join(GOLDEN, "inst30-40-10-2-10.factors.npz"))
    # checkpoint at 19999 then one more iteration: compare the 19999-state + 1 iteration through the library
    inst = capi.parse_file(src)
    body = np.fromfile(out, dtype=np.float64, offset=40)
    L = body[:30 * 10].reshape(30, 10).copy()
    R = body[30 * 10:30 * 10 + 40 * 10].reshape(40, 10).copy()
    capi.backend_factorize(inst, L, R, iters=1)
    assert np.array_equal(L, snap["L_full"]) and np.array_equal(R, snap["R_full"])
    # a checkpoint of another instance is refused
    r = subprocess.run([capi.CLI_PATH, golden_in("inst0")], capture_output=True, env=dict(os.environ, MATFACT_RESUME=ck))
    assert r.returncode == 255 and b"MATFACT_RESUME" in r.stderr


def test_randomised_shapes_all_paths_bit_exact(capi, orc, monkeypatch):
    """A seeded campaign over shapes, K (odd, even, specialised, generic), densities, skew and path overrides:
    single-wave DMA (compile-time and run-time K), register-staged (odd K), cooperative-all (tiny skewed),
    extreme-row products + ordered sum (forced by MF_SWEEP_LONG), HIP-graph replay (many iterations).
    Everything must equal the oracle bit for bit."""
    rng = np.random.default_rng(20261004)
    ks = [1, 2, 3, 6, 10, 14, 20, 30, 31, 50, 64, 66, 100, 128, 130, 200, 256]
    for case in range(48):
        k = int(ks[case % len(ks)])
        u = int(rng.integers(1, 400))
        i = int(rng.integers(1, 400))
        dens = float(rng.choice([0.02, 0.1, 0.4]))
        mask = rng.random((u, i)) < dens
        if case % 3 == 0 and u > 4 and i > 4:          # plant skew: a few dense rows and columns
            mask[rng.integers(0, u, 2), :] = rng.random((2, i)) < 0.95
            mask[:, rng.integers(0, i, 2)] = rng.random((u, 2)) < 0.95
        row, col = np.nonzero(mask)
        iters = 130 if case % 8 == 5 else int(rng.integers(1, 4))     # >= 128 iterations: the graph path
        d = dict(iters=iters, alpha=1e-3 / max(k, 1), feats=k, users=u, items=i, row=row.astype(np.int32),
                 col=col.astype(np.int32), val=rng.integers(1, 6, len(row)).astype(np.float64))
        if case % 4 == 1:
            monkeypatch.setenv("MF_SWEEP_LONG", "40")   # force the extreme-row path on small instances
        else:
            monkeypatch.delenv("MF_SWEEP_LONG", raising=False)
        L, R = capi.init_factors(u, i, k)
        best = capi.backend_run(_inst(capi, d), L, R)
        Lo, Ro, bo = _oracle_run(orc, d)
        assert np.array_equal(L, Lo) and np.array_equal(R, Ro), (case, u, i, k, dens, iters)
        assert np.array_equal(best, bo), (case, u, i, k)


@pytest.mark.parametrize("u,i,k", [(0, 5, 3), (4, 0, 3), (0, 0, 2), (3, 4, 1), (1, 1, 2)])
def test_degenerate_shapes(capi, orc, u, i, k):
    """No users, no items, nothing at all, K = 1: the C ABI must neither crash nor invent recommendations."""
    d = dict(iters=3, alpha=0.01, feats=k, users=u, items=i, row=np.zeros(0, np.int32), col=np.zeros(0, np.int32),
             val=np.zeros(0))
    if u and i:
        d = random_instance(u * 7 + i, u, i, k, density=0.5, iters=3)
    L, R = capi.init_factors(u, i, k)
    best = capi.backend_run(_inst(capi, d), L, R)
    Lo, Ro, bo = _oracle_run(orc, d)
    assert np.array_equal(L, Lo) and np.array_equal(R, Ro) and np.array_equal(best, bo)
    if u >= 2:
        L2, R2 = capi.init_factors(u, i, k)
        best2 = capi.backend_run_multi(_inst(capi, d), L2, R2, [0, 0])
        assert np.allclose(L2, Lo, rtol=1e-9, atol=1e-13) and np.array_equal(best2, bo)


def test_two_live_plans_sharing_a_run_time_k_kernel(capi, orc):
    """Two plans alive at once whose K map to the SAME run-time-K kernel instance but different LDS tile sizes
    (the dynamic-LDS limit is a per-function attribute and must only ever be raised)."""
    da = random_instance(1, 60, 50, 96, density=0.3, iters=2)
    db = random_instance(2, 40, 70, 64, density=0.3, iters=2)
    pa = capi.Plan(60, 50, 96, da["alpha"], da["row"], da["col"], da["val"])
    pb = capi.Plan(40, 70, 64, db["alpha"], db["row"], db["col"], db["val"])     # created second, smaller tiles
    out = []
    for plan, d, (u, i, k) in ((pa, da, (60, 50, 96)), (pb, db, (40, 70, 64))):
        L, R = capi.init_factors(u, i, k)
        plan.upload(L, R)
    pa.iterate(2)
    pb.iterate(2)
    for plan, d, (u, i, k) in ((pa, da, (60, 50, 96)), (pb, db, (40, 70, 64))):
        Lg, Rg = plan.download()
        Lo, Ro = orc.init_factors(u, i, k)
        orc.factorize(orc.Instance(**d), Lo, Ro, iters=2)
        assert np.array_equal(Lg, Lo) and np.array_equal(Rg, Ro)
        plan.close()


@pytest.mark.parametrize("name,iters", [("inst50000-5000-100-2-5", 300), ("inst400-50000-30-200-500", 100),
                                         ("inst600-10000-10-40-400", 300)])
def test_bundled_large_samples_factors_bit_exact(capi, orc, name, iters):
    """The reference's larger samples (users >> items, items >> users, K = 20 / 30 / 10).  Their `.in` files are too
    big to commit; samples_local/ holds copies when the tree was prepared in the build container (git-ignored) --
    skipped otherwise.  Factors after `iters` iterations must equal the oracle's bit for bit."""
    from conftest import ROOT
    path = os.path.join(ROOT, "samples_local", name + ".in")
    if not os.path.exists(path):
        pytest.skip("samples_local/ not present")
    inst = capi.parse_file(path)
    L, R = capi.init_factors(inst.users, inst.items, inst.feats)
    capi.backend_factorize(inst, L, R, iters=iters)
    oi = orc.parse_in(path)
    Lo, Ro = orc.init_factors(oi.users, oi.items, oi.feats)
    orc.factorize(oi, Lo, Ro, iters=iters)
    assert np.array_equal(L, Lo) and np.array_equal(R, Ro)


# ------------------------------------------------------------------ 2-D grid tiles
@pytest.mark.parametrize("k", [20, 7])
def test_tile_sweeps_match_oracle_tile_step(capi, orc, k):
    """matFact-mpi.c:185-205 for every tile of a 3x2 grid: item ids relative to the item block, both factors
    seeded from old only on their communicator's root; aux buffers bit-identical to the oracle's, and their
    sums over the grid rows / columns equal one serial iteration to re-association accuracy."""
    U, I = 90, 70
    d = random_instance(77, U, I, k, density=0.3, iters=1, alpha=0.003)
    L, R = capi.init_factors(U, I, k)
    ub, ib = [0, 25, 61, 90], [0, 31, 70]
    L_sum, R_sum = np.zeros_like(L), np.zeros_like(R)
    for gr in range(3):
        for gc in range(2):
            sel = ((d["row"] >= ub[gr]) & (d["row"] < ub[gr + 1]) & (d["col"] >= ib[gc]) & (d["col"] < ib[gc + 1]))
            row, col, val = d["row"][sel], d["col"][sel], d["val"][sel]
            uc, ic = ub[gr + 1] - ub[gr], ib[gc + 1] - ib[gc]
            plan = capi.Plan(U, ic, k, d["alpha"], row, col - np.int32(ib[gc]), val, user_begin=ub[gr], user_count=uc)
            Lb = np.ascontiguousarray(L[ub[gr]:ub[gr + 1]])
            Rb = np.ascontiguousarray(R[ib[gc]:ib[gc + 1]])
            plan.upload(Lb, Rb)
            plan.sweep_items(seed_from_old=(gr == 0))
            plan.sweep_users(seed_from_old=(gc == 0))
            plan.flip()
            La, Ra = plan.download()
            plan.close()
            Lo, Ro = orc.tile_step(ub[gr], uc, ib[gc], ic, k, row, col, val, d["alpha"], Lb, Rb, gc == 0, gr == 0)
            assert np.array_equal(La, Lo) and np.array_equal(Ra, Ro), (gr, gc)
            L_sum[ub[gr]:ub[gr + 1]] += La
            R_sum[ib[gc]:ib[gc + 1]] += Ra
    Ls, Rs = L.copy(), R.copy()
    orc.factorize(orc.Instance(**d), Ls, Rs, iters=1)
    assert np.allclose(L_sum, Ls, rtol=1e-12, atol=1e-15) and np.allclose(R_sum, Rs, rtol=1e-12, atol=1e-15)


def test_recommend_scored_merges_to_the_serial_answer(capi, orc):
    """mf_plan_recommend_scored per item block + merge_candidates == the serial scan over all items, with planted
    ties across blocks, NaN scores, -inf, fully rated users/blocks; block scores are the oracle's B bit for bit."""
    import importlib
    sh = importlib.import_module("recommender_system_amd.sharded")
    u, i, k = 70, 150, 8
    rng = np.random.default_rng(5)
    L = rng.standard_normal((u, k))
    R = rng.standard_normal((i, k))
    R[140] = R[10]          # tie across item blocks: the lower block must win
    R[64] = R[3]
    d = random_instance(8, u, i, k, density=0.3, full_rows=(5, 66), empty_rows=(6,))
    L[9, :] = np.nan        # every score NaN: first unrated item
    R[100, 0] = np.nan      # one NaN column in the last block
    R[0, 0] = np.inf
    want = orc.recommend(orc.Instance(**d), L, R)
    for ib in ([0, 150], [0, 50, 100, 150], [0, 1, 149, 150], [0, 64, 128, 150]):
        acc = None
        for c in range(len(ib) - 1):
            sel = (d["col"] >= ib[c]) & (d["col"] < ib[c + 1])
            plan = capi.Plan(u, ib[c + 1] - ib[c], k, 0.0, d["row"][sel], d["col"][sel] - np.int32(ib[c]), d["val"][sel])
            Rb = np.ascontiguousarray(R[ib[c]:ib[c + 1]])
            plan.upload(L, Rb)
            cand = plan.recommend_scored()
            plan.close()
            for usr in (0, 9, 33):
                if cand["best"][usr] >= 0:
                    assert cand["score"][usr] == orc.predict_row(L[usr], Rb)[cand["best"][usr]]
            for f in ("best", "first"):
                cand[f][cand[f] >= 0] += ib[c]
            acc = cand if acc is None else sh.merge_candidates(acc, cand)
        assert np.array_equal(sh.finish_candidates(acc), want), ib


def test_four_ranks_one_gpu_grid_bench(capi):
    """bench.py --grid 2x2: four ranks share this box's GPU (gloo moves the CUDA tensors), each holding one tile
    with caller-owned L and R buffers; --check compares with a single-shard run on the same GPU."""
    import json
    import sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3",
           "--warmup", "1", "--config", "twin", "--backend", "gloo", "--grid", "2x2", "--check"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 4 and "2x2 grid" in out["config"]["parallelism"] and out["value"] > 0
    assert out["check"]["L_max_rel"] < 1e-9 and out["check"]["R_max_rel"] < 1e-9, out["check"]
    # the grid's factors differ from the single-shard ones in the last bits (re-associated sums): identical
    # top-1 lists unless a user has two items within that noise, which this instance should not have
    assert out["check"]["recommend_differs"] <= 1, out["check"]


def test_power_law_extreme_path_equals_plain_path_at_scale(capi, monkeypatch):
    """A tenth of the Netflix-shaped instance of bench.py (48k x 4.4k, ~7e6 entries, top item rated by nearly every
    user): the extreme-row path (products, ordered sums under the user sweep, join before the flip) must give the
    bits of the plain one-wave-per-row path, which the smaller tests pin on the oracle."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    cfg = bench.CONFIGS["nflx"]
    U, I, K = cfg["users"] // 10, cfg["items"] // 4, 30
    row, col, val = bench.power_law_large(cfg["seed"], U, I, cfg["power_law_nnz"] // 10)
    L0, R0 = capi.init_factors(U, I, K)
    got = {}
    for mode in ("split", "plain"):
        if mode == "plain":
            monkeypatch.setenv("MF_SWEEP_SKEW", "0")
        plan = capi.Plan(U, I, K, 1e-6, row, col, val)
        desc = plan.describe()
        plan.upload(L0, R0)
        plan.iterate(3)
        got[mode] = plan.download()
        plan.close()
        assert ("long_rows=0/0" in desc) == (mode == "plain"), desc
    assert np.array_equal(got["split"][0], got["plain"][0]) and np.array_equal(got["split"][1], got["plain"][1])


def test_grid_recommend_filter_then_exact_equals_serial(capi, orc):
    """The certified form of the grid recommendation on one GPU: mf_plan_recommend_filter per item block,
    certify_filters across the blocks, mf_plan_recommend_scored_users + merge for the rest == the serial scan.
    Planted: exact ties across blocks, 1-ulp neighbours, NaN / inf scores, fully rated users and blocks."""
    import importlib
    sh = importlib.import_module("recommender_system_amd.sharded")
    u, i, k = 300, 420, 24
    rng = np.random.default_rng(17)
    L = rng.standard_normal((u, k))
    R = rng.standard_normal((i, k))
    R[400] = R[10]                 # exact tie across item blocks: the lower index must win
    R[130] = R[129]                # tie inside a block
    R[301] = np.nextafter(R[300], np.inf)
    L[9, :] = np.nan
    d = random_instance(31, u, i, k, density=0.25, full_rows=(5, 66), empty_rows=(6,))
    margin = capi.recommend_margin(k)
    for ib, poison in (([0, 420], False), ([0, 128, 256, 420], False), ([0, 11, 401, 420], False),
                       ([0, 128, 256, 420], True)):
        if poison:
            R[200, 0] = np.inf     # an infinite norm poisons the margin: nobody is certified, everything exact
        want = orc.recommend(orc.Instance(**d), L, R)
        plans, filts = [], []
        for c in range(len(ib) - 1):
            sel = (d["col"] >= ib[c]) & (d["col"] < ib[c + 1])
            plan = capi.Plan(u, ib[c + 1] - ib[c], k, 0.0, d["row"][sel], d["col"][sel] - np.int32(ib[c]), d["val"][sel])
            plan.upload(L, np.ascontiguousarray(R[ib[c]:ib[c + 1]]))
            f, norm, rmax = plan.recommend_filter()
            f["arg"][f["arg"] >= 0] += ib[c]
            plans.append(plan)
            filts.append((f, norm, rmax))
        ans, certain = sh.certify_filters([f[0] for f in filts], filts[0][1], max(f[2] for f in filts), margin)
        todo = np.flatnonzero(~certain).astype(np.int32)
        assert 9 in todo                                     # the NaN user is never certified
        assert (len(todo) == u - 2) if poison else (0 < len(todo) < u // 2), len(todo)   # 5 and 66 have nothing unrated
        acc = None
        for c, plan in enumerate(plans):
            cand = plan.recommend_scored_users(todo)
            for f in ("best", "first"):
                cand[f][cand[f] >= 0] += ib[c]
            acc = cand if acc is None else sh.merge_candidates(acc, cand)
            plan.close()
        ans[todo] = sh.finish_candidates(acc)
        assert np.array_equal(ans, want), ib


@pytest.mark.parametrize("shape", [(5, 6, 3, 0.4, 9), (6, 5, 12, 0.6, 10), (4, 4, 30, 0.6, 33), (3, 3, 40, 0.8, 8),
                                   (1, 1, 2, 1.0, 15), (7, 2, 5, 0.5, 101)])
def test_resident_toy_kernel_bit_exact(capi, orc, shape, monkeypatch):
    """sweep_resident_kernel (whole iteration loop inside one workgroup, every register-row variant and the generic
    one, odd and even iteration counts) against the oracle and against the ordinary two-launch path."""
    u, i, k, dens, iters = shape
    d = random_instance(500 + u + k, u, i, k, density=dens, iters=iters, alpha=0.01)
    assert len(d["row"]) * k <= 512
    inst = _inst(capi, d)
    L0, R0 = capi.init_factors(u, i, k)
    Lo, Ro = L0.copy(), R0.copy()
    orc.factorize(orc.Instance(**d), Lo, Ro)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MF_RESIDENT", mode)
        L, R = L0.copy(), R0.copy()
        capi.backend_factorize(inst, L, R)
        got[mode] = (L, R)
        assert np.array_equal(L, Lo) and np.array_equal(R, Ro), (shape, mode)


def test_run_top1_gives_the_list_of_the_full_run(capi, orc):
    """mf_backend_run_top1 (what the CLI calls: no copy-back of L and R) == mf_backend_run's list == the oracle's;
    the initial factors handed in stay untouched."""
    d = random_instance(77, 60, 45, 10, density=0.3, iters=25, alpha=0.002)
    inst = _inst(capi, d)
    L0, R0 = capi.init_factors(60, 45, 10)
    Lc, Rc = L0.copy(), R0.copy()
    best = capi.backend_run_top1(inst, Lc, Rc)
    assert np.array_equal(Lc, L0) and np.array_equal(Rc, R0)
    Lo, Ro, bo = orc.run(orc.Instance(**d))
    assert np.array_equal(best, bo)


# ------------------------------------------------------------------ BASELINE.json configs at their own sizes
def _spot_check_one_iteration(orc, U, I, K, alpha, row, col, val, L0, R0, L1, R1, n_users, n_items, seed=0):
    """Exact check of sampled rows after ONE iteration: a row of L_new depends only on that user's entries and the
    old factors, a row of R_new only on that item's entries -- the oracle's block update (matFact-mpi.c:185-205 with
    one rank), fed the filtered entries in file order, reproduces those rows bit for bit."""
    rng = np.random.default_rng(seed)
    users = np.sort(rng.choice(U, min(n_users, U), replace=False))
    sel = np.isin(row, users)
    Lo, _ = orc.shard_step(0, U, I, K, row[sel], col[sel], val[sel], alpha, L0, R0, True)
    assert np.array_equal(L1[users], Lo[users]), "sampled user rows differ from the oracle"
    items = np.sort(rng.choice(I, min(n_items, I), replace=False))
    sel = np.isin(col, items)
    _, Ro = orc.shard_step(0, U, I, K, row[sel], col[sel], val[sel], alpha, L0, R0, True)
    assert np.array_equal(R1[items], Ro[items]), "sampled item rows differ from the oracle"
    return users


def _check_recommend_rows(orc, best, users, row, col, L, R, U, I):
    ptr = np.concatenate([[0], np.cumsum(np.bincount(row, minlength=U))])
    for uu in users:
        b = orc.predict_row(L[uu], R)
        b[col[ptr[uu]:ptr[uu + 1]]] = -np.inf          # rated items never win; scores here are finite
        exp = int(np.argmax(b))                          # first maximum == strict '>' scanning j ascending
        assert best[uu] == exp, (uu, best[uu], exp)


@pytest.mark.parametrize("skew", [False, True])
def test_cfg3_ml1m_shaped_full_size_bit_exact(capi, orc, skew):
    """BASELINE.json configs[2] (instML1M.in is absent from the reference checkout: the ML1M-shaped synthetic
    stand-in of bench.py, 6040 x 3952, K=100, ~1e6 entries), uniform and power-law, at FULL size: two iterations
    bit-exact against the serial oracle, recommendations identical."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    cfg = bench.CONFIGS["cfg3"]
    U, I, K = cfg["users"], cfg["items"], cfg["feats"]
    row, col, val = capi.synth_block(cfg["seed"], U, I, cfg["min_row"], cfg["max_row"], **bench.synth_args(cfg, "uniform"))
    assert row.shape[0] == cfg["nnz"] == 1_000_209      # the row counts sum to nnz exactly
    if skew:
        _, raw_total = capi.synth_counts(cfg["seed"], U, I, cfg["min_row"], cfg["max_row"])   # what bench.py --skew targets
        row, col, val = bench.skewed_instance(cfg["seed"], U, I, raw_total)
    assert 8e5 < row.shape[0] < 1.2e6
    d = dict(iters=2, alpha=cfg["alpha"], feats=K, users=U, items=I, row=row, col=col, val=val)
    L, R = capi.init_factors(U, I, K)
    best = capi.backend_run(_inst(capi, d), L, R)
    Lo, Ro, bo = _oracle_run(orc, d)
    assert np.array_equal(L, Lo) and np.array_equal(R, Ro)
    assert np.array_equal(best, bo)


def test_cfg5_shape_scaled_twin_spot_checks(capi, orc):
    """BASELINE.json configs[4]'s shape (K=256, rows of 250..750 entries, seed 0xC0FFEE+5) at 2e4 x 2e4 (~1e7
    entries): sweep_dma_kernel<256,2> through long rows, sampled rows exact against the oracle, determinism,
    and the K=256 recommendation over 157 user blocks checked exactly on sampled users."""
    U, I, K, alpha = 20_000, 20_000, 256, 1e-5
    row, col, val = capi.synth_block(0xC0FFEE + 5, U, I, 250, 750)
    assert 0.9e7 < row.shape[0] < 1.1e7
    L0, R0 = capi.init_factors(U, I, K)
    plan = capi.Plan(U, I, K, alpha, row, col, val)
    assert "KT=256" in plan.describe(), plan.describe()
    plan.upload(L0, R0)
    plan.iterate(1)
    L1, R1 = plan.download()
    users = _spot_check_one_iteration(orc, U, I, K, alpha, row, col, val, L0, R0, L1, R1, 300, 40, seed=5)
    plan.upload(L0, R0)
    plan.iterate(1)
    L1b, R1b = plan.download()
    assert np.array_equal(L1, L1b) and np.array_equal(R1, R1b)
    best = plan.recommend()
    assert plan.recommend_info() >= 0                    # the matrix-core form ran
    _check_recommend_rows(orc, best, users[:25], row, col, L1, R1, U, I)
    plan.close()


def test_buffers_beyond_2_31_bytes_spot_checks(capi, orc, iter_mode):
    """A factor buffer larger than 2^31 bytes (1.2e6 users x K=256 x 8 B = 2.46e9): every row offset of the user sweep,
    of the gather of the item sweep, of the uploads / downloads and of the recommendation must be 64-bit.  Short rows keep
    it cheap (2.4e7 entries); sampled rows -- the last users among them, whose byte offsets are the largest -- exact
    against the oracle after one iteration."""
    if iter_mode == "sweeps":
        pytest.skip("covered by the other two forms")
    U, I, K, alpha = 1_200_000, 30_000, 256, 1e-5
    row, col, val = capi.synth_block(0xC0FFEE + 5, U, I, 15, 25, columns="uniform", target_nnz=24_000_000)
    assert row.shape[0] == 24_000_000 and U * K * 8 > 2**31
    L0, R0 = capi.init_factors(U, I, K)
    plan = capi.Plan(U, I, K, alpha, row, col, val)
    plan.upload(L0, R0)
    plan.iterate(1)
    L1, R1 = plan.download()
    users = _spot_check_one_iteration(orc, U, I, K, alpha, row, col, val, L0, R0, L1, R1, 200, 30, seed=31)
    tail = np.arange(U - 64, U)                          # byte offsets 2.457e9 .. 2.4576e9
    sel = row >= U - 64
    Lo, _ = orc.shard_step(0, U, I, K, row[sel], col[sel], val[sel], alpha, L0, R0, True)
    assert np.array_equal(L1[tail], Lo[tail])
    best = plan.recommend()
    _check_recommend_rows(orc, best, np.concatenate([users[:4], tail[-4:]]), row, col, L1, R1, U, I)
    plan.close()


def test_cfg5_full_size_one_iteration_spot_checks(capi, orc, iter_mode):
    """BASELINE.json configs[4] at its OWN size -- 1e6 x 1e6, K=256, 5e8 entries (distinct uniform columns, nnz exact):
    both factor buffers are 2.05e9 bytes, 5 % below 2^31, the entry arrays 2e9 and 4e9 bytes.  One iteration, 200 sampled
    users and 24 sampled items exact against the oracle.  (~25 GB of host memory, about a minute: once, not per form.)"""
    if iter_mode != "auto":
        pytest.skip("full size: run once")
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    cfg = bench.CONFIGS["cfg5"]
    U, I, K, alpha = cfg["users"], cfg["items"], cfg["feats"], cfg["alpha"]
    row, col, val = capi.synth_block(cfg["seed"], U, I, cfg["min_row"], cfg["max_row"], **bench.synth_args(cfg, "uniform"))
    assert row.shape[0] == 500_000_000
    L0, R0 = capi.init_factors(U, I, K)
    plan = capi.Plan(U, I, K, alpha, row, col, val)
    assert "KT=256" in plan.describe(), plan.describe()
    plan.upload(L0, R0)
    plan.iterate(1)
    L1, R1 = plan.download()
    plan.close()
    _spot_check_one_iteration(orc, U, I, K, alpha, row, col, val, L0, R0, L1, R1, 200, 24, seed=55)


def test_cfg4_full_size_spot_checks_and_properties(capi, orc):
    """BASELINE.json configs[3] -- the bench workload itself, 1e6 x 1e5, K=100, ~1e8 entries -- at FULL size:
    300 users / 40 items of the first iteration exact against the oracle, bit-reproducible, alpha = 0 is the
    identity, sampled recommendations exact."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    cfg = bench.CONFIGS["cfg4"]
    U, I, K, alpha = cfg["users"], cfg["items"], cfg["feats"], cfg["alpha"]
    row, col, val = capi.synth_block(cfg["seed"], U, I, cfg["min_row"], cfg["max_row"], **bench.synth_args(cfg, "uniform"))
    assert row.shape[0] == 100_000_000                  # the headline workload: distinct uniform columns, nnz exact
    L0, R0 = capi.init_factors(U, I, K)
    plan = capi.Plan(U, I, K, alpha, row, col, val)
    plan.upload(L0, R0)
    plan.iterate(1)
    L1, R1 = plan.download()
    users = _spot_check_one_iteration(orc, U, I, K, alpha, row, col, val, L0, R0, L1, R1, 300, 40, seed=4)
    best = plan.recommend()
    _check_recommend_rows(orc, best, users[:12], row, col, L1, R1, U, I)
    plan.upload(L0, R0)
    plan.iterate(1)
    L1b, R1b = plan.download()
    assert np.array_equal(L1, L1b) and np.array_equal(R1, R1b)
    del L1b, R1b
    plan.close()
    plan0 = capi.Plan(U, I, K, 0.0, row, col, val)
    plan0.upload(L0, R0)
    plan0.iterate(2)
    Lz, Rz = plan0.download()
    plan0.close()
    assert np.array_equal(Lz, L0) and np.array_equal(Rz, R0)


# ------------------------------------------------------------------ mf_backend_run_multi: reducers, bucketing, set-up cost
def test_multi_rccl_reducer_single_rank_is_the_single_gpu_run(capi, orc, monkeypatch):
    """MF_MULTI_REDUCE=rccl through the C host path: ncclCommInitAll + ncclAllReduce(ncclDouble, ncclSum) on the
    reduce stream (matFact-mpi.c:207-208).  A one-GPU box can only host ONE rank (RCCL refuses two ranks on one
    device), so MF_MULTI_FORCE=1 takes the sharded path with a single shard: the all-reduce over one rank is the
    identity and the result must equal the serial run bit for bit.  Repeated ordinals are refused for RCCL."""
    d = random_instance(92, 120, 90, 30, density=0.25, iters=8, alpha=0.002, empty_rows=(3,), full_rows=(8,))
    Lo, Ro, bo = _oracle_run(orc, d)
    monkeypatch.setenv("MF_MULTI_FORCE", "1")
    for reducer in ("rccl", "peer"):
        monkeypatch.setenv("MF_MULTI_REDUCE", reducer)
        L, R = capi.init_factors(120, 90, 30)
        best = capi.backend_run_multi(_inst(capi, d), L, R, [0])
        t = capi.multi_last_timing()
        assert t["reducer"] == reducer and t["shards"] == 1 and t["sliced"]
        assert np.array_equal(L, Lo) and np.array_equal(R, Ro) and np.array_equal(best, bo), reducer
    monkeypatch.setenv("MF_MULTI_REDUCE", "rccl")
    with pytest.raises(capi.HipBackendError) as e:
        capi.backend_run_multi(_inst(capi, d), L, R, [0, 0])
    assert e.value.status == capi.MF_ERR_UNSUPPORTED


def test_multi_unsorted_input_takes_the_bucketing_pass(capi, orc):
    """Entries in a random file order: the shards can no longer be slices of the caller's array, one stable scatter
    pass buckets them by owner; factors as for the sorted file up to re-association, in both cut directions."""
    for (u, i) in ((150, 60), (40, 300)):           # users cut / items cut
        d = random_instance(93 + u, u, i, 20, density=0.3, iters=6, alpha=0.002)
        inst = orc.Instance(**d)
        Lo, Ro = orc.init_factors(u, i, 20)
        rng = np.random.default_rng(3)
        perm = rng.permutation(len(d["row"]))
        dp = dict(d, row=np.ascontiguousarray(d["row"][perm]), col=np.ascontiguousarray(d["col"][perm]),
                  val=np.ascontiguousarray(d["val"][perm]))
        orc.factorize(orc.Instance(**dp), Lo, Ro)    # the serial program on the permuted file
        L, R = capi.init_factors(u, i, 20)
        capi.backend_run_multi(_inst(capi, dp), L, R, [0, 0, 0])
        t = capi.multi_last_timing()
        assert not t["sliced"] and t["shards"] == 3
        assert np.allclose(L, Lo, rtol=1e-9, atol=1e-13) and np.allclose(R, Ro, rtol=1e-9, atol=1e-13), (u, i)
        # the sorted file, items cut: sorted by row but not by column -> bucketing for the factorisation, slices again
        # for the user-block recommendation plans
        Ls, Rs = capi.init_factors(u, i, 20)
        best = capi.backend_run_multi(_inst(capi, d), Ls, Rs, [0, 0])
        assert capi.multi_last_timing()["sliced"] == (u >= i)
        L1, R1 = orc.init_factors(u, i, 20)
        orc.factorize(inst, L1, R1)
        assert np.array_equal(best, orc.recommend(inst, L1, R1))


def test_multi_setup_does_not_grow_with_the_shard_count(capi, monkeypatch):
    """set-up was O(shards * nnz) on the host.  The structural property, not a wall clock: the host
    makes ONE pass over the entries whatever the shard count when the input is sorted by the cut key (the shards are
    slices of the caller's array) and two when it is not (+ one stable scatter).  cfg4's shape at 1/50 scale; the 8-shard
    factors agree with the 1-shard ones to re-association accuracy and the recommendations are identical; the per-shard
    enqueueing threads and the single-thread form give the same bits."""
    import ctypes as C
    U, I, K = 20_000, 2_000, 100
    row, col, val = capi.synth_block(0xC0FFEE + 4, U, I, 50, 150)
    dev1, dev8 = np.zeros(1, np.int32), np.zeros(8, np.int32)
    monkeypatch.setenv("MF_MULTI_FORCE", "1")
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(row))
    res = {}
    for name, dev, order, threads in (("one", dev1, None, "1"), ("eight", dev8, None, "1"), ("eight_serial", dev8, None, "0"),
                                      ("eight_unsorted", dev8, perm, "1")):
        monkeypatch.setenv("MF_MULTI_THREADS", threads)
        r, c, v = (row, col, val) if order is None else (row[order], col[order], val[order])
        inst = capi.Instance(2, 1e-4, K, U, I, np.ascontiguousarray(r), np.ascontiguousarray(c), np.ascontiguousarray(v))
        p, keep = capi._problem(inst)
        L, R = capi.init_factors(U, I, K)
        best = np.empty(U, np.int32)
        capi._check(capi.hip().mf_backend_run_multi(C.byref(p), L, R, best, dev, len(dev)), "mf_backend_run_multi")
        res[name] = (L, R, best, capi.multi_last_timing())
    t1, t8, t8s, t8u = (res[k][3] for k in ("one", "eight", "eight_serial", "eight_unsorted"))
    assert t1["sliced"] and t8["sliced"] and t8["shards"] == 8 and not t8u["sliced"]
    assert t1["entry_passes"] == 1 and t8["entry_passes"] == 1 and t8u["entry_passes"] == 2, (t1, t8, t8u)
    assert t8["host_threads"] == 8 and t8s["host_threads"] == 1 and t1["host_threads"] == 1
    assert 0 < t8["enqueue_s"] <= t8["iterate_s"]
    assert np.array_equal(res["eight"][0], res["eight_serial"][0]) and np.array_equal(res["eight"][1], res["eight_serial"][1])
    assert np.allclose(res["eight"][0], res["one"][0], rtol=1e-9, atol=1e-13)
    assert np.allclose(res["eight"][1], res["one"][1], rtol=1e-9, atol=1e-13)
    assert (res["eight"][2] != res["one"][2]).sum() <= 2    # a near-tie may resolve differently after re-association
    print("multi: 1 shard set-up %.3f s; 8 shards set-up %.3f s, enqueue %.4f of %.4f s iterating (threads) / %.4f of %.4f (one thread)" % (
        t1["setup_s"], t8["setup_s"], t8["enqueue_s"], t8["iterate_s"], t8s["enqueue_s"], t8s["iterate_s"]))


# ------------------------------------------------------------------ errors + streams iteration (mf_stream.hip.h)
@pytest.mark.parametrize("k", [10, 20, 30, 50, 64, 2, 16, 6])
def test_errors_plus_streams_iteration_bit_exact_for_every_k(capi, orc, k, monkeypatch):
    """The errors launch (one wave per <= 64-entry segment of a CSR row) + the LDS-resident streams launch (a column
    slice of all of Y in LDS, a wave per run of consecutive rows) against the serial oracle, bit for bit.  Rows of every
    kind: empty (leading, trailing, in between), one entry, chunk and step boundaries, rows longer than several
    chunks; every K gives a different number and width of column slices (8, 4 or 2 columns: K <= 64 / 32 / 16 with
    the 1100-row factor)."""
    monkeypatch.setenv("MF_ITER_MODE", "es")
    rng = np.random.default_rng(900 + k)
    U, I = 70, 1100
    nch = 64
    lens = [0, 0, 1, 2, 7, 8, 9, nch - 1, nch, nch + 1, 2 * nch, 4 * nch - 1, 4 * nch, 4 * nch + 1, 5 * nch, 5 * nch + 1,
            7 * nch + 3, 0, 9 * nch, 1000, 1100, 0]
    lens = [min(x, I) for x in lens] + list(rng.integers(0, min(I, 6 * nch + 40), U - len(lens) - 2)) + [0, 0]
    rows, cols = [], []
    for u, m in enumerate(lens):
        c = np.sort(rng.choice(I, int(m), replace=False))
        rows.append(np.full(len(c), u))
        cols.append(c)
    row = np.concatenate(rows).astype(np.int32)
    col = np.concatenate(cols).astype(np.int32)
    d = dict(iters=3, alpha=3e-4, feats=k, users=U, items=I, row=row, col=col,
             val=(rng.random(len(row)) * 4 + 1))
    plan = capi.Plan(U, I, k, d["alpha"], row, col, d["val"])
    assert "iterate=errors+resident-streams" in plan.describe(), plan.describe()
    L, R = capi.init_factors(U, I, k)
    plan.upload(L, R)
    plan.iterate(3)
    Lg, Rg = plan.download()
    plan.close()
    Lo, Ro, bo = _oracle_run(orc, d)
    assert np.array_equal(Lg, Lo), "L differs"
    assert np.array_equal(Rg, Ro), "R differs"


def test_errors_plus_streams_many_short_rows_and_wide_factors(capi, orc, monkeypatch):
    """More rows than a wave may own (63): a side of thousands of short rows gets extra workgroups; and a factor too
    large for any resident slice keeps the two sweeps even when the other form is asked for."""
    monkeypatch.setenv("MF_ITER_MODE", "es")
    d = random_instance(5, 2200, 300, 10, density=0.01, iters=4, alpha=0.003, empty_rows=tuple(range(40, 120)))
    plan = capi.Plan(2200, 300, 10, d["alpha"], d["row"], d["col"], d["val"])
    assert "iterate=errors+resident-streams" in plan.describe(), plan.describe()
    L, R = capi.init_factors(2200, 300, 10)
    plan.upload(L, R)
    plan.iterate(4)
    Lg, Rg = plan.download()
    plan.close()
    Lo, Ro, _ = _oracle_run(orc, d)
    assert np.array_equal(Lg, Lo) and np.array_equal(Rg, Ro)
    d2 = random_instance(6, 30000, 40, 10, density=0.05, iters=1)
    plan = capi.Plan(30000, 40, 10, d2["alpha"], d2["row"], d2["col"], d2["val"])
    assert "iterate=sweeps" in plan.describe(), plan.describe()
    plan.close()


def test_errors_plus_streams_unsorted_input_and_graph_replay(capi, orc, monkeypatch):
    """File order that is neither row- nor column-sorted (the CSR -> CSC position map then goes through the inverse of
    the row permutation), 200 iterations so that mf_plan_iterate replays the captured pair of launches from a HIP
    graph, host- and device-built tables."""
    monkeypatch.setenv("MF_ITER_MODE", "es")
    d = random_instance(77, 90, 70, 30, density=0.3, iters=200, alpha=0.002, empty_rows=(4,), full_rows=(9,))
    perm = np.random.default_rng(5).permutation(len(d["row"]))
    dp = dict(d, row=np.ascontiguousarray(d["row"][perm]), col=np.ascontiguousarray(d["col"][perm]),
              val=np.ascontiguousarray(d["val"][perm]))
    Lo, Ro = orc.init_factors(90, 70, 30)
    orc.factorize(orc.Instance(**dp), Lo, Ro)
    for build in ("device", "host"):
        monkeypatch.setenv("MF_BUILD", build)
        L, R = capi.init_factors(90, 70, 30)
        capi.backend_factorize(_inst(capi, dp), L, R)
        assert np.array_equal(L, Lo) and np.array_equal(R, Ro), build


def test_iteration_forms_agree_on_ml100k_and_report_their_timing(capi, monkeypatch):
    """instML100k (BASELINE.json configs[1]): both iteration forms give the golden factors after 40 iterations; the
    per-launch timing interface reports one errors and one streams launch per iteration."""
    inst = capi.parse_file(golden_in("instML100k"))
    L0, R0 = capi.init_factors(inst.users, inst.items, inst.feats)
    res = {}
    for mode in ("es", "sweeps"):
        monkeypatch.setenv("MF_ITER_MODE", mode)
        plan = capi.Plan(inst.users, inst.items, inst.feats, inst.alpha, inst.row, inst.col, inst.val)
        assert ("iterate=errors+" in plan.describe()) == (mode == "es")
        plan.upload(L0, R0)
        plan.timing(True)
        plan.iterate(40)
        plan.timing(False)
        tm = plan.timing_read()
        assert tm["item_launches"] == 40 and tm["user_launches"] == 40
        res[mode] = plan.download()
        plan.close()
    assert np.array_equal(res["es"][0], res["sweeps"][0]) and np.array_equal(res["es"][1], res["sweeps"][1])


@pytest.mark.parametrize("build", ["device", "host"])
def test_recommend_masks_rated_items_of_an_unsorted_file(capi, orc, build, monkeypatch):
    """print_output's cursor (matFact.c:13-23) relies on (row, col)-sorted input; on any other order the
    reference's cursor sticks and masks nothing afterwards.  The backend masks every rated item whatever the file
    order (documented deviation): the mask walks its own copy of the item ids, ascending inside every
    row.  Checked for a fully shuffled file and for one that is row-sorted with shuffled columns; both forms of the
    recommendation; the answer is the serial program's on the SORTED file."""
    monkeypatch.setenv("MF_BUILD", build)
    u, i, k = 300, 420, 24
    rng = np.random.default_rng(23)
    L = rng.standard_normal((u, k))
    R = rng.standard_normal((i, k))
    d = random_instance(41, u, i, k, density=0.3, full_rows=(5,), empty_rows=(6,))
    want = orc.recommend(orc.Instance(**d), L, R)
    order_all = rng.permutation(len(d["row"]))
    keys = d["row"].astype(np.int64) * (i + 1) + rng.permutation(len(d["row"])) % (i + 1)
    order_rows = np.lexsort((rng.random(len(d["row"])), d["row"]))     # rows ascending, columns shuffled inside a row
    for order in (order_all, order_rows):
        dp = dict(d, row=np.ascontiguousarray(d["row"][order]), col=np.ascontiguousarray(d["col"][order]),
                  val=np.ascontiguousarray(d["val"][order]))
        for impl in ("mfma", "exact"):
            monkeypatch.setenv("MF_RECOMMEND_IMPL", impl)
            got = capi.backend_recommend(_inst(capi, dp), L, R)
            assert np.array_equal(got, want), (build, impl)


def test_cli_cached_input_prints_the_same_bytes(capi, tmp_path):
    """MATFACT_CACHE: the first run parses and writes the binary cache, the second maps it; stdout is the
    reference's .out both times."""
    cache = tmp_path / "cache"
    cache.mkdir()
    name = "inst30-40-10-2-10"
    want = open(os.path.join(GOLDEN, name + ".out"), "rb").read()
    for run in range(2):
        r = subprocess.run([capi.CLI_PATH, golden_in(name)], capture_output=True,
                           env=dict(os.environ, MATFACT_CACHE=str(cache), MATFACT_TIMING="1"))
        assert r.returncode == 0, r.stderr
        assert r.stdout == want
        assert (b"parse(cache)" in r.stderr) == (run == 1), r.stderr
    assert len(os.listdir(cache)) == 1


def test_row_pitch_of_own_and_caller_buffers(capi, orc):
    """128-byte row pitch: the plan pads rows of its own factor buffers to whole 128-byte lines
    where that saves gathered lines (K=10: 80-byte rows -> pitch 16 doubles, K=30 -> 32; K=20 and K=100 stay packed);
    caller-owned buffers declare their pitch (mf_shard.items_pitch / users_pitch).  Factors bit-exact for every
    combination of padded / packed, own / caller-owned, and with the padding switched off."""
    import torch
    assert [capi.row_pitch(k) for k in (10, 20, 30, 50, 100, 14, 7)] == [16, 20, 32, 50, 100, 16, 7]
    for k in (10, 30):
        d = random_instance(60 + k, 130, 90, k, density=0.3, iters=4, alpha=0.003, empty_rows=(2,), full_rows=(7,))
        Lo, Ro, bo = _oracle_run(orc, d)
        ld = capi.row_pitch(k)
        dev = torch.device("cuda", 0)
        for own in (True, False):
            kw = {}
            if not own:
                rb = [torch.full((90, ld), float("nan"), dtype=torch.float64, device=dev) for _ in range(2)]
                lb = [torch.full((130, ld), float("nan"), dtype=torch.float64, device=dev) for _ in range(2)]
                kw = dict(items_ext=[t.data_ptr() for t in rb], items_pitch=ld,
                          users_ext=[t.data_ptr() for t in lb], users_pitch=ld)
            plan = capi.Plan(130, 90, k, d["alpha"], d["row"], d["col"], d["val"], **kw)
            assert plan.pitches() == (ld, ld)
            L, R = capi.init_factors(130, 90, k)
            plan.upload(L, R)
            plan.iterate(4)
            Lg, Rg = plan.download()
            best = plan.recommend()
            plan.close()
            assert np.array_equal(Lg, Lo) and np.array_equal(Rg, Ro) and np.array_equal(best, bo), (k, own)
    # a pitch that is odd or smaller than K is refused
    with pytest.raises(capi.HipBackendError) as e:
        capi.Plan(10, 10, 10, 0.1, d["row"][:0], d["col"][:0], d["val"][:0], items_ext=[256, 512], items_pitch=8)
    assert e.value.status == capi.MF_ERR_ARGUMENT
