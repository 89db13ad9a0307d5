import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the in-tree libraries exist (hipcc cross-compiles without a GPU)."""
    pkg = os.path.join(ROOT, "recommender-system_amd")
    need = [os.path.join(pkg, "csrc", "libmatfact_hip.so"), os.path.join(pkg, "host", "libmatfact_host.so"),
            os.path.join(pkg, "host", "matFact"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-s", "-C", pkg, "all"])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


@pytest.fixture(scope="session")
def capi():
    import recommender_system_amd as rs
    return rs.capi


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as O
    return O


def golden_in(name):
    for ext in (".in", ".in.gz"):
        p = os.path.join(GOLDEN, name + ext)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(name)


def random_instance(seed, users, items, feats, density=0.2, iters=3, alpha=0.01, empty_rows=(), full_rows=(),
                    float_ratings=False):
    """Seeded (row, col)-sorted instance with optional empty / completely rated users."""
    rng = np.random.default_rng(seed)
    mask = rng.random((users, items)) < density
    for r in empty_rows:
        mask[r, :] = False
    for r in full_rows:
        mask[r, :] = True
    row, col = np.nonzero(mask)
    val = (rng.random(row.shape[0]) * 4 + 1) if float_ratings else rng.integers(1, 6, row.shape[0]).astype(np.float64)
    return dict(iters=iters, alpha=alpha, feats=feats, users=users, items=items, row=row.astype(np.int32),
                col=col.astype(np.int32), val=val.astype(np.float64))


def to_text(d):
    lines = ["%d" % d["iters"], repr(float(d["alpha"])), "%d" % d["feats"],
             "%d %d %d" % (d["users"], d["items"], len(d["row"]))]
    lines += ["%d %d %r" % (r, c, float(v)) for r, c, v in zip(d["row"], d["col"], d["val"])]
    return "\n".join(lines) + "\n"
