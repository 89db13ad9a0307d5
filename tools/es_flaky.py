"""Hunt for the rare mismatch seen in the GPU suite (round 3): the shapes of the three flaky failures, each run many times
through backend_run (plan creation + iterations + download) against the oracle.  Prints which runs differed and where."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import recommender_system_amd as rs
from oracle import oracle as O
capi = rs.capi
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(20261004)
ks = [1, 2, 3, 6, 10, 14, 20, 30, 31, 50, 64, 66, 100, 128, 130, 200, 256]
cases = []
for case in range(48):
    k = int(ks[case % len(ks)]); u = int(rng.integers(1, 400)); i = int(rng.integers(1, 400))
    dens = float(rng.choice([0.02, 0.1, 0.4])); mask = rng.random((u, i)) < dens
    if case % 3 == 0 and u > 4 and i > 4:
        mask[rng.integers(0, u, 2), :] = rng.random((2, i)) < 0.95
        mask[:, rng.integers(0, i, 2)] = rng.random((u, 2)) < 0.95
    row, col = np.nonzero(mask)
    iters = 130 if case % 8 == 5 else int(rng.integers(1, 4))
    d = dict(iters=iters, alpha=1e-3 / max(k, 1), feats=k, users=u, items=i, row=row.astype(np.int32), col=col.astype(np.int32),
             val=rng.integers(1, 6, len(row)).astype(np.float64))
    cases.append(d)
d = cases[7]
print("case 7:", d["users"], d["items"], d["feats"], len(d["row"]), "iters", d["iters"], "mode", os.environ.get("MF_ITER_MODE", "auto"), flush=True)
inst = capi.Instance(d["iters"], d["alpha"], d["feats"], d["users"], d["items"], d["row"], d["col"], d["val"])
oi = O.Instance(**d)
Lo, Ro = O.init_factors(oi.users, oi.items, oi.feats); O.factorize(oi, Lo, Ro)
bad = 0
for r in range(reps):
    L, R = capi.init_factors(d["users"], d["items"], d["feats"])
    capi.backend_run(inst, L, R)
    bl, br = np.where((L != Lo).any(axis=1))[0], np.where((R != Ro).any(axis=1))[0]
    if len(bl) or len(br):
        bad += 1
        print("run", r, "L rows", bl[:6], "R rows", br[:6], [np.where(L[x] != Lo[x])[0][:8] for x in bl[:2]], [np.where(R[x] != Ro[x])[0][:8] for x in br[:2]],
              "rowlens", np.bincount(d["row"], minlength=d["users"])[bl[:6]], np.bincount(d["col"], minlength=d["items"])[br[:6]], flush=True)
print("done:", bad, "of", reps, "runs differ", flush=True)
