#!/usr/bin/env python3
"""Netflix-shaped power-law instance: the extreme-row path (products + ordered sums, deferred join) against the
plain one-wave-per-row path (MF_SWEEP_SKEW=0, itself pinned bit-exact on the oracle by the tests): factors after a
few iterations must be bit-identical."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import recommender_system_amd as rs

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cfg = bench.CONFIGS["nflx"]
U, I, K = int(cfg["users"] * scale), int(cfg["items"] * max(scale, 0.25)), cfg["feats"]
row, col, val = bench.power_law_large(cfg["seed"], U, I, int(cfg["power_law_nnz"] * scale))
L0, R0 = rs.capi.init_factors(U, I, K)
out = {}
for mode in ("split", "plain"):
    if mode == "plain":
        os.environ["MF_SWEEP_SKEW"] = "0"
    else:
        os.environ.pop("MF_SWEEP_SKEW", None)
    plan = rs.capi.Plan(U, I, K, 1e-6, row, col, val)
    print(mode, plan.describe(), flush=True)
    plan.upload(L0, R0)
    t = time.perf_counter()
    plan.iterate(3)
    plan.synchronize()
    print("  3 iterations %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
    out[mode] = plan.download()
    plan.close()
same = np.array_equal(out["split"][0], out["plain"][0]) and np.array_equal(out["split"][1], out["plain"][1])
print("nnz", row.shape[0], "bit-identical:", same)
sys.exit(0 if same else 1)
