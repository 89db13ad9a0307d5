#!/usr/bin/env python3
"""Times mf_plan_recommend alone on a synthetic shard (for rocprofv3 runs of the recommend kernels)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommender_system_amd as rs

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=131072)
ap.add_argument("--items", type=int, default=100000)
ap.add_argument("--feats", type=int, default=100)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
c = rs.capi
row, col, val = c.synth_block(0xC0FFEE + 4, a.users, a.items, 50, 150)
rng = np.random.default_rng(0)
L = rng.random((a.users, a.feats)) / a.feats
R = rng.random((a.items, a.feats)) / a.feats
plan = c.Plan(a.users, a.items, a.feats, 1e-4, row, col, val)
plan.upload(L, R)
plan.iterate(2)
for r in range(a.reps):
    t = time.perf_counter()
    best = plan.recommend()
    dt = time.perf_counter() - t
    print("recommend %.4f s  %.2f TFLOP/s  exact-pass users %d" % (dt, 2.0 * a.users * a.items * a.feats / dt / 1e12,
                                                                   plan.recommend_info()), flush=True)
