#!/usr/bin/env python3
"""Wall time of the level-1 drop-in call mf_backend_run (host buffers in, host buffers out: CSR/CSC build,
H2D, iterations, recommend, D2H) -- the PCIe-inclusive figure DESIGN.md quotes beside bench.py's value."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommender_system_amd as rs
ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=100_000)
ap.add_argument("--feats", type=int, default=100)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
c = rs.capi
row, col, val = c.synth_block(0xC0FFEE + 4, a.users, a.items, 50, 150)
inst = c.Instance(a.iters, 1e-4, a.feats, a.users, a.items, row, col, val)
L, R = c.init_factors(a.users, a.items, a.feats)
c.device_count()
p, keep = c._problem(inst)
best = np.empty(a.users, np.int32)
for rep in range(2):
    t = time.perf_counter()
    rc = c.hip().mf_backend_run(p, L, R, best, 0)
    dt = time.perf_counter() - t
    assert rc == 0
    print("mf_backend_run: %.3f s wall for %d iterations of %d entries + recommend -> %.3e nnz-updates/s "
          "(host buffers, PCIe and CSR/CSC build included)" % (dt, a.iters, inst.nnz, a.iters * inst.nnz / dt), flush=True)
