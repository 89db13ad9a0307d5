#!/usr/bin/env python3
"""Where waves 0 and 7 of workgroup 0 of recommend_mfma_kernel spend their cycles: shader-clock totals from the
diagnostic build (make -C recommender-system_amd csrc/libmatfact_hip_stamps.so; MF_HIP_LIB pointing at it)."""
import argparse, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommender_system_amd as rs

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=131072)
ap.add_argument("--items", type=int, default=100000)
ap.add_argument("--feats", type=int, default=100)
a = ap.parse_args()
c = rs.capi
row, col, val = c.synth_block(0xC0FFEE + 4, a.users, a.items, 50, 150)
rng = np.random.default_rng(0)
L = rng.random((a.users, a.feats)) / a.feats
R = rng.random((a.items, a.feats)) / a.feats
plan = c.Plan(a.users, a.items, a.feats, 1e-4, row, col, val)
plan.upload(L, R)
plan.iterate(2)
lib = c.hip()
buf = (C.c_ulonglong * 32)()
plan.recommend(); lib.mf_debug_read_rec_stamps(buf)
t = time.perf_counter(); plan.recommend(); dt = time.perf_counter() - t
lib.mf_debug_read_rec_stamps(buf)
print("recommend %.4f s  %.2f TFLOP/s" % (dt, 2.0 * a.users * a.items * a.feats / dt / 1e12))
for w, o in ((0, 0), (7, 16)):
    tiles, chunks, mask, issue, k, land, bar, arg, whole = [int(buf[o + i]) for i in range(9)]
    ksteps = (a.feats + 3) // 4
    print("wave %d: tiles %d chunks %d | cycles per tile: mask %.0f  prefetch issue %.0f  k-steps %.0f (%.0f per k-step; 8 matrix "
          "instructions of 64 cycles = 512, x2 waves per SIMD = 1024)  landing wait %.0f  barrier %.0f  arg-max %.0f | whole %.0f "
          "(ideal %d)" % (w, tiles, chunks, mask / tiles, issue / tiles, k / tiles, k / tiles / ksteps, land / tiles, bar / tiles,
                          arg / tiles, whole / tiles, ksteps * 1024))
