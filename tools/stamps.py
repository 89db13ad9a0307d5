"""Where a wave walking a long row alone spends its cycles: phase clocks of sweep_dma_kernel from the diagnostic build
(make -C recommender-system_amd csrc/libmatfact_hip_stamps.so; run with MF_HIP_LIB pointing at it).
An instance of ONE item rated by N users (and N users with one rating each): the item sweep is a single wave."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommender_system_amd as rs
capi = rs.capi
N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 100
busy = len(sys.argv) > 3   # a second dimension of items so the chip is full beside the long row
items = 1 + (3000 if busy else 0)
row = np.arange(N, dtype=np.int32); col = np.zeros(N, np.int32)
if busy:
    rng = np.random.default_rng(1)
    r2 = np.repeat(np.arange(N, dtype=np.int32), 150); c2 = rng.integers(1, items, r2.shape[0]).astype(np.int32)
    key = np.unique(r2.astype(np.int64) * items + c2)
    key = np.concatenate([row.astype(np.int64) * items, key]); key.sort()
    row, col = (key // items).astype(np.int32), (key % items).astype(np.int32)
val = np.ones(row.shape[0])
os.environ["MF_SWEEP_SKEW"] = "0"; os.environ["MF_ITER_MODE"] = "sweeps"
L0, R0 = capi.init_factors(N, items, K)
plan = capi.Plan(N, items, K, 1e-4, row, col, val)
print(plan.describe())
plan.upload(L0, R0)
buf = (C.c_ulonglong * 8)()
lib = capi.hip()
plan.iterate(3); lib.mf_debug_read_stamps(buf)
plan.iterate(10); lib.mf_debug_read_stamps(buf)
rows, chunks, issue, wait, a, b, whole = [int(buf[i]) for i in range(7)]
print("rows %d chunks %d | cycles per chunk: issue %.0f  landing wait %.0f  phase A %.0f  phase B %.0f  | whole row %.0f per chunk (%.0f cycles per entry)" % (
    rows, chunks, issue / chunks, wait / chunks, a / chunks, b / chunks, whole / chunks, whole / chunks / 16))
plan.close()
