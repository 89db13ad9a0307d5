"""Which of the candidate causes produces the rare wrong ordered sum of round 2's loop order?  Runs the diagnostic build
(make -C recommender-system_amd csrc/libmatfact_hip_osdiag.so; MF_HIP_LIB pointing at it): the tenth-scale Netflix shape
with MF_SWEEP_LONG=3000 in lockstep with the plain sweeps (as tools/skew_dbg.py), reading the kernel's own cross-check
records after every iteration.  See mf_sweep.hip.h, ordered_sum_task_diag, for what the two flag bits mean."""
import ctypes as C, os, struct, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import recommender_system_amd as rs
capi = rs.capi
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cfg = dict(bench.CONFIGS["nflx"], alpha=1e-6)
U, I, K = cfg["users"] // 10, cfg["items"] // 4, int(os.environ.get("DBG_K", "30"))
row, col, val = bench.power_law_large(cfg["seed"], U, I, cfg["power_law_nnz"] // 10)
L0, R0 = capi.init_factors(U, I, K)
def make(env):
    saved = dict(os.environ); os.environ.update(env)
    try:
        plan = capi.Plan(U, I, K, cfg["alpha"], row, col, val)
    finally:
        os.environ.clear(); os.environ.update(saved)
    plan.upload(L0, R0)
    return plan
ref = make({"MF_SWEEP_SKEW": "0", "MF_ITER_MODE": "sweeps", "MF_SWEEP_PAIR": "0"})
var = make({"MF_SWEEP_LONG": os.environ.get("DBG_LONG", "3000"), "MF_ITER_MODE": "sweeps", "MF_SWEEP_PAIR": "0", "MF_SIDE_PRIO": "1"})
print(var.describe(), flush=True)
lib = capi.hip()
buf = (C.c_ulonglong * (2 + 8 * 32))()
lib.mf_debug_read_os_diag(buf, len(buf))
bad = flagged = 0; blocks = 0
f64 = lambda u: struct.unpack("<d", struct.pack("<Q", u))[0]
for it in range(iters):
    ref.iterate(1); var.iterate(1)
    r = ref.download(); g = var.download()
    lib.mf_debug_read_os_diag(buf, len(buf))
    blocks += int(buf[0]); nrec = int(buf[1])
    wrong = bool((g[0] != r[0]).any() or (g[1] != r[1]).any())
    if wrong or nrec:
        bad += wrong; flagged += nrec > 0
        bl = np.where((g[0] != r[0]).any(axis=1))[0]; br = np.where((g[1] != r[1]).any(axis=1))[0]
        print("iteration", it, "result wrong" if wrong else "result right", "| L rows", bl[:4], "R rows", br[:4], "| records", nrec, flush=True)
        for k in range(min(nrec, 32)):
            w = [int(buf[2 + 8 * k + j]) for j in range(8)]
            print("   task %d block %d of %d lane %d flags %d depth %d | reg %.17g slot-now %.17g | dpp %.17g plain %.17g | hw_id %x" % (
                w[0], w[1] & 0xffffffff, w[1] >> 32, w[2] & 0xffffffff, (w[2] >> 32) & 0xff, w[2] >> 40, f64(w[3]), f64(w[4]), f64(w[5]), f64(w[6]), w[7]), flush=True)
        var.upload(r[0], r[1])
print("done: %d iterations, %d blocks checked, %d wrong results, %d iterations with records" % (iters, blocks, bad, flagged), flush=True)
