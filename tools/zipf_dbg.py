"""cfg4 with Zipf(1.0) item popularity: which rows differ between the shipped schedule and the plain single-wave sweeps after
ONE iteration (bench.py --check reported a mismatch after 23)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import recommender_system_amd as rs
capi = rs.capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = dict(bench.CONFIGS["cfg4"])
U, I, K = cfg["users"] // scale, cfg["items"] // scale, cfg["feats"]
cfg["nnz"] //= scale
row, col, val = capi.synth_block(cfg["seed"], U, I, cfg["min_row"], cfg["max_row"], columns="zipf", target_nnz=cfg["nnz"])
ilen = np.bincount(col, minlength=I)
print("instance", U, I, K, len(row), "longest items", np.sort(ilen)[::-1][:6], flush=True)
L0, R0 = capi.init_factors(U, I, K)
def run(env):
    saved = dict(os.environ); os.environ.update(env)
    try:
        plan = capi.Plan(U, I, K, cfg["alpha"], row, col, val)
    finally:
        os.environ.clear(); os.environ.update(saved)
    print(env, plan.describe(), flush=True)
    plan.upload(L0, R0); plan.iterate(1); out = plan.download(); plan.close()
    return out
iters = int(os.environ.get("DBG_ITERS", "0"))
if iters:   # lockstep: the first iteration at which the two schedules part, and where
    def make(env):
        saved = dict(os.environ); os.environ.update(env)
        try:
            plan = capi.Plan(U, I, K, cfg["alpha"], row, col, val)
        finally:
            os.environ.clear(); os.environ.update(saved)
        plan.upload(L0, R0)
        return plan
    ref = make({"MF_SWEEP_SKEW": "0", "MF_ITER_MODE": "sweeps", "MF_SWEEP_DB": "0", "MF_SWEEP_PAIR": "0"})
    var = make(dict(kv.split("=") for kv in os.environ.get("DBG_ENV", "").split(",") if kv))
    print(var.describe(), flush=True)
    ulen = np.bincount(row, minlength=U)
    bad = 0
    for it in range(iters):
        ref.iterate(1); var.iterate(1)
        r = ref.download(); g = var.download()
        bl = np.where((g[0] != r[0]).any(axis=1))[0]; br = np.where((g[1] != r[1]).any(axis=1))[0]
        if len(bl) or len(br):
            bad += 1
            print("iteration", it, "L rows", len(bl), bl[:6], "lens", ulen[bl[:6]], "| R rows", len(br), br[:6], "lens", ilen[br[:6]], flush=True)
            for nm, b, gg, rr in (("L", bl, g[0], r[0]), ("R", br, g[1], r[1])):
                for x in b[:3]:
                    c = np.where(gg[x] != rr[x])[0]
                    print("    ", nm, "row", x, "cols", c[:16], "n", len(c), "rel", float(np.max(np.abs(gg[x][c] - rr[x][c]) / np.abs(rr[x][c]))), flush=True)
            var.upload(r[0], r[1])
            if bad >= 4: break
    print("done:", bad, "bad iterations of", iters, flush=True)
    sys.exit(0)
ref = run({"MF_SWEEP_SKEW": "0", "MF_ITER_MODE": "sweeps", "MF_SWEEP_DB": "0", "MF_SWEEP_PAIR": "0"})
for env in ({}, {"MF_OS_DPP": "0"}):
    got = run(env)
    bl = np.where((got[0] != ref[0]).any(axis=1))[0]; br = np.where((got[1] != ref[1]).any(axis=1))[0]
    print(env, "L rows differ", len(bl), "R rows differ", len(br), "lens", ilen[br][:10], flush=True)
    for x in br[:4]:
        c = np.where(got[1][x] != ref[1][x])[0]
        print("   R row", x, "len", ilen[x], "cols", c[:16], "n", len(c), "rel", float(np.max(np.abs(got[1][x][c] - ref[1][x][c]) / np.abs(ref[1][x][c]))), flush=True)
