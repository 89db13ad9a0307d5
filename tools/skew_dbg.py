"""Which rows differ between a variant of the extreme-row path and the plain sweeps (debugging aid)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import recommender_system_amd as rs
capi = rs.capi
if os.environ.get("DBG_CFG") == "nflx10":   # a tenth of the Netflix shape, K=30 (partly filled last slice)
    cfg = dict(bench.CONFIGS["nflx"], alpha=1e-6)
    U, I, K = cfg["users"] // 10, cfg["items"] // 4, int(os.environ.get("DBG_K", "30"))
    row, col, val = bench.power_law_large(cfg["seed"], U, I, cfg["power_law_nnz"] // 10)
elif os.environ.get("DBG_CFG") == "ml100k":   # the errors + resident-streams iteration against the plain sweeps
    cfg = bench.CONFIGS["ml100k"]
    inst = capi.parse_file(cfg["file"])
    U, I, K = inst.users, inst.items, inst.feats
    row, col, val = inst.row, inst.col, inst.val
    cfg = dict(cfg, alpha=float(os.environ.get("DBG_ALPHA", "1e-4")))
else:
    cfg = bench.CONFIGS["cfg3"]
    U, I, K = cfg["users"], cfg["items"], cfg["feats"]
    counts, total = capi.synth_counts(cfg["seed"], U, I, cfg["min_row"], cfg["max_row"])
    row, col, val = bench.skewed_instance(cfg["seed"], U, I, total)
L0, R0 = capi.init_factors(U, I, K)
ulen = np.bincount(row, minlength=U); ilen = np.bincount(col, minlength=I)
def run(env, iters=1):
    saved = dict(os.environ)
    os.environ.update(env)
    try:
        plan = capi.Plan(U, I, K, cfg["alpha"], row, col, val)
        d = plan.describe()
        plan.upload(L0, R0); plan.iterate(iters); out = plan.download(); plan.close()
    finally:
        os.environ.clear(); os.environ.update(saved)
    return out, d
noise = noise_stream = None
if os.environ.get("DBG_NOISE"):
    import torch
    noise_stream = torch.cuda.Stream()
    noise = torch.ones(64 << 20, dtype=torch.float64, device="cuda")   # 512 MB: ~0.3 ms per pass
variants = [dict(kv.split("=") for kv in a.split(",")) if a != "-" else {} for a in sys.argv[1:]] or [{}]
def make(env):
    saved = dict(os.environ)
    os.environ.update(env)
    try:
        plan = capi.Plan(U, I, K, cfg["alpha"], row, col, val)
    finally:
        os.environ.clear(); os.environ.update(saved)
    plan.upload(L0, R0)
    return plan
ref = make({"MF_SWEEP_SKEW": "0", "MF_ITER_MODE": "sweeps"})
for env in variants:
    env = dict(env, MF_ITER_MODE=os.environ.get("DBG_MODE", "sweeps"))
    var = make(env)
    print(env, var.describe().split("row_bytes")[1][:120], flush=True)
    ref.upload(L0, R0)
    bad = 0
    for it in range(int(os.environ.get("DBG_ITERS", "400"))):
        if noise is not None:   # a memory-bound neighbour on another stream while the variant iterates
            with torch.cuda.stream(noise_stream):
                for _ in range(4): noise.mul_(1.0000001)
        ref.iterate(1); var.iterate(1)
        r = ref.download(); g = var.download()
        bl = np.where((g[0] != r[0]).any(axis=1))[0]; br = np.where((g[1] != r[1]).any(axis=1))[0]
        if len(bl) or len(br):
            bad += 1
            print(env, "iteration", it, "| L rows differ", len(bl), "lens", sorted(ulen[bl])[:6], "| R rows differ", len(br), "lens", sorted(ilen[br])[:6], flush=True)
            for nm, b, gg, rr in (("L", bl, g[0], r[0]), ("R", br, g[1], r[1])):
                for x in b[:3]:
                    c = np.where(gg[x] != rr[x])[0]
                    print("    ", nm, "row", x, "cols", c[:12], "n", len(c), "rel", float(np.max(np.abs(gg[x][c] - rr[x][c]) / np.abs(rr[x][c]))))
            var.upload(r[0], r[1])
            if bad >= 4: break
    print(env, "done:", bad, "bad iterations", flush=True)
    var.close()
