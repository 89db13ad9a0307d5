"""Debug aid: per-step clock stamps of the streams launch on instML100k (MF_ES_DBG=64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MF_ITER_MODE"] = "es"
os.environ["MF_ES_DBG"] = os.environ.get("MF_ES_DBG", "64")
os.environ["MF_ES_STAMPS"] = "gpurun_out/stamps.txt"
import numpy as np
import recommender_system_amd as rs
c = rs.capi
inst = c.parse_file("tests/golden/instML100k.in.gz")
L, R = c.init_factors(inst.users, inst.items, inst.feats)
plan = c.Plan(inst.users, inst.items, inst.feats, inst.alpha, inst.row, inst.col, inst.val)
plan.upload(L, R)
plan.timing(True)   # no graph
plan.iterate(20)
plan.synchronize()
rows = [list(map(int, ln.split(":")[1].split())) for ln in open("gpurun_out/stamps.txt")]
a = np.array(rows, dtype=np.int64)
t0 = a[:, 0].min()
print("waves", a.shape[0], "start spread (cycles)", int(a[:, 0].max() - t0), "end max", int(a[:, 63].max() - t0))
for w in (0, 1, 100, 400, 700, 766):
    st = a[w]
    pts = [(int(st[4 * s + p]) - int(st[0])) if st[4 * s + p] else None for s in range(16) for p in range(4)]
    print("wave", w, "begin@", int(st[0] - t0), "end@", int(st[63] - t0))
    for s in range(16):
        print("   step", s, pts[4 * s:4 * s + 4])
