"""instML100k: wall-clock of mf_plan_iterate(3000) in ONE call (HIP-graph replay of 32-iteration blocks, what the CLI
runs) for both iteration forms -- the per-step figure of bench.py includes one Python -> C call and six event records
per iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import recommender_system_amd as rs
c = rs.capi
inst = c.parse_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "instML100k.in.gz"))
L, R = c.init_factors(inst.users, inst.items, inst.feats)
for mode in ("auto", "sweeps"):
    os.environ["MF_ITER_MODE"] = mode
    plan = c.Plan(inst.users, inst.items, inst.feats, inst.alpha, inst.row, inst.col, inst.val)
    plan.upload(L, R)
    plan.iterate(256)
    plan.synchronize()
    for graph in ("1", "0"):
        os.environ["MF_GRAPH"] = graph
        t = time.perf_counter()
        plan.iterate(3000)
        plan.synchronize()
        dt = time.perf_counter() - t
        print("%-7s graph=%s  %.2f us per iteration  (%s)" % (mode, graph, dt / 3000 * 1e6, plan.describe().split("iterate=")[1][:40]))
    plan.close()
