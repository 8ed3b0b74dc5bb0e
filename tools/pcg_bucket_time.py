"""Developer tool (GPU box): the numpy-stream label shuffle at 1e6 spots — bucketed replay vs the wave kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph
from squidpy_amd._utils import pcg64_states
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = L.default_context()
rows = cols = 1000
n = rows * cols
adj = hex_grid_graph(rows, cols)
labels = np.random.default_rng(0).integers(0, 30, n).astype(np.int32)
g = L.Graph(ctx, adj, with_data=False)
st = pcg64_states(0, P)
res = {}
for kern in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["bucket", "wave"]):
    os.environ["SQGR_PCG_KERNEL"] = kern.split(":")[0].replace("bucket-tags", "bucket")
    os.environ["SQGR_PCG_APPLY"] = "tags" if kern.startswith("bucket-tags") else "claims"   # replay kernel: rounds 4-5's hashed tags | exact claims
    os.environ["SQGR_PCG_DRAWS"] = "64" if kern.startswith("bucket-tags") else "128"        # generator: 64 | 128 draws per trip
    parts = kern.split(":")   # bucket[-tags][:logS[:tag slots]]
    for var, idx in (("SQGR_PCG_BUCKET_LOGS", 1), ("SQGR_PCG_BUCKET_SLOTS", 2)):
        if len(parts) > idx and parts[idx]:
            os.environ[var] = parts[idx]
        else:
            os.environ.pop(var, None)
    plan = L.NhoodPlan(ctx, g, labels, 30)
    plan.run_pcg64(st[:256])
    ctx.timer_enable(True); ctx.timer_reset()
    t = time.perf_counter(); s1, s2, _ = plan.run_pcg64(st); dt = time.perf_counter() - t
    rep = {k: round(v[1], 2) for k, v in ctx.timer_report().items() if v[0]}
    ctx.timer_enable(False)
    res[kern] = (s1.copy(), s2.copy())
    print(f"{kern}: n={n} P={P}: {P/dt:.0f} perms/s kernels_ms={rep}", flush=True)
    plan.close()
keys = list(res)
for k in keys[1:]:
    print(k, "== ", keys[0], bool((res[k][0] == res[keys[0]][0]).all() and (res[k][1] == res[keys[0]][1]).all()))
