"""Developer tool (GPU box): per-kernel HIP-event times of one nhood permutation run at config 5's shape; used with the
count kernel's probe variants (SQGR_COUNT_DEBUG=1 no atomics, =2 no row gathers; results are then meaningless)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph

rows = cols = 1000
K, P = 30, 4096
ctx = L.default_context()
adj = hex_grid_graph(rows, cols)
labels = np.random.default_rng(0).integers(0, K, rows * cols).astype(np.int32)
g = L.Graph(ctx, adj, with_data=False)
plan = L.NhoodPlan(ctx, g, labels, K)
plan.run(1, 0, 1024)
ctx.timer_enable(True); ctx.timer_reset()
t = time.perf_counter(); res = plan.run(1, 0, P); dt = time.perf_counter() - t
rep = ctx.timer_report(); ctx.timer_enable(False)
print({k: round(v[1] / max(v[0], 1), 4) for k, v in rep.items() if k.startswith("nhood")}, f"{P/dt:.0f} perms/s", "sum of counts / P:", int(res[0].sum()) / P, "(edges:", adj.nnz, ")", flush=True)
