import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph
from squidpy_amd._utils import pcg64_states
ctx = L.default_context()
for rows, cols, P in ((250, 400, 1000), (250, 400, 16384), (1000, 1000, 1000), (1000, 1000, 16384), (1000, 1000, 65536)):
    n = rows * cols
    adj = hex_grid_graph(rows, cols)
    labels = np.random.default_rng(0).integers(0, 30, n).astype(np.int32)
    g = L.Graph(ctx, adj, with_data=False)
    plan = L.NhoodPlan(ctx, g, labels, 30)
    t = time.perf_counter(); st = pcg64_states(0, P); t_states = time.perf_counter() - t
    plan.run_pcg64(st[:64])
    ctx.timer_enable(True); ctx.timer_reset()
    t = time.perf_counter(); s1, s2, _ = plan.run_pcg64(st); dt = time.perf_counter() - t
    rep = {k: round(v[1], 1) for k, v in ctx.timer_report().items() if v[0]}
    ctx.timer_enable(False)
    print(f"n={n} P={P}: {P/dt:.0f} perms/s (host state prep {t_states:.2f}s) kernels_ms={rep}", flush=True)
    plan.close(); g.close()
