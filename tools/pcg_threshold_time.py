"""Developer tool (GPU box): numpy-stream label shuffles at small and mid-size arrays — the wave kernel against the bucketed pipeline
(draw generator + replay), to place the switch between them (pcg_use_bucket in csrc/sqgr_pcg.hip)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph
from squidpy_amd._utils import pcg64_states
ctx = L.default_context()
for rows, cols, P in ((50, 100, 1000), (100, 100, 1000), (100, 200, 4000), (200, 200, 4000), (250, 320, 10000), (250, 400, 1000), (250, 400, 10000)):
    n = rows * cols
    adj = hex_grid_graph(rows, cols)
    labels = np.random.default_rng(0).integers(0, 20, n).astype(np.int32)
    g = L.Graph(ctx, adj, with_data=False)
    st = pcg64_states(0, P)
    out = {}
    for kern in ("wave", "bucket"):
        os.environ["SQGR_PCG_KERNEL"] = kern
        plan = L.NhoodPlan(ctx, g, labels, 20)
        plan.run_pcg64(st[:64]); plan.run_pcg64(st)
        ctx.timer_enable(True); ctx.timer_reset()
        t = time.perf_counter(); s1, s2, _ = plan.run_pcg64(st); dt = time.perf_counter() - t
        rep = {k.replace("nhood_pcg64_", ""): round(v[1], 2) for k, v in ctx.timer_report().items() if v[0] and "pcg64" in k}
        ctx.timer_enable(False)
        out[kern] = (s1.copy(), s2.copy())
        print(f"n={n} P={P} {kern}: {dt*1e3:.2f} ms {rep}", flush=True)
        plan.close()
    print("  equal:", bool((out["wave"][0] == out["bucket"][0]).all() and (out["wave"][1] == out["bucket"][1]).all()))
    g.close()
