import sys, time, numpy as np
sys.path.insert(0, ".")
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph
from squidpy_amd.gr._nhood import expected_counts
ctx = L.default_context()
adj = hex_grid_graph(1000, 1000); n = adj.shape[0]
g = L.Graph(ctx, adj, with_data=False)
rng = np.random.default_rng(0)
for name, labels in (("uniform", rng.integers(0, 30, n).astype(np.int32)), ("dirichlet(0.5)", rng.choice(30, size=n, p=rng.dirichlet(np.full(30, 0.5))).astype(np.int32))):
    plan = L.NhoodPlan(ctx, g, labels, 30)
    shift = expected_counts(labels, 30, int(adj.nnz))
    plan.run(3, 0, 10000, shift); ctx.sync()
    ctx.timer_enable(True); ctx.timer_reset()
    t = time.perf_counter()
    for i in range(3): s1, s2, _ = plan.run(3, 10000 * i, 10000 * (i + 1), shift)
    ctx.sync(); dt = (time.perf_counter() - t) / 3
    rep = {k: round(v[1] / 3, 2) for k, v in ctx.timer_report().items() if v[0]}
    ctx.timer_enable(False)
    print(name, "%.0f perms/s" % (10000 / dt), rep, int(s1.sum() % 1000003))
    plan.close()
