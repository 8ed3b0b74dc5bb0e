"""Developer tool (GPU box): does the order of the observations matter to spatial_autocorr?  One 2048-gene block on the 1e5-spot hex
grid in scan order and in random order: preparation (Y = G Z gathers the neighbours' rows) and the permutation dot."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from squidpy_amd import _lib as L
import bench
ctx = L.default_context()
rows, cols, G, P = 250, 400, 2048, 1000
n = rows * cols
g = bench.autocorr_graph(ctx, "hex", rows, cols).tocoo()
vals = np.random.default_rng(1).gamma(2.0, 1.0, size=(G, n))
perm = np.random.default_rng(2).permutation(n); inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
shuf = sp.csr_matrix((g.data, (inv[g.row], inv[g.col])), shape=(n, n)); shuf.sort_indices()
for name, A, V in (("scan order", g.tocsr(), vals), ("random order", shuf, vals[:, perm])):
    graph = L.Graph(ctx, A, with_data=True)
    plan = L.AutocorrPlan(ctx, graph, V); plan.close()
    ctx.timer_enable(True); ctx.timer_reset()
    t = time.perf_counter()
    plan = L.AutocorrPlan(ctx, graph, V)
    sc = plan.scores("moran"); red = plan.perm_stats("moran", sc, seed=3, perm_begin=0, perm_end=P)
    dt = time.perf_counter() - t
    rep = {k: round(v[1], 2) for k, v in ctx.timer_report().items() if v[0]}
    ctx.timer_enable(False)
    print(name, "%.1f ms" % (dt * 1e3), rep, float(np.sort(sc)[:3].sum()))
    plan.close(); graph.close()
