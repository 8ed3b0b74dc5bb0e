"""Developer tool (GPU box): a resident feature block scored in a loop while a background thread uploads a dense matrix in column
blocks (sqgr_matrix_upload_columns on the copy stream) — wall time and HIP-event kernel time of every iteration."""
import sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squidpy_amd import _lib
import bench
ctx = _lib.default_context()
n, G, P, W = 100000, 20000, 1000, 2048
X = np.ones((n, G), dtype=np.float64)
g = bench.autocorr_graph(ctx, "hex", 250, 400)
graph = _lib.Graph(ctx, g, with_data=True)
vals = np.random.default_rng(1).gamma(2.0, 1.0, size=(W, n))
plan = _lib.AutocorrPlan(ctx, graph, vals)
sc = plan.scores("moran")
plan.perm_stats("moran", sc, seed=5, perm_begin=0, perm_end=P)
def loop(tag, iters=8):
    out = []
    for it in range(iters):
        ctx.timer_enable(True); ctx.timer_reset()
        t = time.perf_counter()
        plan.perm_stats("moran", sc, seed=5, perm_begin=0, perm_end=P)
        dt = (time.perf_counter() - t) * 1e3
        k = sum(v[1] for v in ctx.timer_report().values())
        ctx.timer_enable(False)
        out.append((round(dt, 1), round(k, 1)))
    print(tag, "(wall ms, kernel ms):", out, flush=True)
loop("alone")
t0 = time.perf_counter()
dm = _lib.DeviceMatrix(ctx, X, stream_columns=W)
loop("during the streamed upload")
dm.wait_columns()
print("upload finished %.0f ms after it began" % ((time.perf_counter() - t0) * 1e3))
loop("after")
dm.close()
