"""Developer tool (GPU box): does the streamed upload of a dense expression matrix (DeviceMatrix(stream_columns=...)) overlap the
feature blocks' kernels?  Prints when each column block has arrived and when each feature block starts and ends."""
import sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squidpy_amd import _lib
import bench
ctx = _lib.default_context()
n, G, P, W = 100000, 20000, 1000, 2048
X = np.empty((n, G), dtype=np.float64)
base = np.random.default_rng(3).gamma(2.0, 1.0, size=(n, 2000))
for j in range(G // 2000):
    np.add(base, 0.01 * j, out=X[:, j * 2000:(j + 1) * 2000])
g = bench.autocorr_graph(ctx, "hex", 250, 400)
graph = _lib.Graph(ctx, g, with_data=True)
blocks = [(b0, min(G, b0 + W)) for b0 in range(0, G, W)]
for streamed in ((True, True, False, True) if os.environ.get("DIAG_STREAM_FIRST") else (False, True, True)):
    ctx.sync()
    t0 = time.perf_counter()
    dm = _lib.DeviceMatrix(ctx, X, stream_columns=W if streamed else None)
    t_created = time.perf_counter() - t0
    log = []
    for b0, b1 in blocks:
        a0 = ctx.alloc_counters()
        ta = time.perf_counter() - t0
        plan = _lib.AutocorrPlan.from_columns(ctx, graph, dm, b0, b1 - b0)
        tb = time.perf_counter() - t0
        sc = plan.scores("moran")
        red = plan.perm_stats("moran", sc, seed=5, perm_begin=0, perm_end=P)
        plan.close()
        tc = time.perf_counter() - t0
        a1 = ctx.alloc_counters()
        log.append((round(ta * 1e3), round(tb * 1e3), round(tc * 1e3), dm._arrived, {k: round((a1[k] - a0[k]) / (1e6 if k.endswith("_ns") else 1), 1) for k in a1 if a1[k] != a0[k]}))
    dm.close()
    print("streamed" if streamed else "whole", "matrix ready after %.0f ms; per block (start, plan ready, done, columns arrived): %s; total %.0f ms" % (t_created * 1e3, log, (time.perf_counter() - t0) * 1e3), flush=True)
