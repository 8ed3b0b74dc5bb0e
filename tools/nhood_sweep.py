"""Developer tool: time the nhood permutation kernels over the tuning space on the GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import restate as O
from squidpy_amd import _lib as L

rows = cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
ctx = L.default_context()
print(ctx.device_info())
adj = O.hex_grid_graph(rows, cols)
labels = np.random.default_rng(0).integers(0, K, rows * cols).astype(np.int32)
g = L.Graph(ctx, adj, with_data=False)
plan = L.NhoodPlan(ctx, g, labels, K)
bperm = 4 * adj.nnz + 4 * (adj.shape[0] + 1) + 2 * adj.shape[0]
for B in (16, 32):
    for nblk in (256,):
        for nbatch in (8, 32):
            plan.tune(B, nblk, nbatch)
            plan.run(1, 0, B * nbatch)  # warm
            ctx.timer_enable(True); ctx.timer_reset()
            t = time.perf_counter(); plan.run(1, 0, P); dt = time.perf_counter() - t
            rep = ctx.timer_report(); ctx.timer_enable(False)
            ks = {k: round(v[1], 2) for k, v in rep.items() if k.startswith("nhood")}
            t2 = time.perf_counter(); plan.run(1, 0, P); dt2 = time.perf_counter() - t2
            print(f"B={B} nblk={nblk} nbatch={nbatch}: {P/dt2:9.0f} perms/s (timed {P/dt:9.0f}) alg-frac={P/dt2*bperm/8e12:.3f} kernels_ms={ks}", flush=True)
