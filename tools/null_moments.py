"""Null distribution of the neighbourhood counts under the device generator against numpy's own shuffles, at a sample
size where a bias of a fraction of a percent of sigma would show (run on the GPU box):

    python tools/null_moments.py [n_perms]          # default 200000
    SQGR_LIBRARY=/path/to/libsqgr_6rounds.so python tools/null_moments.py

For every one of the K*K count cells: z of the difference of means, z of the difference of variances (normal theory),
and the same for the third central moment estimated from per-permutation counts of a subsample."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph
from squidpy_amd._utils import pcg64_states

P = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
side = int(os.environ.get("SIDE", 1000))
K = 30
ctx = L.default_context()
adj = hex_grid_graph(side, side)
n = adj.shape[0]
labels = np.random.default_rng(0).integers(0, K, n).astype(np.int32)
g = L.Graph(ctx, adj, with_data=False)
plan = L.NhoodPlan(ctx, g, labels, K)
freq = np.bincount(labels, minlength=K) / n
shift = np.rint(adj.nnz * np.outer(freq, freq)).astype(np.int64)

def moments(run):
    t = time.perf_counter(); s1, s2, _ = run(); dt = time.perf_counter() - t
    s1 = s1.astype(np.float64); s2 = s2.astype(np.float64)
    mean = s1 / P
    var = s2 / P - mean * mean
    return mean, var, dt

m_dev, v_dev, t_dev = moments(lambda: plan.run(20240924, 0, P, shift))
m_np, v_np, t_np = moments(lambda: plan.run_pcg64(pcg64_states(7, P), shift))
z_mean = (m_dev - m_np) / np.sqrt(v_dev / P + v_np / P)
z_var = (v_dev - v_np) / (0.5 * (v_dev + v_np) * np.sqrt(4.0 / P))
out = {
    "n": n, "K": K, "n_perms": P, "library": os.environ.get("SQGR_LIBRARY", "default"),
    "seconds": {"device_generator": t_dev, "numpy_streams": t_np},
    "z_mean": {"max_abs": float(np.abs(z_mean).max()), "rms": float(np.sqrt((z_mean ** 2).mean()))},
    "z_var": {"max_abs": float(np.abs(z_var).max()), "rms": float(np.sqrt((z_var ** 2).mean()))},
    "expected": "900 cells: rms ~ 1 (cells are correlated), max |z| ~ 3.3 under the null hypothesis of identical distributions",
    "sigma_over_mean_typical": float(np.sqrt(v_np).mean() / (m_np + shift).mean()),
}
print(json.dumps(out, indent=1))
