import os, sys, ctypes, numpy as np
sys.path.insert(0, os.getcwd())
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph
from squidpy_amd._utils import pcg64_states
ctx = L.default_context()
adj = hex_grid_graph(1000, 1000); n = adj.shape[0]
labels = np.random.default_rng(0).integers(0, 30, n).astype(np.int32)
g = L.Graph(ctx, adj, with_data=False)
plan = L.NhoodPlan(ctx, g, labels, 30)
P = 1024
plan.run_pcg64(pcg64_states(0, P))
out = (ctypes.c_ulonglong * 2)()
lib = ctypes.CDLL(os.environ["SQGR_LIBRARY"])
lib.sqgr_debug_pcg_rounds(out)
print("rounds per permutation: chunks", out[0] / P, "stream", out[1] / P)
