import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from squidpy_amd import _lib as L
from squidpy_amd._utils import pcg64_states
ctx = L.default_context()
bad = 0
for n in (2, 3, 5, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 1000, 4097, 70000, 300001):
    P = 70 if n < 100000 else 8
    st = pcg64_states(n, P)
    got = L.pcg64_permutations(ctx, n, st)
    gens = [np.random.default_rng(s) for s in np.random.SeedSequence(n).spawn(P)]
    want = np.stack([g.permutation(n) for g in gens])
    ok = np.array_equal(got, want)
    bad += not ok
    print(n, ok, flush=True)
    if not ok:
        r = np.where((got != want).any(axis=1))[0]
        print("  rows differing:", r[:10], "first col", np.where(got[r[0]] != want[r[0]])[0][:5])
print("BAD" if bad else "ALL OK")
