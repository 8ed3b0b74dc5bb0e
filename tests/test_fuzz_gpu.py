"""The randomised differential test of tools/fuzz_gpu.py (libsqgr against the CPU oracle over random graphs — symmetric /
directed / with self loops / non-canonical —, cluster counts across every kernel regime incl. 16-bit labels, libraries,
launch geometries, unaligned permutation ranges, both generators, co-occurrence, Ripley pair counts, numpy permutation
streams, ligrec) with a FIXED seed list, so that a failure is reproducible (`FUZZ_ITERS=5 python tools/fuzz_gpu.py 0 <seed>`)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [20240924, 7, 1234567])
def test_fuzz_fixed_seeds(seed):
    env = dict(os.environ, PYTHONPATH=ROOT, FUZZ_ITERS="5")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "0", str(seed)], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "fuzz ok: 5 iterations" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
