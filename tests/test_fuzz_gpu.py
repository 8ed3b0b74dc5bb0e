"""The randomised differential test of tools/fuzz_gpu.py (libsqgr against the CPU oracle over random graphs — symmetric /
directed / with self loops / non-canonical —, cluster counts across every kernel regime incl. 16-bit labels, libraries,
launch geometries, unaligned permutation ranges, both generators, co-occurrence, Ripley pair counts, numpy permutation
streams, ligrec; round 3: batched pair counts, cell-list kNN and its histograms, co-occurrence shards, expression formats x
column lists, the device p-value reductions) and of tools/fuzz_frontend.py (the `sq.gr.*` front ends with random options, value
sources and matrix formats against the oracle's restatement of the reference pipelines, in numpy's streams) with a FIXED seed list, so that a failure is reproducible (`FUZZ_ITERS=5 python tools/fuzz_gpu.py 0 <seed>`)."""
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


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_frontends_fixed_seeds(seed):
    env = dict(os.environ, PYTHONPATH=ROOT, FUZZ_ITERS="150")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_frontend.py"), "0", str(seed)], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "fuzz_frontend ok: 150 iterations" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


def test_fuzz_through_the_round4_kernel_variants():
    """The same two randomised testers with the kernel variants of round 4 FORCED (their sizes would pick the older kernels): the
    bucketed replay of numpy's shuffle with 128-position phases (libraries, tiny arrays, every cluster regime) and the LDS
    permutation kernel on 8 virtual permutations per permutation."""
    env = dict(os.environ, PYTHONPATH=ROOT, SQGR_PCG_KERNEL="bucket", SQGR_PCG_BUCKET_LOGS="7", SQGR_AUTOCORR_KERNEL="lds-split")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "0", "4242"], env=dict(env, FUZZ_ITERS="4"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "fuzz ok: 4 iterations" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_frontend.py"), "0", "4243"], env=dict(env, FUZZ_ITERS="120"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "fuzz_frontend ok: 120 iterations" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
