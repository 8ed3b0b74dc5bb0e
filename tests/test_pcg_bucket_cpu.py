"""The re-ordering argument behind the device's bucketed replay of ``Generator.shuffle`` (csrc/sqgr_pcg.hip:
k_pcg_draws_bucketed + k_pcg_apply_bucketed), checked against numpy itself on the CPU through its numpy restatement
``oracle/pcg_bucket.py``: draws as numpy takes them, swaps applied phase by phase and RANGE BY RANGE with concurrent rounds."""

import numpy as np
import pytest

from oracle import pcg_bucket as B


@pytest.mark.parametrize("n,S,chunk", [(1, 64, 16), (2, 64, 16), (65, 64, 1024), (300, 64, 16), (5000, 256, 64), (5000, 1024, 1024), (70000, 4096, 1024),
                                       (200000, 65536, 1024)])
def test_bucketed_replay_equals_numpy_shuffle(n, S, chunk):
    seed = np.random.SeedSequence(n).spawn(3)[1]
    g_draw, g_ref = np.random.default_rng(seed), np.random.default_rng(seed)
    x = (np.arange(n) * 7919 % 251).astype(np.uint8)
    j = B.shuffle_draws(g_draw, n)
    ref = x.copy()
    g_ref.shuffle(ref)
    np.testing.assert_array_equal(B.apply_sequential(x, j), ref)      # the draws are numpy's
    stats = {}
    np.testing.assert_array_equal(B.apply_bucketed(x, j, S, chunk, stats=stats), ref)  # ... and range-major rounds leave numpy's array
    if n > 1:
        assert stats["rounds"] >= stats["chunks"] >= 1


def test_draws_leave_the_generator_where_numpy_leaves_it():
    """A library that ends in the middle of a 64-bit output leaves its high half buffered for the next one (`_shuffle_group` walks
    the libraries with ONE generator, gr/_utils.py:207-212): two consecutive shuffles from one generator."""
    seed = np.random.SeedSequence(5).spawn(1)[0]
    g_ref = np.random.default_rng(seed)
    a, b = np.arange(101), np.arange(57)
    g_ref.shuffle(a)
    g_ref.shuffle(b)
    # the restatement keeps its own 32-bit buffer per call, so it is replayed on one raw stream by hand
    bg = np.random.default_rng(seed).bit_generator
    buf = []

    def draws(n):
        out = []
        for i in range(n - 1, 0, -1):
            mask = i
            for s in (1, 2, 4, 8, 16):
                mask |= mask >> s
            while True:
                if not buf:
                    raw = int(bg.random_raw())
                    buf.extend([raw & 0xFFFFFFFF, raw >> 32])
                c = buf.pop(0) & mask
                if c <= i:
                    break
            out.append(c)
        return np.array(out, dtype=np.int64)

    np.testing.assert_array_equal(B.apply_bucketed(np.arange(101), draws(101), 64, 16), a)
    np.testing.assert_array_equal(B.apply_bucketed(np.arange(57), draws(57), 64, 16), b)
