"""TEST INFRASTRUCTURE — restatement in numpy of the *device's* bucketed replay of ``Generator.shuffle``
(``csrc/sqgr_pcg_bucket.hip``), so that its re-ordering argument can be checked on the CPU against numpy itself.

numpy's shuffle (reference semantics: ``gr/_nhood.py:533-538`` -> ``Generator.shuffle``) is the reverse Fisher-Yates
``for i = n-1 .. 1: j = random_interval(i); swap(x[i], x[j])`` with masked rejection sampling over 32-bit halves of the PCG64
outputs (low half first).  The device splits it in two:

* **G** — the draws ``j_i`` are a function of the stream alone; they are generated in time order and appended, per *phase*
  (``S`` consecutive steps: the steps whose ``i`` lies in ``[f*S, (f+1)*S)``) and per *range* of ``j`` (``r = j // S``), to
  time-ordered lists.
* **A** — the swaps of a phase are applied RANGE BY RANGE instead of in time order: first the range ``r == f`` (both sides of
  the swap inside the phase's own window), then every range ``r < f``; inside a range in time order, in chunks of ``C`` records
  that are applied concurrently in rounds: a record may go in a round only if it is the earliest pending record of its chunk that
  touches its ``j`` slot (hashed) and — in the window range — no earlier pending record writes to its ``i`` slot.

Why range-major order is exact: position ``i`` is read only by step ``i`` itself and by earlier steps ``e > i`` with
``j_e == i`` (which lie in range ``f`` and are applied first, in order); no later step ever touches ``i`` again (``j_e <= e < i``).
So after the window range has been replayed, ``x[i]`` holds exactly what step ``i`` will move out, whatever happens in the
other ranges, and steps of different ranges ``r < f`` touch disjoint ``j`` positions."""

from __future__ import annotations

import numpy as np


def shuffle_draws(bitgen_state_source: np.random.Generator, n: int) -> np.ndarray:
    """``j_i`` for ``i = n-1 .. 1`` exactly as ``Generator.shuffle`` of an ``n``-array draws them (the generator is advanced
    the same way, up to the buffered 32-bit half).  Returns ``j`` with ``j[t]`` belonging to step ``i = n-1-t``."""
    bg = bitgen_state_source.bit_generator
    out = np.empty(max(n - 1, 0), dtype=np.int64)
    buf: list[int] = []
    t = 0
    i = n - 1
    while i >= 1:
        mask = i
        for s in (1, 2, 4, 8, 16):
            mask |= mask >> s
        while True:
            if not buf:
                raw = int(bg.random_raw())
                buf = [raw & 0xFFFFFFFF, raw >> 32]
            c = buf.pop(0) & mask
            if c <= i:
                break
        out[t] = c
        t += 1
        i -= 1
    return out


def apply_sequential(x: np.ndarray, j: np.ndarray) -> np.ndarray:
    x = x.copy()
    n = len(x)
    for t, jj in enumerate(j):
        i = n - 1 - t
        x[i], x[jj] = x[jj], x[i]
    return x


def apply_bucketed(x: np.ndarray, j: np.ndarray, S: int, chunk: int = 1024, slots: int = 4096, stats: dict | None = None) -> np.ndarray:
    """Kernel A's order of application (see the module docstring).  ``stats`` collects the number of rounds."""
    x = x.copy()
    n = len(x)
    i_of = n - 1 - np.arange(len(j))
    F = (n + S - 1) // S
    rounds_total = chunks_total = 0
    for f in range(F - 1, -1, -1):
        sel = np.nonzero((i_of // S) == f)[0]          # the phase's steps, in time order
        if len(sel) == 0:
            continue
        rng_of = j[sel] // S
        for r in [f] + list(range(f)):
            rec = sel[rng_of == r]                     # time-ordered list of bucket (f, r)
            internal = r == f
            for c0 in range(0, len(rec), chunk):
                ch = rec[c0:c0 + chunk]
                ii, jj = i_of[ch], j[ch]
                shift = 32 - int(np.log2(slots))  # the device's multiplicative hash of the position inside its range
                hj = ((jj % S) * 2654435761 % (1 << 32)) >> shift
                hi = ((ii % S) * 2654435761 % (1 << 32)) >> shift
                pending = np.ones(len(ch), dtype=bool)
                chunks_total += 1
                while pending.any():
                    rounds_total += 1
                    tag = np.full(slots, 1 << 30)
                    lanes = np.nonzero(pending)[0]
                    np.minimum.at(tag, hj[lanes], lanes)
                    go = pending & (tag[hj] == np.arange(len(ch)))
                    if internal:
                        go &= tag[hi] >= np.arange(len(ch))
                    g = np.nonzero(go)[0]
                    assert len(g) > 0
                    # all records of a round at once: gather both sides, then scatter (what concurrent lanes do)
                    a, b = x[ii[g]].copy(), x[jj[g]].copy()
                    x[ii[g]] = b
                    x[jj[g]] = a
                    # a self swap (i == j) reads and writes one cell twice with the same value: nothing to fix
                    pending[g] = False
    if stats is not None:
        stats["rounds"] = rounds_total
        stats["chunks"] = chunks_total
    return x
