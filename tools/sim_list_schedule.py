"""Design model behind the list schedules of csrc/sqgr_autocorr.hip (k_bucket_order / _joint / _steps): LDS cycles per step and
`ds_read_b128` service group (16 lanes) for the two operand reads of the LDS-bucketed permutation dot, on random lists.

A pair of a list is (class of its Z row, class of its Y row), class = row index mod 16 = the 4-bank slot of the 16-byte row; a group
of 16 lanes is served in max(multiplicity of a class) cycles per operand.  The four orders:
  built     as the list builder leaves them (ascending i: both sides random)
  single    every list on its own: lane l' reads Z class (k + l') mod 16 at step k, surplus pairs into the holes (round 3)
  rotation  the same Z rotation, the 16 lanes choose their Y rows jointly; surplus pairs into the holes that collide least (round 4)
  steps     per step a matching lanes <-> Z classes (every `t` steps: the classes, fullest first, take the lane with the most pairs
            left) and lanes <-> Y classes (the classes in turn take the holder with the fewest alternatives) (round 4, the default)
Padding pairs are modelled as the device schedules them since round 4 (on banks nobody reads at that step: free); `--dumb-pads` makes
them read Z class `pad_z` and Y class 0 like rounds 1-3 did.  Usage: python tools/sim_list_schedule.py [--mu 250] [--groups 24]
Numbers quoted in DESIGN.md §3.3 come from this model; the measured kernel times are in profiles/r04_autocorr_experiments.json."""
from __future__ import annotations

import argparse

import numpy as np


def _cyc(classes) -> int:
    return int(np.bincount(np.asarray(classes, dtype=np.int64), minlength=16).max()) if len(classes) else 1


def _ctz(m: int) -> int:
    return (m & -m).bit_length() - 1


def _popc(m: int) -> int:
    return bin(m).count("1")


def random_group(rng, mu: float):
    """16 lanes' lists: Poisson(mu) pairs each, classes uniform."""
    return [np.stack([rng.integers(0, 16, n), rng.integers(0, 16, n)], 1) for n in rng.poisson(mu, 16)]


def _score(sched, rows, dumb_pads=False, pad_z=8):
    """sched[l][k] = (z, y) or None; cycles summed over the steps"""
    z_tot = y_tot = 0
    for k in range(rows):
        zs, ys, pads = [], [], 0
        for l in range(16):
            e = sched[l][k] if k < len(sched[l]) else None
            if e is None:
                pads += 1
            else:
                zs.append(e[0]); ys.append(e[1])
        if dumb_pads and pads:  # all padding lanes read ONE row each side (a broadcast), which still occupies its banks
            zs.append(pad_z); ys.append(0)
        z_tot += _cyc(zs); y_tot += _cyc(ys)
    return z_tot, y_tot


def order_built(ent, rows):
    return [[tuple(p) for p in e] + [None] * (rows - len(e)) for e in ent]


def order_single(ent, rows):
    out = []
    for l, e in enumerate(ent):
        n = len(e); D = (n + 15) // 16
        sched = [None] * rows
        pools = [[p for p in e if p[0] == z] for z in range(16)]
        surplus = []
        for z in range(16):
            for j, p in enumerate(pools[z]):
                if j < D: sched[((z - l) % 16) + 16 * j] = tuple(p)
                else: surplus.append(tuple(p))
        holes = [k for k in range(16 * D) if sched[k] is None]
        for k, p in zip(holes, surplus): sched[k] = p
        out.append(sched)
    return out


def order_rotation(ent, rows):
    n = [len(e) for e in ent]; D = [(x + 15) // 16 for x in n]
    cc = np.zeros((16, 16, 16), int)
    for l, e in enumerate(ent):
        if len(e): np.add.at(cc[l], (e[:, 0], e[:, 1]), 1)
    sched = [[None] * rows for _ in range(16)]
    ymask = [0] * rows; zmask = [0] * rows
    for o in range(16):
        order = sorted(range(16), key=lambda l: (cc[l, (o + l) % 16].sum(), l))
        for j in range(max(D) if D else 0):
            k = o + 16 * j; used = 0
            for l in order:
                if j >= D[l]: continue
                zc = (o + l) % 16
                avail = sum(1 << y for y in range(16) if cc[l, zc, y] > 0)
                if not avail: continue
                m = (avail & ~used) or avail
                s = (j * 5 + o * 3 + l * 7) % 16
                y = (_ctz(((m >> s) | (m << (16 - s))) & 0xFFFF) + s) % 16
                cc[l, zc, y] -= 1; sched[l][k] = (zc, y); used |= 1 << y; zmask[k] |= 1 << zc
            ymask[k] = used
    for l in range(16):
        rem = int(cc[l].sum())
        for lvl in range(3):
            for o2 in range(16):
                for j in range(D[l]):
                    k = o2 + 16 * j
                    if sched[l][k] is not None or rem == 0: continue
                    pick = None
                    for zc in range(16):
                        if lvl == 0 and (zmask[k] >> zc) & 1: continue
                        ys = [y for y in range(16) if cc[l, zc, y] > 0 and (lvl == 2 or not (ymask[k] >> y) & 1)]
                        if ys: pick = (zc, ys[0]); break
                    if pick is None: continue
                    cc[l, pick[0], pick[1]] -= 1; rem -= 1; sched[l][k] = pick
    return sched


def order_steps(ent, rows, t=4):
    cc = np.zeros((16, 16, 16), int)
    for l, e in enumerate(ent):
        if len(e): np.add.at(cc[l], (e[:, 0], e[:, 1]), 1)
    pool = cc.sum(2); rem = pool.sum(1).copy(); cdeg = pool.sum(0).copy()
    horizon = (int(rem.max()) + 15) // 16 * 16
    sched = [[None] * rows for _ in range(16)]
    keep = {}
    for k in range(horizon):
        if k % t == 0:
            keep = {}
            for z in sorted(range(16), key=lambda z: (-cdeg[z], z)):
                cand = [l for l in range(16) if l not in keep and pool[l, z] > 0]
                if cand: keep[max(cand, key=lambda l: (rem[l], -l))] = z
        cur = {l: z for l, z in keep.items() if pool[l, z] > 0}
        for l in range(16):
            if l not in cur and rem[l] > 0 and rem[l] + k >= horizon:
                cur[l] = _ctz(sum(1 << z for z in range(16) if pool[l, z] > 0))
        av = {l: sum(1 << y for y in range(16) if cc[l, cur[l], y] > 0) for l in cur}
        ym = {}
        for i in range(16):
            y = (i + k * 5) % 16
            cand = [l for l in cur if l not in ym and (av[l] >> y) & 1]
            if cand: ym[min(cand, key=lambda l: (_popc(av[l]), l))] = y
        for l in cur:
            y = ym.get(l, _ctz(av[l])); z = cur[l]
            cc[l, z, y] -= 1; pool[l, z] -= 1; rem[l] -= 1; cdeg[z] -= 1
            sched[l][k] = (z, y)
    assert rem.sum() == 0
    return sched


ORDERS = {"built": order_built, "single": order_single, "rotation": order_rotation, "steps": order_steps}


def simulate(order: str, mu: float = 250.0, groups: int = 24, seed: int = 5, dumb_pads: bool = False, **kw):
    """mean LDS cycles per step and group (Z side, Y side); `groups` lane groups, four to a bucket (its rows: the longest of 64 lists)"""
    rng = np.random.default_rng(seed)
    z_tot = y_tot = rows_tot = 0
    for _ in range(groups // 4):
        bucket = [random_group(rng, mu) for _ in range(4)]
        rows = (max(len(e) for g in bucket for e in g) + 15) // 16 * 16
        for g in bucket:
            sched = ORDERS[order](g, rows, **kw)
            for l, e in enumerate(g):  # every pair exactly once
                got = sorted(p for p in sched[l] if p is not None)
                assert got == sorted(map(tuple, e))
            z, y = _score(sched, rows, dumb_pads)
            z_tot += z; y_tot += y; rows_tot += rows
    return z_tot / rows_tot, y_tot / rows_tot


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mu", type=float, default=250.0)
    ap.add_argument("--groups", type=int, default=24)
    ap.add_argument("--dumb-pads", action="store_true")
    a = ap.parse_args()
    for name in ORDERS:
        z, y = simulate(name, a.mu, a.groups, dumb_pads=a.dumb_pads)
        print(f"{name:9s} Z {z:.3f}  Y {y:.3f}  sum {z + y:.3f} cycles per step and group")
