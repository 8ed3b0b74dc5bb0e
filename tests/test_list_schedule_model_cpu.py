"""The design model of the LDS kernel's list schedules (tools/sim_list_schedule.py; DESIGN §3.3 quotes its numbers): every order is a
permutation of every lane's pairs (asserted inside `simulate`), and the orders rank as the measured kernel times do —
lists as built > round 3's per-list order > the joint rotation > the step schedule —, the step schedule only with padding pairs on
free banks (with rounds 1-3's padding pair it loses to the rotation: measured 54.0 vs 51.9 ms before the zero rows went in)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("sim_list_schedule", os.path.join(ROOT, "tools", "sim_list_schedule.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_orders_rank_like_the_measured_kernel_times():
    m = _model()
    cyc = {name: m.simulate(name, mu=250.0, groups=4, seed=11) for name in m.ORDERS}
    total = {k: z + y for k, (z, y) in cyc.items()}
    assert total["built"] > total["single"] > total["rotation"] > total["steps"], total
    assert 5.0 < total["built"] < 6.0 and 2.2 < total["steps"] < 2.9, total
    assert cyc["single"][0] < 1.7 and cyc["single"][1] > 2.4          # one side scheduled, the other random
    assert cyc["steps"][0] < cyc["rotation"][0] and cyc["steps"][1] < cyc["rotation"][1]
    dumb = {name: sum(m.simulate(name, mu=250.0, groups=4, seed=11, dumb_pads=True)) for name in ("rotation", "steps")}
    assert dumb["steps"] > dumb["rotation"] > total["rotation"], dumb   # the padding pair's banks matter as much as the schedule


def test_step_schedule_survives_ragged_groups():
    m = _model()
    import numpy as np

    rng = np.random.default_rng(3)
    for lens in ([0] * 16, [1] + [0] * 15, [300] + [5] * 15, list(range(16)), [64] * 16):
        ent = [np.stack([rng.integers(0, 16, n), rng.integers(0, 16, n)], 1) for n in lens]
        rows = (max(lens) + 15) // 16 * 16 or 16
        for name in ("single", "rotation", "steps"):
            sched = m.ORDERS[name](ent, rows)
            for l, e in enumerate(ent):
                assert sorted(p for p in sched[l] if p is not None) == sorted(map(tuple, e)), (name, lens)
    # one lane whose pairs all share a cell: nothing to schedule, nothing lost
    ent = [np.zeros((40, 2), int)] + [np.stack([rng.integers(0, 16, 30), rng.integers(0, 16, 30)], 1) for _ in range(15)]
    sched = m.ORDERS["steps"](ent, 48)
    assert sum(p is not None for p in sched[0]) == 40
