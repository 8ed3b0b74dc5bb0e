"""Golden vectors of `statsmodels.stats.multitest.multipletests` — the third-party call behind the reference's
`{pval}_{corr_method}` columns (gr/_ppatterns.py:239-245; ligrec's `_fdr_correct`, gr/_ligrec.py:840-870).

statsmodels is not importable from the interpreter the tests run on, but the image's conda python has 0.12.2:

    /opt/conda/bin/python3.9 tests/golden/make_multipletests_golden.py

writes tests/golden/multipletests_golden.json: for several p-value vectors (ties, zeros, ones, one NaN, length 1, a
400-vector) the corrected p-values of every method squidpy_amd implements.  tests/test_stats_cpu.py asserts them."""
import json
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from statsmodels.stats.multitest import multipletests  # noqa: E402
import statsmodels  # noqa: E402

METHODS = ["bonferroni", "sidak", "holm", "holm-sidak", "simes-hochberg", "hommel", "fdr_bh", "fdr_by", "fdr_gbs"]
rng = np.random.default_rng(2024)
cases = {
    "small": [0.01, 0.04, 0.03, 0.20, 0.5],
    "ties_zero_one": [0.0, 0.0, 1.0, 1.0, 0.5, 0.5, 0.25, 1e-300, 0.999999],
    "single": [0.37],
    "uniform_400": rng.uniform(size=400).tolist(),
    "tiny_values": (10.0 ** -rng.uniform(0, 30, size=64)).tolist(),
    "with_nan": [0.2, float("nan"), 0.01, 0.6],
}
out = {"statsmodels": statsmodels.__version__, "numpy": np.__version__, "methods": METHODS, "cases": {}}
for name, p in cases.items():
    rec = {"pvals": p, "corrected": {}}
    for m in METHODS:
        corr = multipletests(np.asarray(p, dtype=float), alpha=0.05, method=m)[1]
        rec["corrected"][m] = [None if np.isnan(v) else float(v) for v in corr]
    out["cases"][name] = rec
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multipletests_golden.json")
with open(path, "w") as fh:
    json.dump(out, fh)
print(path, {k: len(v["pvals"]) for k, v in out["cases"].items()})
