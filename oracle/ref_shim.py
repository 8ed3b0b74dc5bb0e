"""TEST INFRASTRUCTURE — not product code.  Ground truth of record for the oracle.

Executes the reference's *own* kernel source (scverse/squidpy, mounted read-only at
``/root/reference``) under a numba stub, because the reference package itself cannot be
imported in this container (needs Python >= 3.12, numba, anndata, scanpy, spatialdata).

How: the reference file is parsed with :mod:`ast`; the requested top-level functions /
assignments are selected by name, decorators and annotations are dropped, and the code is
``exec``-ed in a namespace where ``njit`` is the identity decorator and ``prange`` is
``range``.  No reference source is copied into this repository: it is read where it lies.

Only usable where ``/root/reference`` exists (the build container).  It is used by
``tests/golden/make_golden.py`` to generate the committed golden vectors and by the CPU
tests that pin ``oracle/restate.py`` against the literal source.  It does not exist on the
GPU box, therefore nothing under ``-m gpu``, ``smoke()`` or ``bench.py`` imports it.

Reference symbols exposed (file:line in /root/reference/src/squidpy):
  gr/_nhood.py:54-141      ``_template``/``_create_function``  (generated count kernel)
  gr/_nhood.py:516-547     ``_nhood_enrichment_helper``
  gr/_nhood.py:412-429     ``_interaction_matrix``
  gr/_utils.py:185-213     ``_shuffle_group``
  _utils.py:240-241        ``spawn_generators``
  gr/_ppatterns.py:258-280 ``_score_helper``
  gr/_ppatterns.py:283-358 ``_occur_count``/``_co_occurrence_helper``
  gr/_ppatterns.py:431-559 ``_find_min_max``/``_p_value_calc``/``_analytic_pval``/``_g_moments``
  gr/_ripley.py:197-271    ``_reshape_res``/``_f_g_function``/``_l_function``/``_ppp``
  gr/_ligrec.py:616-775    ``_score_permutations``/``_analysis``
"""

from __future__ import annotations

import ast
import os
import re
import sys
import types
from contextlib import contextmanager
from enum import Enum
from typing import Any

import numpy as np

REF_ROOT = os.environ.get("SQGR_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src", "squidpy")


def available() -> bool:
    return os.path.isdir(REF_SRC)


# --------------------------------------------------------------------------- numba stub
class _Sig:
    """Stands in for numba type objects: ``dt[:, :](dt[:], ...)`` must evaluate."""

    def __getitem__(self, _item: Any) -> "_Sig":
        return self

    def __call__(self, *_a: Any, **_k: Any) -> "_Sig":
        return self


def _njit(*args: Any, **_kwargs: Any) -> Any:
    if len(args) == 1 and callable(args[0]) and not isinstance(args[0], _Sig):
        return args[0]
    return lambda f: f


def _make_numba_stub() -> tuple[types.ModuleType, types.ModuleType]:
    nb = types.ModuleType("numba")
    nt = types.ModuleType("numba.types")
    nb.njit = _njit  # type: ignore[attr-defined]
    nb.prange = range  # type: ignore[attr-defined]
    nb.types = nt  # type: ignore[attr-defined]
    for name in ("uint32", "int32", "float32", "float64", "boolean", "int64"):
        setattr(nt, name, _Sig())
    nt.UniTuple = _Sig()  # type: ignore[attr-defined]
    return nb, nt


@contextmanager
def _numba_stubbed():
    saved = {k: sys.modules.get(k) for k in ("numba", "numba.types")}
    nb, nt = _make_numba_stub()
    sys.modules["numba"], sys.modules["numba.types"] = nb, nt
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


# --------------------------------------------------------------------------- AST loader
class _Strip(ast.NodeTransformer):
    def visit_FunctionDef(self, node: ast.FunctionDef) -> ast.AST:
        self.generic_visit(node)
        node.decorator_list = []
        node.returns = None
        for a in node.args.args + node.args.kwonlyargs + node.args.posonlyargs:
            a.annotation = None
        if node.args.vararg:
            node.args.vararg.annotation = None
        if node.args.kwarg:
            node.args.kwarg.annotation = None
        return node

    def visit_AnnAssign(self, node: ast.AnnAssign) -> ast.AST:
        self.generic_visit(node)
        if node.value is None:
            return ast.Pass()
        return ast.copy_location(ast.Assign(targets=[node.target], value=node.value), node)


def _extract(relpath: str, names: list[str], namespace: dict[str, Any]) -> dict[str, Any]:
    path = os.path.join(REF_SRC, relpath)
    with open(path) as fh:
        src = fh.read()
    # PEP 695 `type X = ...` aliases do not parse on this interpreter (3.10): blank them, keeping line numbers
    src = re.sub(r"(?m)^type\s+\w+(\[[^\]]*\])?\s*=.*$", "pass", src)
    tree = ast.parse(src, filename=path)
    keep: list[ast.stmt] = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            keep.append(node)
        elif isinstance(node, ast.Assign):
            tg = [t.id for t in node.targets if isinstance(t, ast.Name)]
            if any(t in names for t in tg):
                keep.append(node)
    found = {n.name if isinstance(n, ast.FunctionDef) else n.targets[0].id for n in keep}  # type: ignore[union-attr]
    missing = set(names) - found
    if missing:
        raise RuntimeError(f"reference symbols {sorted(missing)} not found in {path}")
    mod = ast.Module(body=[_Strip().visit(n) for n in keep], type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), namespace)
    return namespace


class _Signal(Enum):
    NONE = 0
    UPDATE = 1
    FINISH = 2
    UPDATE_FINISH = 3


class _ModeStr:
    """`SpatialAutocorr.MORAN.s`-style constants used inside ``match`` statements."""

    class MORAN:
        s = "moran"

    class GEARY:
        s = "geary"

    def __init__(self, v: Any):
        self.value = str(v)


_cache: dict[str, dict[str, Any]] = {}


def nhood() -> dict[str, Any]:
    """Namespace with the reference's nhood kernels (gr/_nhood.py)."""
    if "nhood" in _cache:
        return _cache["nhood"]
    import pandas as pd

    ns: dict[str, Any] = {
        "np": np,
        "pd": pd,
        "njit": _njit,
        "prange": range,
        "dt": _Sig(),
        "ndt": np.uint32,
        "Signal": _Signal,
    }
    ns["_shuffle_group"] = utils()["_shuffle_group"]
    _extract("gr/_nhood.py", ["_template", "_create_function", "_nhood_enrichment_helper", "_interaction_matrix"], ns)
    orig_create = ns["_create_function"]

    def _create_function(n_cls: int, parallel: bool = False):
        with _numba_stubbed():
            return orig_create(n_cls, parallel)

    ns["create_function"] = _create_function
    _cache["nhood"] = ns
    return ns


def utils() -> dict[str, Any]:
    """``_shuffle_group`` (gr/_utils.py:185-213) and ``spawn_generators`` (_utils.py:240-241)."""
    if "utils" in _cache:
        return _cache["utils"]
    import pandas as pd

    ns: dict[str, Any] = {"np": np, "pd": pd}
    _extract("gr/_utils.py", ["_shuffle_group"], ns)
    _extract("_utils.py", ["spawn_generators"], ns)
    _cache["utils"] = ns
    return ns


def ppatterns(morans_i: Any = None, gearys_c: Any = None) -> dict[str, Any]:
    """Namespace with the reference's point-pattern helpers (gr/_ppatterns.py).

    ``morans_i`` / ``gearys_c`` are *not* in the reference tree (scanpy.metrics); pass the
    restatements from :mod:`oracle.restate` so that ``_score_helper`` can run literally.
    """
    key = f"ppatterns-{id(morans_i)}-{id(gearys_c)}"
    if key in _cache:
        return _cache[key]
    import pandas as pd
    from scipy import stats
    from scipy.sparse import spmatrix
    from sklearn.metrics import pairwise_distances

    class SpatialAutocorr(Enum):
        MORAN = "moran"
        GEARY = "geary"

        @property
        def s(self) -> str:
            return str(self.value)

    ns: dict[str, Any] = {
        "np": np,
        "pd": pd,
        "njit": _njit,
        "prange": range,
        "stats": stats,
        "spmatrix": spmatrix,
        "pairwise_distances": pairwise_distances,
        "fp": np.float32,
        "ip": np.int32,
        "Signal": _Signal,
        "SpatialAutocorr": SpatialAutocorr,
        "morans_i": morans_i,
        "gearys_c": gearys_c,
    }
    _extract(
        "gr/_ppatterns.py",
        [
            "_score_helper",
            "_occur_count",
            "_co_occurrence_helper",
            "_find_min_max",
            "_p_value_calc",
            "_analytic_pval",
            "_g_moments",
        ],
        ns,
    )
    _cache[key] = ns
    return ns


def ripley() -> dict[str, Any]:
    """Namespace with the reference's Ripley helpers (gr/_ripley.py:197-271)."""
    if "ripley" in _cache:
        return _cache["ripley"]
    import pandas as pd
    from scipy.spatial import ConvexHull, Delaunay
    from sklearn.neighbors import KDTree, NearestNeighbors

    ns: dict[str, Any] = {
        "np": np,
        "pd": pd,
        "ConvexHull": ConvexHull,
        "Delaunay": Delaunay,
        "KDTree": KDTree,
        "NearestNeighbors": NearestNeighbors,
    }
    _extract("gr/_ripley.py", ["_reshape_res", "_f_g_function", "_l_function", "_ppp"], ns)
    _cache["ripley"] = ns
    return ns


class _NullProgress:
    """Stands in for ``numba_progress.ProgressBar`` (context manager with ``update``)."""

    def __init__(self, *_a: Any, **_k: Any):
        pass

    def __enter__(self) -> "_NullProgress":
        return self

    def __exit__(self, *_a: Any) -> None:
        return None

    def update(self, _n: int) -> None:
        return None


@contextmanager
def _null_threads(_n: Any):
    yield


def ligrec() -> dict[str, Any]:
    """Namespace with the reference's ligrec permutation kernel and its driver (gr/_ligrec.py:616-775)."""
    if "ligrec" in _cache:
        return _cache["ligrec"]
    from collections import namedtuple

    import pandas as pd

    ns: dict[str, Any] = {
        "np": np,
        "pd": pd,
        "njit": _njit,
        "prange": range,
        "List": list,
        "ProgressBar": _NullProgress,
        "numba_threads": _null_threads,
        "TempResult": namedtuple("TempResult", ["means", "pvalues"]),
        "spawn_generators": utils()["spawn_generators"],
    }
    _extract("gr/_ligrec.py", ["_score_permutations", "_analysis"], ns)
    _cache["ligrec"] = ns
    return ns
