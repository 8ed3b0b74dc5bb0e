"""Drop-in check: the Python signatures of squidpy_amd.gr.* equal the reference's (positional order, defaults,
keyword-only ``table_key``); our additions are keyword-only with defaults.  The reference signatures were extracted by
tests/golden/make_signatures.py (AST, build container) into tests/golden/reference_signatures.json."""

from __future__ import annotations

import inspect
import json
import os

from types import MappingProxyType

import pytest

import squidpy_amd as sq

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "reference_signatures.json")))
EXTRA = {"rng", "device", "fma", "gene_block", "shard"}
# defaults the reference spells through its constants
SPECIAL = {
    "Key.obsp.spatial_conn()": "spatial_connectivities",
    "Key.obsm.spatial": "spatial",
    "ComplexPolicy.MIN.v": "min",
    "CorrAxis.CLUSTERS.v": "clusters",
}


def _norm(default_src: str | None):
    if default_src is None:
        return inspect.Parameter.empty
    if default_src in SPECIAL:
        return SPECIAL[default_src]
    return eval(default_src, {"MappingProxyType": MappingProxyType})  # literals only: numbers, strings, None, True/False


@pytest.mark.parametrize("name", sorted(REF))
def test_signature_matches_reference(name):
    ref = REF[name]
    sig = inspect.signature(getattr(sq.gr, name))
    params = list(sig.parameters.values())
    pos = [p for p in params if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert [p.name for p in pos] == [a["name"] for a in ref["positional"]], ref["file"]
    for p, a in zip(pos, ref["positional"]):
        assert p.default == _norm(a["default"]), (name, p.name, ref["file"])
    kwonly = {p.name: p for p in params if p.kind == p.KEYWORD_ONLY}
    for a in ref["keyword_only"]:
        assert a["name"] in kwonly and kwonly[a["name"]].default == _norm(a["default"])
    ours = set(kwonly) - {a["name"] for a in ref["keyword_only"]}
    assert ours <= EXTRA, ours
    assert all(kwonly[k].default is not inspect.Parameter.empty for k in ours)  # additions never change a call that omits them


def test_co_occurrence_deprecated_keywords_match_reference():
    deco = " ".join(REF["co_occurrence"]["decorators"])
    for kw in ("n_splits", "n_jobs", "backend", "show_progress_bar"):
        assert kw in deco  # the reference deprecates exactly these; ours warns on the same ones (tests/test_cooccur_gpu.py)
