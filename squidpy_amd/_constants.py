"""Slot-name and mode constants of the ``sq.gr`` hot path.

Mirrors the contract of the reference's ``_constants`` package for this path only
(/root/reference/src/squidpy/_constants/_pkg_constants.py:103-121,198-213 — slot names;
_constants/_constants.py:93-110 — ``SpatialAutocorr`` / ``RipleyStat``; _constants/_utils.py:30-41 —
the error text of an invalid mode)."""

from __future__ import annotations

from enum import Enum
from typing import Any


class ModeEnum(str, Enum):
    """String enum whose invalid-value error lists the valid options (as the reference's ModeEnum does)."""

    @classmethod
    def _missing_(cls, value: Any) -> Any:
        raise ValueError(
            f"Invalid option `{value}` for `{cls.__name__}`. Valid options are: `{[m.value for m in cls]}`."
        )

    @property
    def s(self) -> str:
        return str(self.value)

    def __str__(self) -> str:
        return str(self.value)


class SpatialAutocorr(ModeEnum):
    MORAN = "moran"
    GEARY = "geary"


class RipleyStat(ModeEnum):
    F = "F"
    G = "G"
    L = "L"


class CorrAxis(ModeEnum):
    """Axis of the FDR correction in ``ligrec`` (_constants/_constants.py:21-23)."""

    INTERACTIONS = "interactions"
    CLUSTERS = "clusters"


class ComplexPolicy(ModeEnum):
    """Treatment of protein complexes in ``ligrec`` (_constants/_constants.py:27-29)."""

    MIN = "min"
    ALL = "all"


class Transform(Enum):
    """Adjacency transform of the graph builders (_constants/_constants.py:33-36).  A plain enum like the reference's — ``NONE``
    carries ``None`` — with its ``s`` / ``v`` accessors and its error text for an invalid option."""

    SPECTRAL = "spectral"
    COSINE = "cosine"
    NONE = None

    @classmethod
    def _missing_(cls, value: Any) -> Any:
        raise ValueError(f"Invalid option `{value}` for `{cls.__name__}`. Valid options are: `{[m.value for m in cls]}`.")

    @property
    def s(self) -> str:
        return str(self.value)

    @property
    def v(self) -> Any:
        return self.value

    def __str__(self) -> str:
        return str(self.value)


def _suffixed(value: str | None, suffix: str) -> str:
    """``None`` -> ``spatial_<suffix>``; ``"foo"`` -> ``"foo_<suffix>"``; an already suffixed key is kept."""
    stem = "spatial" if value is None else value
    return stem if stem.endswith("_" + suffix) else f"{stem}_{suffix}"


class _Obsm:
    spatial = "spatial"


class _Obsp:
    spatial_conn = staticmethod(lambda value=None: _suffixed(value, "connectivities"))
    spatial_dist = staticmethod(lambda value=None: _suffixed(value, "distances"))


class _Uns:
    spatial_neighs = staticmethod(lambda value=None: "spatial_neighbors" if value is None else f"{value}_neighbors")
    nhood_enrichment = staticmethod(lambda cluster: cluster + "_nhood_enrichment")
    interaction_matrix = staticmethod(lambda cluster: cluster + "_interactions")
    co_occurrence = staticmethod(lambda cluster: cluster + "_co_occurrence")
    ripley = staticmethod(lambda cluster, mode: f"{cluster}_ripley_{mode}")
    ligrec = staticmethod(lambda cluster, value=None: f"{cluster}_ligrec" if value is None else value)


class Key:
    """Names of the AnnData slots read / written by the hot path."""

    obsm = _Obsm
    obsp = _Obsp
    uns = _Uns
