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


class Key:
    """Names of the AnnData slots read / written by the hot path."""

    class obsm:
        spatial = "spatial"

    class obsp:
        @staticmethod
        def _spatial_key(value: str | None, suffix: str) -> str:
            if value is None:
                return f"{Key.obsm.spatial}_{suffix}"
            if value.endswith(f"_{suffix}"):
                return value
            return f"{value}_{suffix}"

        @classmethod
        def spatial_dist(cls, value: str | None = None) -> str:
            return cls._spatial_key(value, "distances")

        @classmethod
        def spatial_conn(cls, value: str | None = None) -> str:
            return cls._spatial_key(value, "connectivities")

    class uns:
        @classmethod
        def nhood_enrichment(cls, cluster: str) -> str:
            return f"{cluster}_nhood_enrichment"

        @classmethod
        def interaction_matrix(cls, cluster: str) -> str:
            return f"{cluster}_interactions"

        @classmethod
        def co_occurrence(cls, cluster: str) -> str:
            return f"{cluster}_co_occurrence"

        @classmethod
        def ripley(cls, cluster: str, mode: str) -> str:
            return f"{cluster}_ripley_{mode}"
