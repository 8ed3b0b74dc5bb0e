"""A minimal AnnData stand-in (anndata is not installable in the build image).

The ``sq.gr`` functions here duck-type their ``adata`` argument: anything exposing ``obs`` (DataFrame),
``obsm`` / ``obsp`` / ``uns`` / ``layers`` (mappings), ``X``, ``var`` / ``var_names``, ``shape`` and
``__getitem__`` column subsetting works — a real :class:`anndata.AnnData` does.  This class provides just
that for tests, ``bench.py`` and users without anndata."""

from __future__ import annotations

from typing import Any

import numpy as np
import pandas as pd
from scipy import sparse


class AnnDataLite:
    def __init__(
        self,
        X: Any = None,
        obs: pd.DataFrame | dict | None = None,
        var: pd.DataFrame | dict | None = None,
        obsm: dict | None = None,
        obsp: dict | None = None,
        uns: dict | None = None,
        layers: dict | None = None,
        raw: "AnnDataLite | None" = None,
    ):
        n_obs = None
        if X is not None:
            n_obs, n_var = X.shape
        obs = pd.DataFrame(obs) if obs is not None else pd.DataFrame(index=pd.RangeIndex(n_obs or 0).astype(str))
        if n_obs is None:
            n_obs = len(obs)
            n_var = 0 if var is None else len(pd.DataFrame(var))
        if not isinstance(obs.index[0] if len(obs) else "", str):
            obs.index = obs.index.astype(str)
        var = pd.DataFrame(var) if var is not None else pd.DataFrame(index=[f"g{i}" for i in range(n_var)])
        if len(var) == 0 and n_var:
            var = pd.DataFrame(index=[f"g{i}" for i in range(n_var)])
        self.X = X
        self.obs = obs
        self.var = var
        self.obsm = dict(obsm or {})
        self.obsp = dict(obsp or {})
        self.uns = dict(uns or {})
        self.layers = dict(layers or {})
        self.raw = raw

    @property
    def shape(self) -> tuple[int, int]:
        return (len(self.obs), len(self.var))

    @property
    def n_obs(self) -> int:
        return len(self.obs)

    @property
    def var_names(self) -> pd.Index:
        return self.var.index

    @property
    def obs_names(self) -> pd.Index:
        return self.obs.index

    def _var_indexer(self, sel: Any) -> np.ndarray:
        if isinstance(sel, slice):
            return np.arange(len(self.var))[sel]
        if isinstance(sel, pd.Series):
            sel = sel.to_numpy()
        sel = np.asarray(sel)
        if sel.dtype == bool:
            return np.where(sel)[0]
        if sel.dtype.kind in "iu":
            return sel.astype(np.int64)
        idx = self.var.index.get_indexer(np.atleast_1d(sel))
        if (idx < 0).any():
            missing = np.atleast_1d(sel)[idx < 0]
            raise KeyError(f"Values {list(missing)} are not valid var names.")
        return idx

    def __getitem__(self, key: Any) -> "AnnDataLite":
        if not isinstance(key, tuple) or len(key) != 2:
            raise IndexError("AnnDataLite supports only adata[obs_sel, var_sel]")
        osel, vsel = key
        vidx = self._var_indexer(vsel)
        if isinstance(osel, slice) and osel == slice(None):
            oidx = np.arange(self.n_obs)
        else:
            osel = osel.to_numpy() if isinstance(osel, pd.Series) else np.asarray(osel)
            oidx = np.where(osel)[0] if osel.dtype == bool else osel.astype(np.int64)

        all_obs = len(oidx) == self.n_obs and bool(np.array_equal(oidx, np.arange(self.n_obs)))
        v_range = len(vidx) > 0 and bool(np.array_equal(vidx, np.arange(vidx[0], vidx[0] + len(vidx))))

        def sub(m: Any) -> Any:
            if m is None:
                return None
            if sparse.issparse(m):
                m = m.tocsr()
                return m[:, vidx] if all_obs else m[oidx, :][:, vidx]
            m = np.asarray(m)
            if all_obs:  # like AnnData views: no copy for a contiguous run of variables (e.g. all genes)
                return m[:, vidx[0] : vidx[0] + len(vidx)] if v_range else m[:, vidx]
            return m[np.ix_(oidx, vidx)]

        return AnnDataLite(
            X=sub(self.X),
            obs=self.obs.iloc[oidx].copy(),
            var=self.var.iloc[vidx].copy(),
            obsm={k: np.asarray(v)[oidx] for k, v in self.obsm.items()},
            obsp={k: (v.tocsr()[oidx, :][:, oidx] if sparse.issparse(v) else np.asarray(v)[np.ix_(oidx, oidx)]) for k, v in self.obsp.items()},
            uns=dict(self.uns),
            layers={k: sub(v) for k, v in self.layers.items()},
            raw=self.raw,
        )

    def copy(self) -> "AnnDataLite":
        import copy as _copy

        return AnnDataLite(
            X=None if self.X is None else self.X.copy(),
            obs=self.obs.copy(),
            var=self.var.copy(),
            obsm={k: np.array(v, copy=True) for k, v in self.obsm.items()},
            obsp={k: v.copy() for k, v in self.obsp.items()},
            uns=_copy.deepcopy(self.uns),
            layers={k: v.copy() for k, v in self.layers.items()},
            raw=self.raw,
        )
