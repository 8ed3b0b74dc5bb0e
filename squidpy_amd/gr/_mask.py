"""``mask_graph`` — drop-in for ``squidpy.gr.mask_graph`` (gr/_build.py:852-954): keep (or remove) the edges of a spatial
graph that lie fully inside a polygon.  Host code, like the reference (which goes through geopandas/shapely:
``LineString(coords[i], coords[j]).within(polygon_mask)``); the predicate is restated here with numpy so that the function
needs neither package.  ``polygon_mask`` may be a shapely ``Polygon`` / ``MultiPolygon`` (anything with ``geom_type`` and
``exterior.coords`` / ``interiors`` / ``geoms``) or the light :class:`Polygon` / :class:`MultiPolygon` of this module.

``within`` (OGC, DE-9IM ``T*F**F***``): no point of the segment lies in the polygon's exterior and some point of the
segment lies in the polygon's open interior.  Evaluated exactly in that form: the segment is cut at every parameter where it
meets a ring edge, and the midpoint of every piece (and both end points) is classified as inside / on the boundary / outside.
Parity note: the arithmetic is plain float64 cross products, GEOS uses exact predicates — the two agree unless a graph edge
touches the polygon boundary to within rounding; the reference's own test (tests/graph/test_spatial_neighbors.py:388-456)
is ported in tests/test_mask_graph_cpu.py."""

from __future__ import annotations

from typing import Any, Sequence

import numpy as np
from scipy import sparse

from .._constants import Key
from .._utils import _save_data, extract_adata_if_sdata

__all__ = ["mask_graph", "Polygon", "MultiPolygon", "segments_within"]


class _Ring:
    def __init__(self, coords: Any):
        c = np.asarray(coords, dtype=np.float64).reshape(-1, 2)
        if len(c) < 3:
            raise ValueError("A ring needs at least 3 points.")
        if not np.array_equal(c[0], c[-1]):
            c = np.vstack([c, c[:1]])
        self.coords = c


class Polygon:
    """Minimal polygon (shell + holes) with the attribute names of ``shapely.Polygon`` that :func:`mask_graph` reads."""

    geom_type = "Polygon"

    def __init__(self, shell: Any, holes: Sequence[Any] | None = None):
        self.exterior = _Ring(shell)
        self.interiors = [_Ring(h) for h in (holes or [])]


class MultiPolygon:
    """Minimal collection of polygons (``shapely.MultiPolygon``'s ``geoms``)."""

    geom_type = "MultiPolygon"

    def __init__(self, polygons: Sequence[Any]):
        self.geoms = list(polygons)


def _polygons(mask: Any) -> list[list[np.ndarray]]:
    """-> one list of closed rings (shell first) per polygon."""
    kind = getattr(mask, "geom_type", None)
    if kind == "Polygon":
        parts = [mask]
    elif kind == "MultiPolygon":
        parts = list(mask.geoms)
    else:
        raise ValueError(f"`polygon_mask` should be of type `Polygon` or `MultiPolygon`, got {type(mask)}")
    out = []
    for p in parts:
        rings = [np.asarray(p.exterior.coords, dtype=np.float64)[:, :2]] + [np.asarray(r.coords, dtype=np.float64)[:, :2] for r in p.interiors]
        rings = [r if np.array_equal(r[0], r[-1]) else np.vstack([r, r[:1]]) for r in rings]
        out.append(rings)
    return out


def _cross(ax: np.ndarray, ay: np.ndarray, bx: np.ndarray, by: np.ndarray) -> np.ndarray:
    return ax * by - ay * bx


def _classify(pts: np.ndarray, polys: list[list[np.ndarray]]) -> np.ndarray:
    """2 = strictly inside some polygon, 1 = on a boundary (and inside none), 0 = outside all.  pts: (n, 2)."""
    x, y = pts[:, 0][:, None], pts[:, 1][:, None]
    best = np.zeros(len(pts), dtype=np.int8)
    for rings in polys:
        inside_shell = None
        on_edge = np.zeros(len(pts), dtype=bool)
        in_hole = np.zeros(len(pts), dtype=bool)
        for k, ring in enumerate(rings):
            px, py = ring[:-1, 0][None, :], ring[:-1, 1][None, :]
            qx, qy = ring[1:, 0][None, :], ring[1:, 1][None, :]
            # on an edge: collinear and inside its bounding box
            cr = _cross(qx - px, qy - py, x - px, y - py)
            on = (cr == 0) & (x >= np.minimum(px, qx)) & (x <= np.maximum(px, qx)) & (y >= np.minimum(py, qy)) & (y <= np.maximum(py, qy))
            on_edge |= on.any(axis=1)
            # crossing number (half-open rule on y)
            straddle = (py > y) != (qy > y)
            with np.errstate(divide="ignore", invalid="ignore"):
                xi = px + (y - py) * (qx - px) / (qy - py)
            odd = (straddle & (x < xi)).sum(axis=1) % 2 == 1
            if k == 0:
                inside_shell = odd
            else:
                in_hole |= odd
        cls = np.where(on_edge, 1, np.where(inside_shell & ~in_hole, 2, 0)).astype(np.int8)
        best = np.maximum(best, cls)
    return best


def segments_within(a: np.ndarray, b: np.ndarray, polygon_mask: Any, chunk: int = 65536) -> np.ndarray:
    """``LineString([a_k, b_k]).within(polygon_mask)`` for every row k (see the module docstring)."""
    polys = _polygons(polygon_mask)
    a = np.asarray(a, dtype=np.float64).reshape(-1, 2)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 2)
    edges = np.concatenate([np.stack([r[:-1], r[1:]], axis=1) for rings in polys for r in rings])  # (R, 2, 2)
    p, q = edges[:, 0], edges[:, 1]
    out = np.zeros(len(a), dtype=bool)
    for s in range(0, len(a), chunk):
        aa, bb = a[s : s + chunk], b[s : s + chunk]
        n = len(aa)
        ca, cb = _classify(aa, polys), _classify(bb, polys)
        mid = _classify(0.5 * (aa + bb), polys)
        # parameters t in [0, 1] where the segment meets a ring edge
        dx, dy = (bb - aa)[:, 0][:, None], (bb - aa)[:, 1][:, None]
        ex, ey = (q - p)[:, 0][None, :], (q - p)[:, 1][None, :]
        wx, wy = p[:, 0][None, :] - aa[:, 0][:, None], p[:, 1][None, :] - aa[:, 1][:, None]
        den = _cross(dx, dy, ex, ey)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = _cross(wx, wy, ex, ey) / den
            u = _cross(wx, wy, dx, dy) / den
        hit = (den != 0) & (t >= 0) & (t <= 1) & (u >= 0) & (u <= 1)
        # parallel and collinear: the ring edge's end points projected on the segment
        col = (den == 0) & (_cross(wx, wy, dx, dy) == 0)
        touched = hit.any(axis=1) | col.any(axis=1)
        res = np.zeros(n, dtype=bool)
        # no contact with any ring: the whole segment has the class of its midpoint
        free = ~touched
        res[free] = mid[free] == 2
        for k in np.nonzero(touched)[0]:
            if ca[k] == 0 or cb[k] == 0:
                continue
            ts = [0.0, 1.0] + list(t[k][hit[k]])
            if col[k].any():
                d2 = float(dx[k, 0] ** 2 + dy[k, 0] ** 2)
                for r in np.nonzero(col[k])[0]:
                    for pt in (p[r], q[r]):
                        tt = ((pt[0] - aa[k, 0]) * dx[k, 0] + (pt[1] - aa[k, 1]) * dy[k, 0]) / d2 if d2 > 0 else 0.0
                        if 0.0 <= tt <= 1.0:
                            ts.append(tt)
            ts = np.unique(np.asarray(ts, dtype=np.float64))
            if len(ts) == 1:  # zero-length segment: a point
                res[k] = ca[k] == 2
                continue
            tm = 0.5 * (ts[:-1] + ts[1:])
            cls = _classify(aa[k][None, :] + tm[:, None] * (bb[k] - aa[k])[None, :], polys)
            res[k] = bool((cls > 0).all() and (cls == 2).any())
        out[s : s + chunk] = res
    return out


def mask_graph(
    sdata: Any,
    table_key: str,
    polygon_mask: Any,
    negative_mask: bool = False,
    spatial_key: str = Key.obsm.spatial,
    key_added: str = "mask",
    copy: bool = False,
) -> tuple[sparse.csr_matrix, sparse.csr_matrix] | None:
    """Mask the graph based on a polygon mask (drop-in for ``squidpy.gr.mask_graph``, gr/_build.py:852-954).

    Only the edges fully contained in the polygon(s) are kept (``negative_mask=True``: only those are removed).  Reads
    ``obsp['{spatial_key}_connectivities' | '{spatial_key}_distances']`` and ``obsm[spatial_key]`` of the table; ``copy=True``
    returns ``(connectivities, distances)``, otherwise writes ``obsp['{key_added}_{spatial_key}_connectivities']``,
    ``obsp['{key_added}_{spatial_key}_distances']`` and ``uns['{key_added}_{spatial_key}_neighbors']`` like the reference."""
    neighs_key = Key.uns.spatial_neighs(spatial_key)
    conns_key = Key.obsp.spatial_conn(spatial_key)
    dists_key = Key.obsp.spatial_dist(spatial_key)
    _polygons(polygon_mask)  # type check first, like the reference
    table = extract_adata_if_sdata(sdata, table_key=table_key)
    coords = np.asarray(table.obsm[spatial_key], dtype=np.float64)
    # (the reference edits the stored matrices in place and stores the same objects under the new keys; copies here)
    adj = sparse.csr_matrix(table.obsp[conns_key]).copy()
    dst = sparse.csr_matrix(table.obsp[dists_key]).copy()
    rows = np.repeat(np.arange(adj.shape[0]), np.diff(adj.indptr))
    cols = adj.indices
    within = segments_within(coords[rows, :2], coords[cols, :2], polygon_mask)
    drop = within if negative_mask else ~within
    r, c = rows[drop], cols[drop]
    if len(r):
        adj = adj.tolil()
        dst = dst.tolil()
        adj[r, c] = 0
        dst[r, c] = 0
        adj = adj.tocsr()
        dst = dst.tocsr()
    adj.eliminate_zeros()
    dst.eliminate_zeros()
    mask_conns_key = f"{key_added}_{conns_key}"
    mask_dists_key = f"{key_added}_{dists_key}"
    mask_neighs_key = f"{key_added}_{neighs_key}"
    neighbors_dict = {
        "connectivities_key": mask_conns_key,
        "distances_key": mask_dists_key,
        "unfiltered_graph_key": conns_key,
        "params": {"negative_mask": negative_mask, "table_key": table_key},
    }
    if copy:
        return adj, dst
    _save_data(table, attr="obsp", key=mask_conns_key, data=adj)
    _save_data(table, attr="obsp", key=mask_dists_key, data=dst)
    _save_data(table, attr="uns", key=mask_neighs_key, data=neighbors_dict)
    return None
