"""``squidpy_amd.gr`` — the MI355X-native ``sq.gr`` spatial-statistics hot path."""

from . import neighbors
from ._build import (
    SpatialNeighborsResult,
    spatial_neighbors,
    spatial_neighbors_delaunay,
    spatial_neighbors_from_builder,
    spatial_neighbors_grid,
    spatial_neighbors_knn,
    spatial_neighbors_radius,
)
from ._ligrec import PermutationTest, ligrec
from ._mask import MultiPolygon, Polygon, mask_graph
from ._nhood import NhoodEnrichmentResult, interaction_matrix, nhood_enrichment
from ._ppatterns import co_occurrence, spatial_autocorr
from ._ripley import ripley

__all__ = ["nhood_enrichment", "interaction_matrix", "NhoodEnrichmentResult", "co_occurrence", "spatial_autocorr", "ripley", "ligrec", "PermutationTest", "spatial_neighbors",
           "spatial_neighbors_knn", "spatial_neighbors_delaunay", "spatial_neighbors_radius", "spatial_neighbors_grid", "spatial_neighbors_from_builder",
           "neighbors", "SpatialNeighborsResult", "mask_graph", "Polygon", "MultiPolygon"]
