"""squidpy_amd — MI355X-native implementation of Squidpy's ``sq.gr`` spatial-statistics hot path.

``import squidpy_amd as sq; sq.gr.nhood_enrichment(adata, ...)`` mirrors ``squidpy.gr`` for
``nhood_enrichment``, ``spatial_autocorr``, ``co_occurrence``, ``ripley`` (and ``interaction_matrix``).
All compute runs in ``libsqgr.so`` (hand-written HIP for gfx950); there is no CPU fallback."""

from . import gr
from ._anndata_lite import AnnDataLite
from ._dist import init as init_distributed
from ._dist import shutdown as shutdown_distributed
from ._lib import clear_graph_cache, trim_device_memory
from ._order import edge_locality, edge_span, spatial_order

__all__ = ["gr", "AnnDataLite", "init_distributed", "shutdown_distributed", "clear_graph_cache", "trim_device_memory", "edge_locality", "edge_span", "spatial_order"]
__version__ = "0.1.0"
