"""MI355X implementation of the ADMM MAP-inference hot path of xiaocai00/ADMM.

The compute path is libadmm_hip.so (hand-written HIP kernels for gfx950 behind
the C ABI of include/admm_hip.h); this package is only the host-side mirror of
the reference's driver API.
"""
from .grid import (BLOCK, COL, EDGE, ROW, Grid, AdmmError, grid_edges, lcg_fill,
                   pairwise_table, partition, unique_id, communicate_group, edge_plan)

__all__ = ["Grid", "AdmmError", "ROW", "COL", "BLOCK", "EDGE", "grid_edges", "lcg_fill",
           "pairwise_table", "partition", "unique_id", "communicate_group", "edge_plan"]
