"""Module path of the reference (gridencoder/grid.py); the implementation lives in sanerf_hq_amd.ops."""
from ..ops import GridEncoder, _grid_encode, _host_offsets, grid_encode  # noqa: F401
