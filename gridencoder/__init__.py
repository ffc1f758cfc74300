"""``from gridencoder import GridEncoder`` -- the import the reference performs at encoding.py:63
(gridencoder/__init__.py:1), served by the MI355X implementation."""
from instag_amd.gridencoder import GridEncoder, grid_encode  # noqa: F401
