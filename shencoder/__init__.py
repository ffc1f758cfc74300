"""``from shencoder import SHEncoder`` -- the import the reference performs at encoding.py:59,
served by the MI355X implementation."""
from instag_amd.shencoder import SHEncoder, sh_encode  # noqa: F401
