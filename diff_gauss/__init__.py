"""``from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer`` -- the import the
reference performs at gaussian_renderer/__init__.py:15, served by the MI355X implementation."""
from instag_amd.diff_gauss import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                   rasterize_gaussians)
