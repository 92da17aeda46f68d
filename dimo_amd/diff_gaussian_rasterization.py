"""Drop-in for `diff_gaussian_rasterization` (ashawkey fork) as imported at
renderer/latent_gs_renderer.py:9-12: same names, call signature and return order
(image, radii, depth, alpha)."""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401
