"""Drop-in for `diff_gauss` (slothfulxtx/diff-gaussian-rasterization) as imported at
renderer/latent_gs_renderer.py:13-16: same names, call signature and return order
(image, depth, normal, alpha, radii, extra)."""
from .rasterizer import GaussianRasterizationSettings  # noqa: F401
from .rasterizer import GaussianRasterizerNormal as GaussianRasterizer  # noqa: F401
