"""Drop-in package for `simple_knn` (camenduru/simple-knn): `from simple_knn._C import distCUDA2`
(renderer/latent_gs_renderer.py:17,426)."""
