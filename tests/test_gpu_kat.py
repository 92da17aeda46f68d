"""GPU: the HIP rasterizer (C ABI) against the INDEPENDENT float64 torch-autograd restatement of the published
algorithm (oracle/raster_torch64.py) -- forward channels and gradients -- on small scenes, including the EWA clamp
branch and partial tiles.  Complements tests/test_gpu_raster.py, whose C oracle once shared helper code with the
kernels."""
import numpy as np
import pytest
import torch

from tests.scenes import camera_np, random_scene
from tests.test_gpu_raster import _run_hip
from tests.test_oracle_kat import _torch_render

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W,seed,off_axis", [(40, 48, 40, 1, False), (60, 32, 64, 2, False), (200, 96, 80, 5, True)])
def test_hip_rasterizer_vs_independent_float64_restatement(N, H, W, seed, off_axis):
    cam = camera_np(33.0 * seed, elevation=7 * seed, W=W, H=H)
    sc = random_scene(N, seed=seed, scale=0.06, anisotropy=0.8, opacity=(0.3, 0.99))
    if off_axis:  # spread wide enough that some Gaussians sit beyond 1.3 tan(fov/2): the clamp branch of the Jacobian
        sc["means3D"][: N // 4] *= 2.5
        sc["scales"][: N // 4] *= 3.0
    bg = (0.2, 0.5, 0.9)
    rng = np.random.default_rng(seed)
    gw = [rng.standard_normal(s).astype(np.float32) for s in ((3, H, W), (1, H, W), (3, H, W), (1, H, W))]
    out, gt = _torch_render(sc, cam, bg, [g.astype(np.float64) for g in gw])
    res, st, R, g = _run_hip(sc, cam, bg, 0, True, grads=[torch.tensor(x, device="cuda") for x in gw])
    n = lambda x: x.detach().cpu().numpy()
    assert np.array_equal(n(res["radii"]), out["geom"]["radius"].numpy())
    for k in ("image", "depth", "normal", "alpha"):
        a, b = n(res["color" if k == "image" else k]), out[k].detach().numpy()
        assert np.abs(a - b).mean() <= 1e-5 * max(1.0, np.abs(b).mean()), k
    assert (n(st["n_contrib"]).view(np.uint32).astype(np.int64) != out["n_contrib"].numpy()).mean() <= 5e-3
    for k in ("means3D", "scales", "rotations", "opacities", "shs"):
        a, b = n(g[k]).reshape(-1).astype(np.float64), gt[k].reshape(-1)
        err = np.abs(a - b).sum() / (np.abs(b).sum() + 1e-30)
        assert err <= 1e-4, (k, err)
    if off_axis:
        pv = np.concatenate([sc["means3D"], np.ones((N, 1))], 1) @ np.asarray(cam["view"], np.float64)
        vis = n(res["radii"]) > 0
        assert (vis & (np.abs(pv[:, 0] / pv[:, 2]) > 1.3 * cam["tanfovx"])).any(), "clamp branch not exercised"
