"""Seeded synthetic scenes shared by the parity tests (numpy, fp32/fp64)."""
import math

import numpy as np

from dimo_amd.camera import OrbitCamera, orbit_camera


def camera_np(azimuth=0.0, elevation=0.0, radius=2.0, W=128, H=128, fovy_deg=33.9, near=0.01, far=100.0):
    """Same matrices MiniCam builds (renderer/latent_gs_renderer.py:943-970), as float64 numpy."""
    oc = OrbitCamera(W, H, r=radius, fovy=fovy_deg, near=near, far=far)
    c2w = orbit_camera(elevation, azimuth, radius).astype(np.float64)
    w2c = np.linalg.inv(c2w)
    w2c[1:3, :3] *= -1
    w2c[:3, 3] *= -1
    view = w2c.T.copy()
    ty, tx = math.tan(oc.fovy / 2), math.tan(oc.fovx / 2)
    P = np.zeros((4, 4))
    P[0, 0], P[1, 1], P[3, 2] = 1 / tx, 1 / ty, 1.0
    P[2, 2], P[2, 3] = far / (far - near), -(far * near) / (far - near)
    proj = view @ P.T
    campos = -c2w[:3, 3]
    return dict(view=view, proj=proj, campos=campos, tanfovx=tx, tanfovy=ty, H=H, W=W)


def random_scene(N, seed=0, sh_coeffs=1, scale=0.03, opacity=(0.2, 0.95), radius=0.5, anisotropy=0.5):
    rng = np.random.default_rng(seed)
    u, v, w = rng.random(N), rng.random(N), rng.random(N)
    r = radius * np.cbrt(u)
    th, ph = np.arccos(2 * v - 1), 2 * np.pi * w
    xyz = np.stack([r * np.sin(th) * np.cos(ph), r * np.sin(th) * np.sin(ph), r * np.cos(th)], 1)
    scales = scale * np.exp(anisotropy * rng.standard_normal((N, 3)))
    q = rng.standard_normal((N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    op = rng.uniform(opacity[0], opacity[1], (N, 1))
    shs = rng.standard_normal((N, sh_coeffs, 3)) * 0.5
    shs[:, 0] += 0.8
    return dict(means3D=xyz, scales=scales, rotations=q, opacities=op, shs=shs)


TIMENET_SHAPES = (
    [("deformnet.0", 256, 104)] + [(f"deformnet.{i}", 256, 360 if i == 5 else 256) for i in range(1, 8)]
    + [("pts_layers.0", 256, 256), ("pts_layers.2", 3, 256), ("rot_layers.0", 256, 256), ("rot_layers.2", 4, 256)]
)


def timenet_weights(seed, head_std=1e-2):
    """Deterministic TimeNet state dict (names/shapes of renderer/latent_gs_renderer.py:184-203) from a numpy seed,
    so fixtures need not store 647k weights.  Xavier-uniform-like body, small random heads, rot bias [1,0,0,0]."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, o, i in TIMENET_SHAPES:
        if name in ("pts_layers.2", "rot_layers.2"):
            w = rng.standard_normal((o, i)) * head_std
            b = np.array([1.0, 0, 0, 0]) if o == 4 else np.zeros(3)
        else:
            a = np.sqrt(6.0 / (i + o))
            w = rng.uniform(-a, a, (o, i))
            b = rng.uniform(-0.05, 0.05, o)
        sd[name + ".weight"] = w.astype(np.float32)
        sd[name + ".bias"] = b.astype(np.float32)
    return sd


GRAD_STRIDE = 97  # golden fixtures keep grad.flatten()[::GRAD_STRIDE] of the TimeNet weights
