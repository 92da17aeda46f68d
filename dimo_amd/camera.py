"""Host-side camera math of the render path.

Mirrors the reference interface the trainer uses to drive `Renderer.render`:
  orbit_camera / look_at      utils/cam_utils.py:21-58
  OrbitCamera (fov, near/far) utils/cam_utils.py:61-76
  getProjectionMatrix         renderer/latent_gs_renderer.py:927-940
  MiniCam                     renderer/latent_gs_renderer.py:943-970

Conventions (row-vector): a world point is projected as
`[x y z 1] @ full_proj_transform`; `world_view_transform` is the transposed
world->camera matrix after the y/z flip; `camera_center = -c2w[:3, 3]`
(the sign is the reference's, kept for drop-in parity).

Unlike the reference, matrices are built once per (view, resolution) and cached
on the device -- the trainer renders the same 9 views thousands of times.
"""
import math

import numpy as np
import torch


def _unit(v, eps=1e-20):
    return v / np.sqrt(np.maximum(np.sum(v * v, axis=-1, keepdims=True), eps))


def look_at(campos, target, opengl=True):
    """Camera rotation [3,3] (columns right, up, forward)."""
    up = np.array([0, 1, 0], dtype=np.float32)
    if opengl:  # forward = +z
        fwd = _unit(campos - target)
        right = _unit(np.cross(up, fwd))
        up = _unit(np.cross(fwd, right))
    else:  # forward = -z
        fwd = _unit(target - campos)
        right = _unit(np.cross(fwd, up))
        up = _unit(np.cross(right, fwd))
    return np.stack([right, up, fwd], axis=1)


_POSES = {}


def orbit_camera(elevation, azimuth, radius=1, is_degree=True, target=None, opengl=True):
    """Elevation/azimuth -> camera-to-world pose [4,4] float32.  The reference calls this for every render
    (main_train_dimo.py:286) with one of its 9 azimuths: the poses are memoised (a copy is returned)."""
    key = (float(elevation), float(azimuth), float(radius), bool(is_degree), bool(opengl)) if target is None else None
    if key is not None:
        hit = _POSES.get(key)
        if hit is not None:
            return hit.copy()
    pose = _orbit_camera(elevation, azimuth, radius, is_degree, target, opengl)
    if key is not None:
        if len(_POSES) > 4096:
            _POSES.clear()
        _POSES[key] = pose.copy()
    return pose


def _orbit_camera(elevation, azimuth, radius, is_degree, target, opengl):
    if is_degree:
        elevation = np.deg2rad(elevation)
        azimuth = np.deg2rad(azimuth)
    pos = np.array([
        radius * np.cos(elevation) * np.sin(azimuth),
        -radius * np.sin(elevation),
        radius * np.cos(elevation) * np.cos(azimuth),
    ])
    if target is None:
        target = np.zeros([3], dtype=np.float32)
    pos = pos + target
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = look_at(pos, target, opengl)
    pose[:3, 3] = pos
    return pose


class OrbitCamera:
    """Only the intrinsics part of the reference class (the GUI orbit/pan part is out of scope)."""

    def __init__(self, W, H, r=2, fovy=60, near=0.01, far=100):
        self.W, self.H = W, H
        self.radius = r
        self.fovy = np.deg2rad(fovy)
        self.near, self.far = near, far

    @property
    def fovx(self):
        return 2 * np.arctan(np.tan(self.fovy / 2) * self.W / self.H)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    ty, tx = math.tan(fovY / 2), math.tan(fovX / 2)
    P = torch.zeros(4, 4)
    P[0, 0] = 1 / tx
    P[1, 1] = 1 / ty
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


_DEVICE_MATRICES = {}


class MiniCam:
    """renderer/latent_gs_renderer.py:943-970.  The reference builds one per render (main_train_dimo.py:286-287): a
    numpy inverse and four synchronous host-to-device copies, each of which waits for everything queued on the stream.
    The device matrices are therefore cached by VALUE (pose bytes, size, fov, planes, device): the same 9 views come
    back thousands of times, and a loop that constructs a MiniCam per render never touches the device after the first
    time it sees a view.

    The four device tensors are therefore SHARED by every MiniCam of the same pose / fov / planes and are READ-ONLY: an
    in-place edit (jitter, `.mul_`) would change every later camera of that view -- `clone()` the matrix first.  (A clone
    per construction would put four copy launches per render back on the reference loop's host path.)"""

    def __init__(self, c2w, width, height, fovy, fovx, znear, zfar, device=None):
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.image_width, self.image_height = width, height
        self.FoVy, self.FoVx = fovy, fovx
        self.znear, self.zfar = znear, zfar
        self.tanfovx = math.tan(fovx * 0.5)
        self.tanfovy = math.tan(fovy * 0.5)
        c2w = np.asarray(c2w)
        key = (c2w.tobytes(), str(c2w.dtype), float(fovy), float(fovx), float(znear), float(zfar), str(device))
        hit = _DEVICE_MATRICES.get(key)
        if hit is None:
            w2c = np.linalg.inv(c2w)
            w2c[1:3, :3] *= -1  # flip y and z rows of the rotation
            w2c[:3, 3] *= -1  # negate the translation
            wv = torch.tensor(w2c).transpose(0, 1)
            proj = getProjectionMatrix(znear=znear, zfar=zfar, fovX=fovx, fovY=fovy).transpose(0, 1)
            # all four are tiny: build on the host, upload once
            hit = (wv.contiguous().to(device), proj.contiguous().to(device),
                   (wv @ proj.to(wv.dtype)).contiguous().to(device), (-torch.tensor(c2w[:3, 3])).to(device))
            if len(_DEVICE_MATRICES) > 4096:
                _DEVICE_MATRICES.clear()
            _DEVICE_MATRICES[key] = hit
        self.world_view_transform, self.projection_matrix, self.full_proj_transform, self.camera_center = hit


class CameraCache:
    """(elevation, azimuth, radius, resolution) -> MiniCam, built once and kept on the device.

    The reference rebuilds the camera (numpy inverse + 3 H2D copies) for every
    render (main_train_dimo.py:286-287); with 16-128 renders per step that host
    work sits on the critical path."""

    def __init__(self, fovy_deg=33.9, near=0.01, far=100, device=None):
        self.fovy_deg, self.near, self.far, self.device = fovy_deg, near, far, device
        self._cams = {}

    def get(self, elevation, azimuth, radius, width, height):
        key = (float(elevation), float(azimuth), float(radius), int(width), int(height))
        cam = self._cams.get(key)
        if cam is None:
            oc = OrbitCamera(width, height, r=radius, fovy=self.fovy_deg, near=self.near, far=self.far)
            pose = orbit_camera(elevation, azimuth, radius)
            cam = MiniCam(pose, width, height, oc.fovy, oc.fovx, oc.near, oc.far, device=self.device)
            self._cams[key] = cam
        return cam
