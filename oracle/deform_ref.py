"""CPU ORACLE (test infrastructure): torch restatement of the stage-s2 skinning block of
Renderer.render, renderer/latent_gs_renderer.py:1187-1219 with helpers :112-147 (build_rotation_3d,
quat_mul) and the activations :341-351,382-383,1219.  Works in float32 or float64; gradients come from
autograd.  Pinned by tests/golden/deform_latent.npz / deform_vae.npz, which were produced by running the
reference's own render() with a capturing rasterizer stub (tests/golden/make_golden.py).
Never imported by the product path."""
import torch


def _rot_from_unit_quat(q):
    w, x, y, z = q.unbind(-1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, -1).reshape(q.shape[:-1] + (3, 3))


def skinning_ref(xyz, rotation, scaling, opacity, c_xyz, c_log_radius, d_xyz, d_rot, nn_dist, nn_idx,
                 local_frame=True):
    """-> (pts3D [N,3], unit rotations [N,4], exp(scaling) [N,3], sigmoid(opacity) [N,1])."""
    r = torch.exp(c_log_radius).reshape(-1)[nn_idx]                      # [N,k]
    w = torch.exp(-1.0 * nn_dist ** 2 / (2.0 * r ** 2)) + 1e-7
    w = w / w.abs().sum(dim=1, keepdim=True).clamp_min(1e-12)            # F.normalize(p=1)
    c, dc, dq = c_xyz[nn_idx], d_xyz[nn_idx], d_rot[nn_idx]              # [N,k,3] [N,k,3] [N,k,4]
    if local_frame:
        R = _rot_from_unit_quat(dq / dq.norm(dim=-1, keepdim=True))
        y = (R @ (xyz[:, None] - c)[..., None]).squeeze(-1) + c + dc
        pts = (w[..., None] * y).sum(1)
    else:
        pts = xyz + (w[..., None] * dc).sum(1)
    a = (w[..., None] * dq).sum(1)
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = rotation.unbind(-1)
    q = torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return pts, q, torch.exp(scaling), torch.sigmoid(opacity)
