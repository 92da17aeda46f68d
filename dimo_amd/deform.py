"""Deformation stage of `Renderer.render`: latent-conditioned TimeNet on the control points,
then KNN-weighted linear-blend skinning (LBS) of every Gaussian.

Mirrors (same names / argument meaning / state-dict keys) the reference's
  TimeNet                      renderer/latent_gs_renderer.py:184-245
  get_embedder / Embedder      src/pos_enc.py:6-54          (sin/cos per frequency, no raw input)
  build_rotation(_3d)          renderer/latent_gs_renderer.py:90-133
  quat_mul                     renderer/latent_gs_renderer.py:135-147   (w, x, y, z Hamilton product)
  LBS block                    renderer/latent_gs_renderer.py:1191-1209
so that checkpoints (`timenet.pth`) load unchanged.  Device-agnostic PyTorch: the MLP's
GEMMs run on rocBLAS/hipBLASLt under PyTorch-ROCm (M = 512 rows: latency-, not MFMA-bound).
"""
import torch
import torch.nn.functional as F
from torch import nn


class Embedder(nn.Module):
    """NeRF positional encoding; output order: for each frequency, sin(all dims) then cos(all dims)."""

    def __init__(self, multires, input_dims):
        super().__init__()
        self.register_buffer("freq_bands", 2.0 ** torch.linspace(0.0, multires - 1, steps=multires), persistent=False)
        self.out_dim = 2 * multires * input_dims

    def forward(self, x):
        xf = x.unsqueeze(-2) * self.freq_bands.to(x.dtype).unsqueeze(-1)  # [..., F, D]
        return torch.stack((torch.sin(xf), torch.cos(xf)), dim=-2).flatten(-3)  # [..., F*2*D]


def get_embedder(multires, input_dims, i=0):
    if i == -1:
        return nn.Identity(), 3
    e = Embedder(multires, input_dims)
    return e, e.out_dim


def _xavier(m):
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight, gain=1)


class TimeNet(nn.Module):
    """(control point, time, latent code) -> (delta xyz [.,3], delta quaternion [.,4])."""

    def __init__(self, D=8, W=256, skips=(4,), latent_code_dim=32, device=None):
        super().__init__()
        self.pts_ch, self.times_ch = 10, 6
        self.pts_emb_fn, pts_dims = get_embedder(self.pts_ch, 3)
        self.times_emb_fn, t_dims = get_embedder(self.times_ch, 1)
        self.input_ch = pts_dims + t_dims + latent_code_dim
        self.skips = list(skips)
        self.deformnet = nn.ModuleList(
            [nn.Linear(self.input_ch, W)]
            + [nn.Linear(W + self.input_ch, W) if i in self.skips else nn.Linear(W, W) for i in range(D - 1)])
        self.pts_layers = nn.Sequential(nn.Linear(W, W), nn.ReLU(), nn.Linear(W, 3))
        self.rot_layers = nn.Sequential(nn.Linear(W, W), nn.ReLU(), nn.Linear(W, 4))
        self.apply(_xavier)
        with torch.no_grad():  # zero motion / identity rotation at initialisation
            self.pts_layers[-1].weight.zero_()
            self.pts_layers[-1].bias.zero_()
            self.rot_layers[-1].weight.zero_()
            self.rot_layers[-1].bias.copy_(torch.tensor([1.0, 0.0, 0.0, 0.0]))
        if device is not None:
            self.to(device)

    def forward(self, pts, t, latent_code, nobatch=False, t_apply=False):
        if pts.dim() == 2:
            nobatch = True
            pts = pts.unsqueeze(0)
        if t_apply:  # t: [T, N, 1] per-point times, pts broadcast over T
            times = t
            pts = pts.expand(times.shape[0], -1, -1)
        else:
            times = torch.full((1, pts.shape[1], 1), float(t), dtype=pts.dtype, device=pts.device)
        if latent_code.dim() == 1:
            latent_code = latent_code.expand(pts.shape[0], pts.shape[1], -1)
        emb = torch.cat([self.pts_emb_fn(pts), self.times_emb_fn(times), latent_code], dim=-1)
        h = emb
        for i, layer in enumerate(self.deformnet):
            h = F.relu(layer(h))
            if i in self.skips:
                h = torch.cat([emb, h], dim=-1)
        pts_t, rot_t = self.pts_layers(h), self.rot_layers(h)
        if nobatch:
            pts_t, rot_t = pts_t[0], rot_t[0]
        return pts_t, rot_t

    def get_mlp_parameters(self):
        rot = [p for n, p in self.named_parameters() if n.split(".")[0] == "rot_layers"]
        rest = [p for n, p in self.named_parameters() if n.split(".")[0] != "rot_layers"]
        return rest, rot


def build_rotation(r):
    """Unit-normalises q = (w, x, y, z) [..., 4] and returns rotation matrices [..., 3, 3]."""
    q = r / torch.sqrt((r * r).sum(-1, keepdim=True))
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


build_rotation_3d = build_rotation  # the reference keeps a second copy for [N, K, 4] inputs


def quat_mul(q1, q2):
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack([
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)


class _FusedSkinning(torch.autograd.Function):
    """dimo_deform_forward / dimo_deform_backward (dimo_amd/csrc/deform.hip): the whole stage-s2 skinning
    block plus the exp / sigmoid / normalize activations as one HIP kernel per direction."""

    @staticmethod
    def forward(ctx, xyz, rotation, scaling, opacity, c_xyz, c_log_radius, d_xyz, d_rot, nn_dist, nn_idx,
                local_frame):
        from . import _lib
        L = _lib.lib()
        c = lambda t: t.detach().float().contiguous()
        xyz, rotation, scaling, opacity = c(xyz), c(rotation), c(scaling), c(opacity)
        c_xyz, c_log_radius, d_xyz, d_rot = c(c_xyz), c(c_log_radius), c(d_xyz), c(d_rot)
        nn_dist, nn_idx = c(nn_dist), nn_idx.contiguous()
        N, M = xyz.shape[0], c_xyz.shape[0]
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=xyz.device)
        o_xyz, o_rot, o_scales, o_opac = new(N, 3), new(N, 4), new(N, 3), new(N, 1)
        _lib.check(L.dimo_deform_forward(
            N, M, int(bool(local_frame)), _lib.ptr(xyz), _lib.ptr(rotation), _lib.ptr(scaling), _lib.ptr(opacity),
            _lib.ptr(c_xyz), _lib.ptr(c_log_radius), _lib.ptr(d_xyz), _lib.ptr(d_rot), _lib.ptr(nn_dist),
            _lib.ptr(nn_idx), _lib.ptr(o_xyz), _lib.ptr(o_rot), _lib.ptr(o_scales), _lib.ptr(o_opac),
            _lib.current_stream()), "dimo_deform_forward")
        ctx.save_for_backward(xyz, rotation, scaling, opacity, c_xyz, c_log_radius, d_xyz, d_rot, nn_dist, nn_idx)
        ctx.local_frame = bool(local_frame)
        return o_xyz, o_rot, o_scales, o_opac

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_scales, g_opac):
        from . import _lib
        L = _lib.lib()
        xyz, rotation, scaling, opacity, c_xyz, c_lr, d_xyz, d_rot, nn_dist, nn_idx = ctx.saved_tensors
        N, M = xyz.shape[0], c_xyz.shape[0]
        dev = xyz.device
        z = lambda g, *s: (torch.zeros(*s, dtype=torch.float32, device=dev) if g is None else g.float().contiguous())
        g_xyz, g_rot, g_scales, g_opac = z(g_xyz, N, 3), z(g_rot, N, 4), z(g_scales, N, 3), z(g_opac, N, 1)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d = [new(N, 3), new(N, 4), new(N, 3), new(*opacity.shape), new(M, 3), new(*c_lr.shape), new(M, 3), new(M, 4)]
        scratch = torch.empty(L.dimo_deform_backward_scratch_bytes(N, M), dtype=torch.uint8, device=dev)
        _lib.check(L.dimo_deform_backward(
            N, M, int(ctx.local_frame), 0, _lib.ptr(xyz), _lib.ptr(rotation), _lib.ptr(scaling), _lib.ptr(opacity),
            _lib.ptr(c_xyz), _lib.ptr(c_lr), _lib.ptr(d_xyz), _lib.ptr(d_rot), _lib.ptr(nn_dist), _lib.ptr(nn_idx),
            _lib.ptr(g_xyz), _lib.ptr(g_rot), _lib.ptr(g_scales), _lib.ptr(g_opac), *[_lib.ptr(t) for t in d],
            _lib.ptr(scratch), scratch.numel(), _lib.current_stream()), "dimo_deform_backward")
        return (*d, None, None, None)


def fused_skinning_available(xyz, c_xyz, nn_idx):
    if not xyz.is_cuda or nn_idx is None or nn_idx.shape[-1] != 4:
        return False
    from . import _lib
    return 0 < c_xyz.shape[0] <= _lib.lib().dimo_deform_max_ctrl_points()


def fused_skinning(xyz, rotation, scaling, opacity, c_xyz, c_log_radius, d_xyz, d_rot, nn_dist, nn_idx,
                   local_frame=True):
    """(pts3D [N,3], unit rotations [N,4], exp(scaling) [N,3], sigmoid(opacity) [N,1]) on the GPU."""
    return _FusedSkinning.apply(xyz, rotation, scaling, opacity, c_xyz, c_log_radius, d_xyz, d_rot, nn_dist, nn_idx,
                                local_frame)


def lbs_weights(neighbor_dists, c_radius_n, eps=1e-7):
    """w = L1-normalise(exp(-d^2 / (2 r^2)) + eps) over the k neighbours (latent_gs_renderer.py:1193-1199)."""
    w = torch.exp(-1.0 * neighbor_dists ** 2 / (2.0 * (c_radius_n[:, :, 0] ** 2))) + eps
    return F.normalize(w, p=1)


def lbs_deform(means3D, rotations, c_means3D, c_radius, means3D_deform, rots_deform, neighbor_dists,
               neighbor_indices, local_frame=True):
    """Skin every Gaussian to its k nearest control points (latent_gs_renderer.py:1191-1209).

    Returns (pts3D [N,3], rotations [N,4] = quat_mul(sum_k w_k dq_k, rotations), not yet normalised)."""
    w = lbs_weights(neighbor_dists, c_radius[neighbor_indices])
    c_n = c_means3D[neighbor_indices]  # [N,k,3]
    dc_n = means3D_deform[neighbor_indices]  # [N,k,3]
    dq_n = rots_deform[neighbor_indices]  # [N,k,4]
    if local_frame:
        local = (build_rotation(dq_n) @ (means3D[:, None] - c_n)[..., None]).squeeze(-1)
        pts3D = (w[..., None] * (local + c_n + dc_n)).sum(dim=1)
    else:
        pts3D = means3D + (w[..., None] * dc_n).sum(dim=1)
    rots3D = (w[..., None] * dq_n).sum(dim=1)
    return pts3D, quat_mul(rots3D, rotations)
