"""Regularisers of the reference's objective that live beside the render path:

  ARAP energy          `Renderer.arap_loss_v2`            renderer/latent_gs_renderer.py:1081-1094
                       `cal_connectivity_from_points_v2`  utils/deform_utils.py:115-141   (ball-query graph)
                       `estimate_rotation`                utils/deform_utils.py:161-197   (batched 3x3 SVD)
                       `cal_arap_error`                   utils/deform_utils.py:208-236
  geometry-anchor loss main_train_dimo.py:295-303         (chamferdist.ChamferDistance forward, or L1)
  farthest point sampling  main_train_dimo.py:511-515     (pytorch3d.ops.sample_farthest_points -> prune_points)

They run once per motion and step (M ~ 512 control points x 8 sampled times), off the per-render critical path:
device-agnostic tensor code on PyTorch-ROCm (rocSOLVER does the 3x3 SVDs), except farthest point sampling, which is
K sequential rounds and has its own kernel (`dimo_farthest_point_sample`).  `pytorch3d` and `chamferdist` are not
vendored by the reference: `ball_query`, `chamfer_forward` and the sampling restate their DOCUMENTED behaviour
(parity unpinned for exactly those three; everything built on top is pinned by tests/golden/arap.npz, which the
reference's own functions produced).
"""
import numpy as np
import torch
import torch.nn.functional as F


def ball_query(p1, p2, K, radius):
    """pytorch3d.ops.ball_query: for every point of p1 [T,N1,3] the FIRST K points of p2 [T,N2,3] (index order, not
    nearest) with squared distance < radius^2.  Returns (squared dists [T,N1,K] padded with 0, idx padded with -1)."""
    d2 = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
    within = d2 < radius * radius
    n2 = p2.shape[1]
    ar = torch.arange(n2, device=p1.device)
    key = torch.where(within, ar, ar + n2)  # hits keep their index, misses sort behind every hit
    k = min(K, n2)
    order = torch.topk(key, k, dim=-1, largest=False, sorted=True).values
    idx = torch.where(order < n2, order, torch.full_like(order, -1))
    dist = torch.gather(d2, -1, idx.clamp(min=0)) * (idx >= 0)
    if k < K:
        idx = F.pad(idx, (0, K - k), value=-1)
        dist = F.pad(dist, (0, K - k), value=0.0)
    return dist, idx


def cal_connectivity_from_points_v2(points, radius=0.1, K=10):
    """Edges (ii, jj, nn) of the ball-query graph that hold at EVERY time of `points` [T,Nv,3]: nn-th neighbour of
    vertex ii is vertex jj (utils/deform_utils.py:115-141, same tensor gymnastics, same uint8 neighbour count)."""
    Nv = points.shape[1]
    dev = points.device
    _, nn_idx = ball_query(points, points, K=10 + 1, radius=radius)
    nn_idx = nn_idx[:, :, 1:]
    hot = F.one_hot(nn_idx + 1, num_classes=Nv + 1).to(torch.bool)
    hot = hot.any(dim=2).all(dim=0).to(torch.float)
    hot[:, 0] = 0.0
    num_nonzero = hot.sum(dim=1).to(torch.uint8)
    _, top = torch.topk(hot, k=10, dim=1, largest=True)
    top = (top - 1).abs()
    ii = torch.arange(Nv, device=dev)[:, None].long().expand(Nv, K)
    nn = torch.arange(K, device=dev)[None].long().expand(Nv, K)
    mask = torch.arange(top.shape[1], device=dev).expand_as(top) < num_nonzero[:, None]
    return ii[mask], top[mask], nn[mask], None


def produce_edge_matrix_nfmt(verts, edge_shape, ii, jj, nn):
    """E[i, n] = p_i - p_(J[n])  (utils/deform_utils.py:37-44)."""
    E = torch.zeros(edge_shape, device=verts.device, dtype=verts.dtype)
    E[ii, nn] = verts[ii] - verts[jj]
    return E


def estimate_rotation(source, target, ii, jj, nn, K=10, weight=None, sample_idx=None):
    """Per-vertex best rotation source edges -> target edges by SVD of the weighted covariance, reflections
    repaired by flipping the column of the smallest singular value (utils/deform_utils.py:161-197)."""
    Nv = len(source)
    src = produce_edge_matrix_nfmt(source, (Nv, K, 3), ii, jj, nn)
    tgt = produce_edge_matrix_nfmt(target, (Nv, K, 3), ii, jj, nn)
    if weight is None:
        weight = torch.zeros(Nv, K, device=source.device)
        weight[ii, nn] = 1
    if sample_idx is not None:
        src, tgt = src[sample_idx], tgt[sample_idx]
    D = torch.diag_embed(weight, dim1=1, dim2=2)
    S = torch.bmm(src.permute(0, 2, 1), torch.bmm(D, tgt))
    unchanged = torch.unique(torch.where((src == tgt).all(dim=1))[0])
    S[unchanged] = 0
    U, sig, W = torch.svd(S)
    R = torch.bmm(W, U.permute(0, 2, 1))
    flip = torch.nonzero(torch.det(R) <= 0, as_tuple=False).flatten()
    if len(flip) > 0:
        Umod = U.clone()
        cols = torch.argmin(sig[flip], dim=1)
        Umod[flip, :, cols] *= -1
        R[flip] = torch.bmm(W[flip], Umod[flip].permute(0, 2, 1))
    return R


def cal_arap_error(nodes_sequence, ii, jj, nn, K=10, weight=None, sample_num=512):
    """sum_t sum_edges w ||e_t - R_t e_0||^2 against the first time of `nodes_sequence` [Nt,Nv,3]
    (utils/deform_utils.py:208-236; rotations carry no gradient)."""
    Nt, Nv, _ = nodes_sequence.shape
    dev = nodes_sequence.device
    if weight is None:
        weight = torch.zeros(Nv, K, device=dev)
        weight[ii, nn] = 1
    src = produce_edge_matrix_nfmt(nodes_sequence[0], (Nv, K, 3), ii, jj, nn)
    sample_idx = torch.arange(Nv, device=dev)
    if Nv > sample_num:
        sample_idx = torch.from_numpy(np.random.choice(Nv, sample_num)).long().to(dev)
    else:
        src = src[sample_idx]
    weight = weight[sample_idx]
    err = 0
    for t in range(1, Nt):
        with torch.no_grad():
            rot = estimate_rotation(nodes_sequence[0], nodes_sequence[t], ii, jj, nn, K=K, weight=weight,
                                    sample_idx=sample_idx)
        tgt = produce_edge_matrix_nfmt(nodes_sequence[t], (Nv, K, 3), ii, jj, nn)[sample_idx]
        rigid = torch.bmm(rot, src[sample_idx].permute(0, 2, 1)).permute(0, 2, 1)
        err = err + (weight * (torch.norm(tgt - rigid, dim=2) ** 2)).sum()
    return err


def arap_loss_v2(gaussians, stage="s1", latent_index=0, t_samp_num=8):
    """`Renderer.arap_loss_v2` (renderer/latent_gs_renderer.py:1081-1094): TimeNet at `t_samp_num` random times on
    the Gaussians (s1) / control points (s2), ball-query graph common to all times, ARAP energy."""
    g = gaussians
    means3D = (g._xyz if stage == "s1" else g._c_xyz)[None]
    q_times = torch.rand(t_samp_num).to(means3D.device)
    q_times = q_times[:, None, None].repeat(1, means3D.shape[1], 1)
    deform, _ = g._timenet(means3D, q_times, g.latent_code(latent_index), t_apply=True)
    means3D_t = means3D.repeat(t_samp_num, 1, 1).detach() + deform
    ii, jj, nn, _ = cal_connectivity_from_points_v2(means3D_t, K=10)
    return cal_arap_error(means3D_t, ii, jj, nn), (ii, jj, nn, None)


def chamfer_forward(source, target):
    """chamferdist.ChamferDistance()(source [B,N,3], target [B,M,3]) with its defaults: sum over the source points
    of the squared distance to their nearest target point, mean over the batch (main_train_dimo.py:299)."""
    d2 = ((source[:, :, None, :] - target[:, None, :, :]) ** 2).sum(-1)
    return d2.min(dim=2).values.sum(dim=1).mean()


def geometry_anchor_loss(cpts, cpts_ori, ga_chamfer=True, lambda_ga1=10.0, lambda_ga2=10000.0):
    """main_train_dimo.py:295-303: keeps the stage-2 control points near their stage-1 positions."""
    cpts_ori = cpts_ori.detach()
    if ga_chamfer:
        return lambda_ga1 * chamfer_forward(cpts[None], cpts_ori[None])
    return lambda_ga2 * (cpts - cpts_ori).abs().mean()


def sample_farthest_points(points, K):
    """pytorch3d.ops.sample_farthest_points(points [1,N,3], K) -> (points [1,K,3], idx [1,K]); HIP kernel, GPU only."""
    import ctypes as C  # noqa: F401

    from . import _lib
    if not points.is_cuda:
        raise RuntimeError("sample_farthest_points needs GPU tensors (no CPU fallback in the product path)")
    assert points.dim() == 3 and points.shape[0] == 1 and points.shape[2] == 3
    xyz = points[0].detach().float().contiguous()
    n = xyz.shape[0]
    K = min(int(K), n)
    idx = torch.empty(K, dtype=torch.int64, device=xyz.device)
    scratch = torch.empty(max(n, 1), dtype=torch.float32, device=xyz.device)
    _lib.check(_lib.lib().dimo_farthest_point_sample(n, K, _lib.ptr(xyz), _lib.ptr(scratch), _lib.ptr(idx),
                                                     _lib.current_stream()), "dimo_farthest_point_sample")
    return points[:, idx], idx[None]
