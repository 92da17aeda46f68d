"""`distCUDA2(points[N,3]) -> [N]`: mean squared distance to the 3 nearest other points
(dimo_dist2 in dimo_amd/csrc/knn.hip).  No CPU fallback."""
import torch

from .. import _lib


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("dimo_amd.simple_knn needs GPU tensors (no CPU fallback in the product path)")
    pts = points.detach().float().contiguous()
    out = torch.empty(pts.shape[0], dtype=torch.float32, device=pts.device)
    _lib.check(_lib.lib().dimo_dist2(pts.shape[0], _lib.ptr(pts), _lib.ptr(out), _lib.current_stream()), "dimo_dist2")
    return out
