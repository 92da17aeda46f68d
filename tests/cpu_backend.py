"""CPU stand-ins (oracle-backed) for the native ops, injected into the product's host logic by the
no-GPU tests and by the `cpu_baseline` leg of bench.py.  Never imported by dimo_amd/."""
import numpy as np
import torch

from oracle import raster_oracle as ro
from oracle.losses_ref import ssim_ref


class OracleRasterizer:
    def __init__(self, settings, with_normal):
        self.s, self.with_normal = settings, with_normal

    def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3Ds_precomp=None, cov3D_precomp=None, extra_attrs=None):
        cov = cov3Ds_precomp if cov3Ds_precomp is not None else cov3D_precomp
        image, depth, normal, alpha, radii = ro.rasterize_torch(means3D, means2D, shs, colors_precomp, opacities,
                                                                 scales, rotations, cov, self.s)
        if self.with_normal:
            return image, depth, normal, alpha, radii, None
        return image, radii, depth, alpha


def knn_cpu(ref, query, k):
    d, i = ro.knn(ref.detach().numpy(), query.detach().numpy(), k)
    return torch.from_numpy(d), torch.from_numpy(i)


def dist2_cpu(pts):
    return torch.from_numpy(ro.dist2(pts.detach().cpu().numpy()))


def fps_cpu(xyz, K):
    from oracle.regularizers_ref import farthest_point_sample_ref
    return torch.from_numpy(farthest_point_sample_ref(xyz.detach().numpy(), min(int(K), xyz.shape[0])))


def make_cpu_trainer(cfg, rank=0, world=1, pg=None, regime="trained"):
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import Trainer
    rd = Renderer(sh_degree=cfg.sh_degree, white_background=True, radius=cfg.radius,
                  num_latent_code=cfg.num_motions, latent_code_dim=cfg.latent_code_dim, add_normal=cfg.add_normal,
                  vae_latent=cfg.vae_latent, device="cpu", rasterizer_factory=OracleRasterizer, dist2_fn=dist2_cpu)
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=cfg.seed, regime=regime, num_latent=cfg.num_motions)
    if cfg.stage == "s1":  # the shared log-radius create_from_pcd makes (renderer/latent_gs_renderer.py:449-451)
        g = rd.gaussians
        g._r = torch.nn.Parameter(g._scaling.detach().mean() * torch.ones(1, 1))
    return Trainer(cfg, rd, rank=rank, world_size=world, process_group=pg, ssim_fn=ssim_ref, knn_fn=knn_cpu,
                   fps_fn=fps_cpu)
