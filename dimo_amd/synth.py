"""Seeded synthetic inputs of the benchmark / parity configurations.

No dataset is available offline (the "Trump n51" example is a Google-Drive download,
README.md:44-46), so the measured workload uses Gaussians initialised exactly the way the
reference initialises them, random-init TimeNet weights of the reference architecture, and
seeded random target images of the reference's shapes (51 motions x 9 views x 21 frames).
"""
import math

import numpy as np
import torch
from torch import nn

from .gaussian_model import RGB2SH, inverse_sigmoid


def ball_points(n, radius, rng):
    """Uniform in a ball: r = R cbrt(u), theta = arccos(2u-1), phi = 2 pi u (latent_gs_renderer.py:999-1007)."""
    phis = rng.random(n) * 2 * np.pi
    costheta = rng.random(n) * 2 - 1
    thetas = np.arccos(costheta)
    r = radius * np.cbrt(rng.random(n))
    return np.stack((r * np.sin(thetas) * np.cos(phis), r * np.sin(thetas) * np.sin(phis), r * np.cos(thetas)), 1)


REGIMES = ("init", "low", "trained")


def init_synthetic_model(renderer, num_pts, num_cpts, seed=0, regime="trained", num_latent=51):
    """Fills renderer.gaussians with the SURVEY 8d synthetic state (stage-2 layout: Gaussians + control points).
    `regime` sets the opacities: "init" = every Gaussian at 0.05, what the reference creates them with
    (renderer/latent_gs_renderer.py:431) and stage s2 starts from (:1038-1058) -- nothing saturates, every pixel looks
    through its tile's whole list; "low" = U(0.01, 0.1); "trained" = sigmoid(U(-2, 4)), lists saturate early."""
    g = renderer.gaussians
    dev = g.device
    rng = np.random.default_rng(seed)
    tg = torch.Generator().manual_seed(seed)
    xyz = torch.tensor(ball_points(num_pts, 0.5, rng), dtype=torch.float32, device=dev)
    dist2 = torch.clamp_min(g._dist2(xyz), 1e-7)
    scales = torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3)
    rots = torch.zeros(num_pts, 4)
    rots[:, 0] = 1
    rots = rots + 0.1 * torch.randn(num_pts, 4, generator=tg)
    if regime == "init":  # the reference's own initial state: every opacity 0.05 (latent_gs_renderer.py:431)
        opac = inverse_sigmoid(0.05 * torch.ones(num_pts, 1))
    elif regime == "low":  # early training, right behind "init": opacities U(0.01, 0.1)
        opac = inverse_sigmoid(0.01 + 0.09 * torch.rand(num_pts, 1, generator=tg))
    elif regime == "trained":  # logits U(-2, 4)
        opac = torch.rand(num_pts, 1, generator=tg) * 6 - 2
    else:
        raise ValueError(f"unknown regime {regime!r} (init | low | trained)")
    f_dc = RGB2SH(torch.rand(num_pts, 1, 3, generator=tg))
    k = (g.max_sh_degree + 1) ** 2
    P = lambda t: nn.Parameter(t.to(dev).float().contiguous().requires_grad_(True))
    g._xyz, g._scaling, g._rotation, g._opacity = P(xyz), P(scales), P(rots), P(opac)
    g._features_dc, g._features_rest = P(f_dc), P(torch.zeros(num_pts, k - 1, 3))
    g._c_xyz = P(torch.tensor(ball_points(num_cpts, 0.5, rng), dtype=torch.float32))
    g._c_radius = P(scales.mean().cpu() * torch.ones(num_cpts, 1))
    g._r = torch.empty(0, device=dev)
    g.max_radii2D = torch.zeros(num_pts, device=dev)
    g.spatial_lr_scale = 1
    with torch.no_grad():
        # weights AND biases drawn on the host from the seeded generator (identical on CPU, GPU and every DP rank)
        for m in g._timenet.modules():
            if isinstance(m, nn.Linear):
                w = torch.empty(m.weight.shape)
                nn.init.xavier_uniform_(w, gain=1, generator=tg)
                bound = 1.0 / math.sqrt(m.in_features)
                b = (torch.rand(m.bias.shape, generator=tg) * 2 - 1) * bound
                m.weight.copy_(w)
                m.bias.copy_(b)
        g._timenet.pts_layers[-1].bias.zero_()
        g._timenet.rot_layers[-1].bias.copy_(torch.tensor([1.0, 0.0, 0.0, 0.0]))
        # the reference zero-inits the heads (no motion); use small random heads so deformation is non-trivial
        g._timenet.pts_layers[-1].weight.copy_(torch.randn(g._timenet.pts_layers[-1].weight.shape, generator=tg) * 1e-2)
        g._timenet.rot_layers[-1].weight.copy_(torch.randn(g._timenet.rot_layers[-1].weight.shape, generator=tg) * 1e-2)
        lat = torch.randn(num_latent, g.latent_code_dim, generator=tg).to(dev)
        if g.vae_latent:
            g._mu, g._log_var = nn.Parameter(lat), nn.Parameter(torch.full_like(lat, -4.0))
        else:
            g._latent_codes = nn.Parameter(lat)
    g.num_latent_code = num_latent
    return g


class SyntheticTargets:
    """Device-resident target images / masks, generated on first use from (motion, view, frame)."""

    POOL = 64  # distinct target images kept resident; (motion, view, frame) hashes into the pool

    def __init__(self, resolution, device, seed=0):
        self.res, self.device, self.seed = resolution, device, seed
        self._pool = {}
        yy, xx = torch.meshgrid(torch.arange(resolution), torch.arange(resolution), indexing="ij")
        c = (resolution - 1) / 2
        self._disc = (((xx - c) ** 2 + (yy - c) ** 2) <= (0.4 * resolution) ** 2).float()[None].to(device)
        for slot in range(self.POOL):  # generated once, on the host RNG (identical on every rank), uploaded once
            gen = torch.Generator().manual_seed(self.seed * 1_000_003 + slot)
            self._pool[slot] = (torch.rand(3, resolution, resolution, generator=gen).to(device), self._disc)

    def get(self, motion, view, frame):
        """The reference keeps every ground-truth frame in host RAM and uploads one per render
        (main_train_dimo.py:283-284); here targets stay in HBM (288 GB) and are never regenerated."""
        slot = (motion * 10_007 + view * 101 + frame) % self.POOL
        t = self._pool.get(slot)
        if t is None:
            gen = torch.Generator().manual_seed(self.seed * 1_000_003 + slot)
            img = torch.rand(3, self.res, self.res, generator=gen).to(self.device)
            t = self._pool[slot] = (img, self._disc)
        return t


class TeacherTargets:
    """Self-consistent targets: the renders of a hidden, seeded TEACHER model at every (motion, view, frame) of the
    schedule -- image = the teacher's clamped render over the white background, mask = its alpha (what the reference's
    data loader gets from the alpha matte of a frame, utils/load_utils.py) -- generated once, resident in HBM.
    Unlike `SyntheticTargets` (seeded noise, SURVEY.md 8d's headline workload) these can be LEARNED: a loss that falls
    and a PSNR that rises on them is evidence that the step trains, which single-step parity cannot give."""

    def __init__(self, teacher, num_motions, num_views, num_frames, resolution, radius=2, fovy=33.9, elevation=0):
        from .camera import CameraCache
        self.res, self.device = resolution, teacher.device
        self.shape = (num_motions, num_views, num_frames)
        cams = CameraCache(fovy_deg=fovy, device=teacher.device)
        az, times = default_azimuths(num_views), frame_times(num_frames)
        g = teacher.gaussians
        if g.neighbor_indices is None:
            from .knn_cuda import knn_points
            g.neighbor_dists, g.neighbor_indices = knn_points(g._c_xyz.detach(), g._xyz.detach(), 4)
        n = num_motions * num_views * num_frames
        self.images = torch.empty(n, 3, resolution, resolution, device=teacher.device)
        self.masks = torch.empty(n, 1, resolution, resolution, device=teacher.device)
        k = 0
        with torch.no_grad():
            for m in range(num_motions):
                for v in range(num_views):
                    cam = cams.get(elevation, az[v], radius, resolution, resolution)
                    outs = [teacher.render(cam, time=times[f], stage="s2", latent_index=m) for f in range(num_frames)]
                    for o in outs:
                        self.images[k].copy_(o["image"] + 0)
                        self.masks[k].copy_(o["alpha"] + 0)
                        k += 1
        teacher.flush()

    def get(self, motion, view, frame):
        M, V, F = self.shape
        k = (motion * V + view) * F + frame
        return self.images[k], self.masks[k]


def make_teacher(device, num_cpts=48, pts_per_cpt=40, num_motions=3, seed=1234, motion_gain=1.0, scale=0.025):
    """A hidden model of the family the schedule can reach: `num_cpts` control points in the unit ball, `pts_per_cpt`
    opaque coloured Gaussians around each (colour by position: the object has structure to match), a TimeNet whose
    heads (N(0, 1e-2) weights, scaled by `motion_gain`) move the control points; `scale`: the Gaussians' size
    ("dist2": from the nearest-neighbour distances, the reference's initialisation rule)."""
    from .renderer import Renderer
    rd = Renderer(sh_degree=0, white_background=True, num_latent_code=num_motions, add_normal=True, device=device)
    n = num_cpts * pts_per_cpt
    init_synthetic_model(rd, n, num_cpts, seed=seed, regime="trained", num_latent=num_motions)
    g = rd.gaussians
    tg = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        c = g._c_xyz.detach().cpu()
        # an object, not a cloud: control points on a blobby shell, the Gaussians close around them
        c = c / c.norm(dim=1, keepdim=True).clamp_min(1e-6) * (0.25 + 0.1 * torch.rand(num_cpts, 1, generator=tg))
        g._c_xyz.copy_(c.to(g.device))
        local = 0.05 * torch.randn(num_cpts, pts_per_cpt, 3, generator=tg)
        xyz = (c[:, None, :] + local).reshape(-1, 3)
        g._xyz.copy_(xyz.to(g.device))
        if scale == "dist2":  # the reference's initialisation rule (latent_gs_renderer.py:426-427): nearest-neighbour sized
            g._scaling.copy_(torch.log(torch.sqrt(torch.clamp_min(g._dist2(g._xyz.detach()), 1e-7)))[:, None].repeat(1, 3))
        else:
            g._scaling.fill_(math.log(scale))
        g._opacity.fill_(3.0)
        col = (xyz / 0.4 * 0.5 + 0.5).clamp(0.05, 0.95)
        g._features_dc.copy_(RGB2SH(col)[:, None, :].to(g.device))
        g._c_radius.fill_(math.log(0.08))
        g._timenet.pts_layers[-1].weight.mul_(motion_gain)
    return rd


def default_azimuths(num_views=9):
    return [360.0 / num_views * i for i in range(num_views)]  # main_train_dimo.py:80


def frame_times(num_frames=21):
    return [f / num_frames for f in range(num_frames)]  # main_train_dimo.py:104


def focal_pixels(resolution, fovy_deg=33.9):
    return resolution / (2 * math.tan(math.radians(fovy_deg) / 2))
