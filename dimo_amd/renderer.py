"""`Renderer` -- the call surface the trainer depends on (drop-in for
renderer/latent_gs_renderer.py:973-1293 and, with `vae_latent=True`, renderer/gaussian_gs_renderer.py).

`render()` keeps the reference's signature, stage semantics ("s1": TimeNet moves the Gaussians
directly; "s2": TimeNet moves the control points and the Gaussians follow by LBS) and return dict
{image, depth, normal, alpha, viewspace_points, visibility_filter, radii, pts_t, cpts_t}.

Differences that do not change results:
  * rasterizer = dimo_amd HIP kernels (both flavours; `add_normal=False` returns normal=None instead of
    the reference's NameError, latent_gs_renderer.py:1286);
  * settings tuples / tan(fov) come from a per-camera cache instead of being rebuilt per call;
  * on the GPU the instance buffers are sized by a `CapacityPolicy` (no host synchronisation per render;
    `capacity=False` restores the exact, synchronising sizing of the CUDA original);
  * on the GPU, in the configuration DIMO trains in (stage s2, degree-0 colour, <= 1800 control points), the renders
    of a step are BATCHED behind this call surface: `render()` returns lazy stand-ins for image / depth / normal /
    alpha / radii and the pending renders run as one launch chain the first time one of them is used
    (dimo_amd/batched_render.py; `batch_renders=False` switches it off).
"""
import math

import numpy as np
import torch

from .deform import fused_skinning, fused_skinning_available, lbs_deform
from .gaussian_model import BasicPointCloud, GaussianModel, SH2RGB


def _ball_points(n, radius, rng):
    """Uniform-in-ball sampling exactly as latent_gs_renderer.py:999-1007 (draw order: phi, cos(theta), mu)."""
    phis = rng.random((n,)) * 2 * np.pi
    costheta = rng.random((n,)) * 2 - 1
    thetas = np.arccos(costheta)
    mu = rng.random((n,))
    r = radius * np.cbrt(mu)
    return np.stack((r * np.sin(thetas) * np.cos(phis), r * np.sin(thetas) * np.sin(phis), r * np.cos(thetas)), axis=1)


class Renderer:
    def __init__(self, sh_degree=3, white_background=True, radius=1, delta_t=1 / 32, num_latent_code=1,
                 latent_code_dim=32, add_normal=False, vae_latent=False, device=None, rasterizer_factory=None,
                 capacity=None, dist2_fn=None, batch_renders=True):
        self.sh_degree = sh_degree
        self.white_background = white_background
        self.radius = radius
        self.gaussians = GaussianModel(sh_degree=sh_degree, num_latent_code=num_latent_code,
                                       latent_code_dim=latent_code_dim, vae_latent=vae_latent, device=device,
                                       dist2_fn=dist2_fn)
        self.device = self.gaussians.device
        self.bg_color = torch.tensor([1, 1, 1] if white_background else [0, 0, 0], dtype=torch.float32,
                                     device=self.device)
        self.delta_t = delta_t
        self.add_normal = add_normal
        # capacity: a CapacityPolicy, None (GPU: a default policy is created at the first render) or False (exact
        # sizing from a read-back of the instance count: one stream sync per render, like the CUDA original)
        self.capacity = capacity
        self.batch_renders = batch_renders
        self._batcher = None
        # tests inject a CPU rasterizer here; the product default is the HIP one (no fallback)
        self._rasterizer_factory = rasterizer_factory
        self._np_rng = np.random  # the reference draws from numpy's global RNG

    # ------------------------------------------------------------------ initialisation
    def initialize(self, input=None, num_pts=5000, num_cpts=512, radius=0.5, radius2=0.5, only_init_gaussians=False):
        if input is None:
            rng = self._np_rng
            xyz = _ball_points(num_pts, radius, rng)
            shs = rng.random((num_pts, 3)) / 255.0
            pcd = BasicPointCloud(points=xyz, colors=SH2RGB(shs), normals=np.zeros((num_pts, 3)))
            xyz2 = _ball_points(num_cpts, radius2, rng)
            shs2 = rng.random((num_cpts, 3)) / 255.0
            pcd2 = BasicPointCloud(points=xyz2, colors=SH2RGB(shs2), normals=np.zeros((num_cpts, 3)))
            self.gaussians.create_from_pcd(pcd, pcd2, 1, only_init_gaussians=only_init_gaussians)
        elif isinstance(input, BasicPointCloud):
            self.gaussians.create_from_pcd(input, input, 1)
        else:
            raise ValueError("Unsupported initialization type!!!")

    def initialize_ag(self, c_xyz, c_radius, num_cpts=512, num_pts_per_cpt=200, init_ratio=1):
        """'Adaptive Gaussian' re-init: num_pts_per_cpt Gaussians around every control point
        (latent_gs_renderer.py:1038-1058)."""
        rng = self._np_rng
        local = _ball_points(num_pts_per_cpt, c_radius.mean().item() * init_ratio, rng)
        xyz = torch.tensor(local)[None].repeat(num_cpts, 1, 1).flatten(0, 1)
        centers = c_xyz.detach().cpu()[:, None].repeat(1, num_pts_per_cpt, 1).flatten(0, 1)
        xyz = (xyz + centers).numpy()
        shs = rng.random((num_pts_per_cpt * num_cpts, 3)) / 255.0
        pcd = BasicPointCloud(points=xyz, colors=SH2RGB(shs), normals=np.zeros((num_pts_per_cpt * num_cpts, 3)))
        self.gaussians.create_from_pcd(pcd, pcd, 1, only_init_gaussians=True)

    def reparameterize(self, mu, log_var):
        std = torch.exp(0.5 * log_var)
        return torch.randn_like(std) * std + mu

    # ------------------------------------------------------------------ rasterizer plumbing
    def capacity_policy(self):
        """The CapacityPolicy of this renderer (created on first use unless `capacity=False`); None = exact sizing."""
        if self.capacity is None and self.device.type == "cuda" and self._rasterizer_factory is None:
            from .rasterizer import CapacityPolicy
            self.capacity = CapacityPolicy(initial=max(1 << 20, 40 * max(1, self.gaussians._xyz.shape[0])))
        return self.capacity or None

    def _make_rasterizer(self, settings):
        if self._rasterizer_factory is not None:
            return self._rasterizer_factory(settings, self.add_normal)
        from . import rasterizer as rz
        cls = rz.GaussianRasterizerNormal if self.add_normal else rz.GaussianRasterizer
        return cls(raster_settings=settings, capacity=self.capacity_policy())

    def _guard_capacity(self):
        """Nobody consumes the policy's instance counts (a trainer does, once per step): look at them every 64 renders
        so that an overflow of the instance capacity cannot pass unnoticed (one stream sync per 64 renders)."""
        cap = self.capacity
        if cap and sum(t.shape[0] for t in cap._pending) >= 64 and not cap.check():
            raise RuntimeError("a render since the last check overflowed the instance capacity and dropped tile "
                               f"instances; the capacity is now {cap.capacity}: repeat the step")

    def flush(self):
        """Runs the renders `render()` has queued (see dimo_amd/batched_render.py).  Never needed for correctness: the
        first use of any queued output does it."""
        if self._batcher is not None:
            self._batcher.flush()

    def _batchable(self, stage, override_color, bg_color, xyz_detach, deform):
        g = self.gaussians
        if not (self.batch_renders and self.device.type == "cuda" and self._rasterizer_factory is None):
            return False
        if stage < "s2" or len(g._r) != 0 or override_color is not None or bg_color is not None or xyz_detach:
            return False
        if g._features_rest.numel() != 0 or g.neighbor_indices is None or self.capacity is False:
            return False
        from .deform import fused_skinning_available
        if not fused_skinning_available(g._xyz, g._c_xyz, g.neighbor_indices) or g._c_xyz.shape[0] > 1800:
            return False
        if deform is None:
            from .batched_render import timenet_fusable
            if g.vae_latent or not timenet_fusable(g._timenet, g._c_xyz):
                return False
        else:
            dx, dq = deform
            if not (dx.is_cuda and dx.dtype == torch.float32 and dq.dtype == torch.float32):
                return False
        return True

    def _render_batched(self, cam, scaling_modifier, local_frame, deform, time, latent_index):
        """Queues the render (dimo_amd/batched_render.py); None if no render slot is free."""
        from .batched_render import LazyTensor, RenderBatcher
        from .batched_render import _meta as _meta_of
        g = self.gaussians
        if self._batcher is None:
            self._batcher = RenderBatcher(self)
        b = self._batcher
        tanfovx = getattr(cam, "tanfovx", None) or math.tan(cam.FoVx * 0.5)
        tanfovy = getattr(cam, "tanfovy", None) or math.tan(cam.FoVy * 0.5)
        key = (int(cam.image_height), int(cam.image_width), float(scaling_modifier), bool(local_frame))
        if deform is not None:
            deform = (deform[0].contiguous(), deform[1].contiguous())
        handle = b.add(cam, tanfovx, tanfovy, key, deform, time, latent_index)
        if handle is None:
            return None
        pend, i = handle
        sink = pend["reqs"][i].sink
        g._flush_pending_renders = self._flush_ref  # the model's mutators and its optimizer run what is queued first
        dev, N, M = self.device, g._xyz.shape[0], g._c_xyz.shape[0]
        H, W = key[0], key[1]
        f32, meta = torch.float32, _meta_of

        def lazy(name, shape, dtype=f32):
            return LazyTensor(lambda: b.output(pend, i, name), meta(shape, dtype), dev, (pend, i, name, "plain"),
                              unit_range=name == "image")  # (the returned image is clamped: latent_gs_renderer.py:1279)

        radii = lazy("radii", (N,), torch.int32)
        slot = pend["reqs"][i].deform
        # (no deferred indexing on cpts_t / pts_t: the reference hands cpts_t[None] to a third-party autograd
        # extension -- chamferdist, main_train_dimo.py:297-299 -- which needs a real tensor)
        cpts_t = LazyTensor(lambda: g._c_xyz + b.deform_of(pend, slot)[0], meta((M, 3), f32), dev, defer=False)

        def pts_t():  # the skinned Gaussians live in a render slot (overwritten by the batch's backward)
            return b.ex.slots[b.pts_slot(pend, i)]["pts"].clone()

        return {
            "image": lazy("image", (3, H, W)), "depth": lazy("depth", (1, H, W)),
            "normal": lazy("normal", (3, H, W)) if self.add_normal else None,
            "alpha": lazy("alpha", (1, H, W)), "viewspace_points": sink,
            "visibility_filter": LazyTensor(lambda: radii.materialize() > 0, meta((N,), torch.bool), dev, defer=False),
            "radii": radii, "pts_t": LazyTensor(pts_t, meta((N, 3), f32), dev, defer=False), "cpts_t": cpts_t,
        }

    def _flush_ref(self):
        if self._batcher is not None:
            self._batcher.flush()

    def _settings(self, cam, scaling_modifier, bg_color):
        from .rasterizer import GaussianRasterizationSettings
        tanfovx = getattr(cam, "tanfovx", None) or math.tan(cam.FoVx * 0.5)
        tanfovy = getattr(cam, "tanfovy", None) or math.tan(cam.FoVy * 0.5)
        return GaussianRasterizationSettings(
            image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=tanfovx, tanfovy=tanfovy,
            bg=self.bg_color if bg_color is None else bg_color, scale_modifier=scaling_modifier,
            viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
            sh_degree=self.gaussians.active_sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)

    # ------------------------------------------------------------------ the hot path
    def render(self, viewpoint_camera, scaling_modifier=1.0, bg_color=None, override_color=None,
               compute_cov3D_python=False, convert_SHs_python=False, time=0.0, stage="s1", rot_as_res=True,
               xyz_detach=False, local_frame=True, direct_deform=False, vertices_deform=None, latent_index=0,
               deform=None):
        """`deform=(delta_xyz[M,3], delta_quat[M,4])` (extension): TimeNet outputs computed elsewhere, e.g. by
        `Trainer` for all renders of a step in one batched MLP call; `time`/`latent_index` are then unused."""
        g = self.gaussians
        if compute_cov3D_python or convert_SHs_python:
            raise NotImplementedError("python-side covariance / SH conversion is not on DIMO's training path")
        self._guard_capacity()
        if self._batchable(stage, override_color, bg_color, xyz_detach, deform):
            out = self._render_batched(viewpoint_camera, scaling_modifier, local_frame, deform, time, latent_index)
            if out is not None:
                return out
        elif self._batcher is not None:
            self._batcher.flush()  # (keeps the order of renders that share state with an unbatched one)
        # gradient sink for the screen-space means (densification statistics read .grad)
        screenspace_points = torch.zeros_like(g.get_xyz, requires_grad=True)
        settings = self._settings(viewpoint_camera, scaling_modifier, bg_color)
        rasterizer = self._make_rasterizer(settings)

        means3D = g.get_xyz
        rotations = g._rotation
        use_fused = (stage >= "s2" and len(g._r) == 0
                     and fused_skinning_available(means3D, g.get_c_xyz, g.neighbor_indices))
        if not use_fused:  # the fused kernel applies exp / sigmoid itself
            opacity = g.get_opacity
            scales = g.get_scaling
        latent_code = g.latent_code(latent_index) if deform is None else None

        fused = False
        if stage >= "s2":
            c_means3D = g.get_c_xyz
            if deform is not None:  # batched TimeNet output handed in by the trainer
                means3D_deform, rots_deform = deform
            else:
                means3D_deform, rots_deform = g._timenet(c_means3D, time, latent_code)
            cpts_t = c_means3D + means3D_deform
            if use_fused:
                # one HIP kernel: skinning + quat product + normalize + exp/sigmoid activations
                means3D, rotations, scales, opacity = fused_skinning(
                    g._xyz, g._rotation, g._scaling, g._opacity, c_means3D, g._c_radius, means3D_deform, rots_deform,
                    g.neighbor_dists, g.neighbor_indices, local_frame)
                fused = True
            else:
                means3D, rotations = lbs_deform(means3D, rotations, c_means3D, g.get_c_radius(stage),
                                                means3D_deform, rots_deform, g.neighbor_dists, g.neighbor_indices,
                                                local_frame)
        elif stage == "s1":
            means3D_deform, _ = g._timenet(means3D, time, latent_code)
            cpts_t = means3D + means3D_deform
            means3D = means3D + means3D_deform
        else:
            raise ValueError("Nonexistent stage!!!")
        if xyz_detach:
            means3D = means3D.detach()
        if not fused:
            rotations = g.rotation_activation(rotations)

        shs = colors_precomp = None
        if override_color is None:
            shs = g.get_features
        else:
            colors_precomp = override_color

        if self.add_normal:
            image, depth, normal, alpha, radii, _extra = rasterizer(
                means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                opacities=opacity, scales=scales, rotations=rotations, cov3Ds_precomp=None, extra_attrs=None)
        else:
            image, radii, depth, alpha = rasterizer(
                means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=None)
            normal = None
        image = image.clamp(0, 1)
        return {
            "image": image, "depth": depth, "normal": normal, "alpha": alpha,
            "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
            "pts_t": means3D, "cpts_t": cpts_t,
        }
