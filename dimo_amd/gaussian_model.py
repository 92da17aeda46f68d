"""Canonical-Gaussian parameter store of the render path.

Mirrors the parts of the reference's `GaussianModel` the training step touches
(renderer/latent_gs_renderer.py:248-515; VAE variant renderer/gaussian_gs_renderer.py:286-291,478-479):
parameter tensors and their names, activations, `create_from_pcd`, the 12 Adam groups with
eps=1e-15, and the learning-rate schedule.  One implementation serves both latent flavours
(`vae_latent=False`: `_latent_codes`; `True`: `_mu` / `_log_var`).

MI355X-first difference: after `training_setup` every trainable tensor is a VIEW into one flat
fp32 parameter buffer and every `.grad` a view into one flat gradient buffer
(`flat_params` / `flat_grads`), so data-parallel training all-reduces ONE contiguous bucket
over RCCL/xGMI with no packing copy, and zeroing gradients is a single memset.

Densify / prune / optimizer surgery live in densify.py, PLY and .pth I/O in ply_io.py;
both are mixed into `GaussianModel`.
"""
import math
from typing import NamedTuple

import numpy as np
import torch
from torch import nn

from .deform import TimeNet

C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation lr_init -> lr_final with an optional delayed warm-up
    (renderer/latent_gs_renderer.py:29-51)."""

    def helper(step):
        if lr_init == lr_final:
            return lr_init
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)

    return helper


from .densify import PER_GAUSSIAN, DensifyMixin  # noqa: E402
from .ply_io import PlyMixin  # noqa: E402


class BasicPointCloud(NamedTuple):
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray


class GaussianModel(DensifyMixin, PlyMixin):
    def __init__(self, sh_degree: int, num_latent_code: int = 1, latent_code_dim: int = 32, vae_latent: bool = False,
                 device=None, dist2_fn=None):
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.num_latent_code, self.latent_code_dim, self.vae_latent = num_latent_code, latent_code_dim, vae_latent
        e = lambda: torch.empty(0, device=self.device)
        self._xyz, self._features_dc, self._features_rest = e(), e(), e()
        self._scaling, self._rotation, self._opacity = e(), e(), e()
        self._c_xyz, self._c_radius, self._r = e(), e(), e()
        self.max_radii2D, self.xyz_gradient_accum, self.denom = e(), e(), e()
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        if vae_latent:
            self._mu = nn.Parameter(torch.zeros(num_latent_code, latent_code_dim, device=self.device))
            self._log_var = nn.Parameter(torch.zeros(num_latent_code, latent_code_dim, device=self.device))
        else:
            self._latent_codes = nn.Parameter(torch.randn(num_latent_code, latent_code_dim, device=self.device))
        self._timenet = TimeNet(latent_code_dim=latent_code_dim, device=self.device)
        self.neighbor_dists = self.neighbor_indices = None
        self.flat_params = self.flat_grads = None
        self._dist2_fn = dist2_fn  # defaults to the HIP distCUDA2 drop-in (no CPU fallback)
        self._flush_pending_renders = None  # set by a Renderer that queues renders (dimo_amd/batched_render.py)

    def flush_pending_renders(self):
        """Runs the renders a batching `Renderer` has queued and nobody has looked at yet.  Called before anything
        changes the parameters (optimizer step, densify / prune / sort): a queued render shows the model of the
        moment `render()` was called, exactly as an immediate render would."""
        fn = self._flush_pending_renders
        if fn is not None:
            fn()

    # ------------------------------------------------------------------ activations / accessors
    scaling_activation = staticmethod(torch.exp)
    scaling_inverse_activation = staticmethod(torch.log)
    opacity_activation = staticmethod(torch.sigmoid)
    inverse_opacity_activation = staticmethod(inverse_sigmoid)
    rotation_activation = staticmethod(torch.nn.functional.normalize)

    @property
    def get_scaling(self):
        n = self._xyz.shape[0]
        if len(self._r) == 0:
            return self.scaling_activation(self._scaling)
        if self._r.shape[0] != n:
            return self.scaling_activation(self._r.repeat(n, 3))
        if self._r.shape[1] == 1:
            return self.scaling_activation(self._r.repeat(1, 3))
        if self._r.shape == self._xyz.shape:
            return self.scaling_activation(self._r)
        raise ValueError("Shape of _r is not supported.")

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_c_xyz(self):
        return self._c_xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_c_radius(self, stage="s2"):
        if stage < "s2":
            return torch.exp(self._r.repeat(self._xyz.shape[0], 1))
        return torch.exp(self._c_radius)

    def latent_code(self, latent_index, generator=None):
        """Latent provider: plain code, or the VAE reparameterisation eps*std + mu
        (gaussian_gs_renderer.py:1088-1098)."""
        if not self.vae_latent:
            return self._latent_codes[latent_index]
        mu, log_var = self._mu[latent_index], self._log_var[latent_index]
        std = torch.exp(0.5 * log_var)
        eps = torch.randn(std.shape, dtype=std.dtype, device=std.device, generator=generator)
        return eps * std + mu

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------------ initialisation
    def _dist2(self, pts):
        if self._dist2_fn is not None:
            return self._dist2_fn(pts)
        from .simple_knn._C import distCUDA2
        return distCUDA2(pts)

    def create_from_pcd(self, pcd: BasicPointCloud, pcd2: BasicPointCloud, spatial_lr_scale: float = 1,
                        only_init_gaussians=False):
        """renderer/latent_gs_renderer.py:416-451."""
        dev = self.device
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points)).float().to(dev)
        color = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().to(dev))
        n, k = pts.shape[0], (self.max_sh_degree + 1) ** 2
        features = torch.zeros((n, 3, k), device=dev)
        features[:, :3, 0] = color
        dist2 = torch.clamp_min(self._dist2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=dev)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.05 * torch.ones((n, 1), device=dev))
        self._xyz = nn.Parameter(pts.requires_grad_(True))
        self._features_dc = nn.Parameter(features[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(features[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(opacities.requires_grad_(True))
        self.max_radii2D = torch.zeros(n, device=dev)
        if not only_init_gaussians:
            cpts = torch.tensor(np.asarray(pcd2.points)).float().to(dev)
            self._c_xyz = nn.Parameter(cpts.requires_grad_(True))
            # NB the reference takes the first scale column of the GAUSSIANS (pcd == pcd2 sizes in its callers)
            self._c_radius = nn.Parameter(scales[: cpts.shape[0], :1].clone().requires_grad_(True))
            self._r = nn.Parameter((scales.mean() * torch.ones((1, 1), device=dev)).requires_grad_(True))

    # ------------------------------------------------------------------ optimizer
    def param_groups(self, a):
        """The reference's Adam groups, in order (latent_gs_renderer.py:460-473; VAE: gaussian_gs_renderer.py:478-479)."""
        mlp, mlp_rot = self._timenet.get_mlp_parameters()
        latent = ([{"params": [self._mu], "lr": a.latent_code_lr_init, "name": "latent_code_mu"},
                   {"params": [self._log_var], "lr": a.latent_code_lr_init, "name": "latent_code_log_var"}]
                  if self.vae_latent else
                  [{"params": [self._latent_codes], "lr": a.latent_code_lr_init, "name": "latent_code"}])
        return [
            {"params": [self._xyz], "lr": a.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": a.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": a.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": a.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": a.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": a.rotation_lr, "name": "rotation"},
            *latent,
            {"params": list(mlp), "lr": a.deform_lr_init, "name": "deform"},
            {"params": list(mlp_rot), "lr": a.deform_lr_init, "name": "deform_rot"},
            {"params": [self._c_xyz], "lr": a.c_position_lr_init * self.spatial_lr_scale, "name": "c_xyz"},
            {"params": [self._c_radius], "lr": a.c_radius_lr, "name": "c_radius"},
            {"params": [self._r], "lr": a.r_lr, "name": "r"},
        ]

    FLAG_WORDS = 4  # floats in front of the gradients inside `flat_grads_ext` (16 bytes: the gradients stay aligned)

    def flatten_parameters(self, groups):
        """Re-home every trainable tensor (and its .grad) into one flat fp32 buffer each."""
        params = [p for g in groups for p in g["params"] if p.numel() > 0]
        # every tensor starts on a 16-byte boundary (the GEMM kernels of csrc/timenet.hip use 16-byte loads on
        # aligned operands); the <= 3 padding floats keep zero value and zero gradient, so Adam leaves them at 0
        pad4 = lambda n: (n + 3) & ~3
        total = sum(pad4(p.numel()) for p in params)
        flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        # 4 extra floats ride along with the gradient bucket (and its all-reduce): [0] = "some render of this
        # step overflowed its instance capacity on some rank" -> every replica skips the update together.  They LEAD the
        # bucket: the optimizer's early launch over the per-Gaussian head (Trainer: under the TimeNet backward, behind
        # the head's own all-reduce) must see the same all-reduced flag as its late launch over the tail
        self.flat_grads_ext = torch.zeros(self.FLAG_WORDS + total, dtype=torch.float32, device=self.device)
        grads = self.flat_grads_ext[self.FLAG_WORDS:]
        self.grad_flag = self.flat_grads_ext[0:1]
        o = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = flat[o:o + n].view(p.shape)
                p.grad = grads[o:o + n].view(p.shape)
                o += pad4(n)
        # the per-Gaussian groups lead the bucket: [0, flat_split) is final as soon as the last skinning backward has
        # accumulated, before the TimeNet backward runs (Trainer overlaps that part's all-reduce with it)
        self.flat_split = 0
        for g in groups:
            if g.get("name") not in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
                break
            self.flat_split += sum(pad4(p.numel()) for p in g["params"] if p.numel() > 0)
        self.flat_params, self.flat_grads = flat, grads
        return flat, grads

    # ------------------------------------------------------------------ optimizer surgery (densify / prune)
    def per_gaussian(self):
        """The six per-Gaussian parameter tensors by the reference's optimizer group names."""
        d = dict(xyz=self._xyz, f_dc=self._features_dc, f_rest=self._features_rest, opacity=self._opacity,
                 scaling=self._scaling, rotation=self._rotation)
        if self.r_is_per_point():  # a per-point `_r` is pruned / extended with the Gaussians (group "r")
            d["r"] = self._r
        return d

    def r_is_per_point(self):
        """The reference extends / prunes `_r` with the Gaussians only when it has one row per Gaussian
        (latent_gs_renderer.py:701,727,847,869); the shared (1, 1) radius of stage s1 is left alone."""
        return len(self._r) > 0 and self._r.dim() == 2 and self._r.shape[0] == self._xyz.shape[0] and self._r.shape[0] > 1

    def _moments(self, p):
        """(exp_avg, exp_avg_sq, step) of parameter p, shaped like p; None before the first step of torch Adam."""
        opt = self.optimizer
        if type(opt).__name__ == "FlatAdam":
            off = (p.data_ptr() - self.flat_params.data_ptr()) // 4
            n = p.numel()
            return (opt.exp_avg[off:off + n].view(p.shape), opt.exp_avg_sq[off:off + n].view(p.shape), opt.step_count)
        st = opt.state.get(p, None)
        if not st:
            return None
        return st["exp_avg"], st["exp_avg_sq"], st["step"]

    def rebuild(self, plan, zero_moments=(), ctrl_keep=None):
        """Applies a densify.py plan: new per-Gaussian parameters (values from the plan), Adam moments gathered
        from the plan's source rows (zero for fresh rows / the `zero_moments` groups), everything else carried over;
        ONE rebuild of the flat parameter / gradient / moment buckets.  `ctrl_keep` (bool mask over the control
        points, `prune_s1_end` only): the control points, their radii and their moments keep those rows."""
        assert self.optimizer is not None, "training_setup first"
        self.flush_pending_renders()
        old = self.per_gaussian()
        new_moments, carried = {}, {}
        for k, p in old.items():
            mo = self._moments(p)
            if mo is None:
                continue
            m, v = mo[0][plan.src].clone(), mo[1][plan.src].clone()
            m[plan.fresh], v[plan.fresh] = 0, 0
            if k in zero_moments:
                m.zero_(), v.zero_()
            new_moments[k] = (m, v, mo[2])
        per_ids = {id(p) for p in old.values()}
        for grp in self.optimizer.param_groups:
            for p in grp["params"]:
                if id(p) not in per_ids:
                    mo = self._moments(p)
                    if mo is not None:
                        carried[id(p)] = (mo[0].clone(), mo[1].clone(), mo[2])
        if ctrl_keep is not None:
            for name in ("_c_xyz", "_c_radius"):
                old_p = getattr(self, name)
                new_p = nn.Parameter(old_p.detach()[ctrl_keep].clone().contiguous().requires_grad_(True))
                mo = carried.pop(id(old_p), None)
                if mo is not None:
                    carried[id(new_p)] = (mo[0][ctrl_keep].clone(), mo[1][ctrl_keep].clone(), mo[2])
                setattr(self, name, new_p)
        lrs = {grp["name"]: grp["lr"] for grp in self.optimizer.param_groups}
        old_opt = self.optimizer
        P = lambda t: nn.Parameter(t.detach().clone().contiguous().requires_grad_(True))
        v = plan.values
        self._xyz, self._features_dc, self._features_rest = P(v["xyz"]), P(v["f_dc"]), P(v["f_rest"])
        self._opacity, self._scaling, self._rotation = P(v["opacity"]), P(v["scaling"]), P(v["rotation"])
        if "r" in v:
            self._r = P(v["r"])
        self._make_optimizer(self._training_args, self._fused)
        for grp in self.optimizer.param_groups:
            grp["lr"] = lrs.get(grp["name"], grp["lr"])
        if type(old_opt).__name__ == "FlatAdam":  # launch counter and the device's skipped-launch words carry over
            o = self.optimizer
            o.launches, o.skipped_host, o._skipped = old_opt.launches, old_opt.skipped_host, old_opt._skipped
        new = self.per_gaussian()
        with torch.no_grad():
            for grp in self.optimizer.param_groups:
                for p in grp["params"]:
                    key = next((k for k, q in new.items() if q is p), None)
                    src = new_moments.get(key) if key is not None else carried.get(id(p))
                    if src is None:
                        continue
                    if type(self.optimizer).__name__ == "FlatAdam":
                        m, s2, _ = self._moments(p)
                        m.copy_(src[0]), s2.copy_(src[1])
                    else:
                        step = src[2]
                        self.optimizer.state[p] = {"step": step.clone() if torch.is_tensor(step) else step,
                                                   "exp_avg": src[0], "exp_avg_sq": src[1]}
        self.neighbor_dists = self.neighbor_indices = None  # per-Gaussian KNN results are stale

    def _make_optimizer(self, training_args, fused):
        groups = self.param_groups(training_args)
        groups = [g for g in groups if all(p.numel() > 0 for p in g["params"])]
        self.flatten_parameters(groups)
        if fused == "flat":  # one HIP launch over the flat bucket (dimo_amd/csrc/adam.hip)
            from .flat_adam import FlatAdam
            self.optimizer = FlatAdam(groups, self.flat_params, self.flat_grads, eps=1e-15)
            self.optimizer.pre_step = self.flush_pending_renders
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, **({"fused": True} if fused else {}))
            self.optimizer.register_step_pre_hook(lambda *a, **k: self.flush_pending_renders())

    def training_setup(self, training_args, fused=None):
        self.percent_dense = training_args.percent_dense
        n = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        if fused is None:
            fused = "flat" if self.device.type == "cuda" else False
        self._training_args, self._fused = training_args, fused
        if self.max_radii2D.shape[0] != n:
            self.max_radii2D = torch.zeros(n, device=self.device)
        self._make_optimizer(training_args, fused)
        self.lr_setup(training_args)

    def zero_grad(self):
        """One memset over the flat gradient bucket (keeps the .grad views alive)."""
        self.flat_grads.zero_()

    def lr_setup(self, a):
        s = self.spatial_lr_scale
        self.xyz_scheduler_args = get_expon_lr_func(a.position_lr_init * s, a.position_lr_final * s,
                                                    lr_delay_mult=a.position_lr_delay_mult,
                                                    max_steps=a.position_lr_max_steps)
        self.c_xyz_scheduler_args = get_expon_lr_func(a.c_position_lr_init * s, a.c_position_lr_final * s,
                                                      lr_delay_mult=a.c_position_lr_delay_mult,
                                                      max_steps=a.position_lr_max_steps)
        self.latent_code_scheduler_args = get_expon_lr_func(a.latent_code_lr_init, a.latent_code_lr_final,
                                                            lr_delay_mult=a.position_lr_delay_mult,
                                                            max_steps=a.position_lr_max_steps)
        self.deform_scheduler_args = get_expon_lr_func(a.deform_lr_init * s, a.deform_lr_final * s,
                                                       lr_delay_mult=a.position_lr_delay_mult,
                                                       max_steps=a.position_lr_max_steps)
        self.deform_rot_scheduler_args = self.deform_scheduler_args

    def update_learning_rate(self, iteration, stage):
        """renderer/latent_gs_renderer.py:497-515."""
        for g in self.optimizer.param_groups:
            name = g["name"]
            if name == "xyz":
                g["lr"] = self.xyz_scheduler_args(iteration)
            if stage >= "s2":
                if name == "c_xyz":
                    g["lr"] = self.c_xyz_scheduler_args(iteration)
                elif name in ("latent_code", "latent_code_mu", "latent_code_log_var"):
                    g["lr"] = self.latent_code_scheduler_args(iteration)
                elif name in ("deform", "deform_rot"):
                    g["lr"] = self.deform_scheduler_args(iteration)
