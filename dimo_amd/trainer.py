"""One training step of DIMO's motion-latent stage, sharded over (motion, view, frame).

Mirrors `GUI.train_step` (main_train_dimo.py:221-451): learning-rate update, KNN of every Gaussian among
the control points (stage s2), sampling of motions x views x frames, one render per triple, per-motion image
losses (weighted MSE, SSIM, mask MSE, edge-aware depth smoothness, bilateral normal smoothness), ONE backward,
Adam (eps 1e-15), gradient reset.  LPIPS (needs downloaded VGG weights), ARAP and the chamfer "GA" term are
excluded and reported as excluded by bench.py.

Data parallelism (new -- the reference is single-GPU): every rank holds a full replica, draws the SAME
sample (shared seeds), renders its contiguous slice of the triples, and the flat gradient bucket
(GaussianModel.flat_grads_ext) is summed with one RCCL all-reduce before the identical Adam update.  The
reference's batch loss is a SUM over renders/motions, so the reduction is SUM, not mean; per-motion
mean-type terms are written per image with weight 1/(images per motion), which makes the loss
independent of how the triples are split.

Two interchangeable forward+backward pipelines (tested equal):
  * autograd : `Renderer.render` per triple + torch losses + one `loss.backward()` -- the reference's shape;
  * direct   : the native step executor (dimo_amd/executor.py): two host calls run every render's HIP kernel
               chain on private streams, losses are two fused kernels per motion, only TimeNet uses autograd.
"""
import random
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .camera import CameraCache
_LOSS_WORDS = 512  # include/dimo_hip.h: DIMO_LOSS_WORDS (image_loss.py imports the HIP library; keep this module CPU-importable)
from .losses import compute_bilateral_normal_smoothness_loss, compute_edge_aware_smoothness_loss
from .synth import SyntheticTargets, default_azimuths, frame_times


@dataclass
class TrainConfig:
    """The hot-path-relevant keys of configs/train_config.yaml (same names, same defaults)."""
    # scene
    num_pts: int = 100000
    num_cpts: int = 512
    sh_degree: int = 0
    latent_code_dim: int = 32
    num_motions: int = 51
    num_frames: int = 21
    num_views: int = 9
    vae_latent: bool = False
    # sampling (the reference draws batch_size frames, batch_size views, min(2*batch_size, motions) motions)
    motions_per_step: int = 4
    views_per_step: int = 2
    frames_per_step: int = 2
    resolution: int = 512
    # main_train_dimo.py:261: renders (and the targets, resampled bilinearly) at 128^2 while step < 300, 256^2 while
    # step < 450, full size afterwards -- never above `resolution`
    progressive_resolution: bool = True
    # camera
    radius: float = 2
    fovy: float = 33.9
    elevation: float = 0
    # losses
    lambda_mse: float = 5000.0
    lambda_ssim: float = 500.0
    lambda_mask: float = 500.0
    add_depth: bool = True
    depth_reg_start_iter: int = 200   # the term is on for step > this (main_train_dimo.py:362)
    lambda_smooth: float = 100.0
    add_normal: bool = True
    normal_reg_start_iter: int = 200  # main_train_dimo.py:368
    lambda_bilateral: float = 0.05
    lambda_kl: float = 0.05
    # optimizer
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    percent_dense: float = 0.01
    position_lr_init: float = 0.01
    position_lr_final: float = 0.0002
    position_lr_delay_mult: float = 0.02
    position_lr_max_steps: int = 1000
    feature_lr: float = 0.01
    rotation_lr: float = 0.005
    c_radius_lr: float = 0.005
    latent_code_lr_init: float = 0.005
    latent_code_lr_final: float = 0.0002
    deform_lr_init: float = 0.0002
    deform_lr_final: float = 0.000002
    c_position_lr_init: float = 0.000002
    c_position_lr_final: float = 0.000002
    c_position_lr_delay_mult: float = 0.02
    r_lr: float = 0.01
    # densification / pruning schedule (configs/train_config.yaml:79-88; main_train_dimo.py:426-443)
    density_start_iter: int = 100
    density_end_iter: int = 1000
    density_end_iter_s2: int = 5000
    densification_interval: int = 100
    densification_interval_s2: int = 1000
    opacity_reset_interval: int = 200000
    densify_grad_threshold: float = 0.01
    densify_opacity_threshold_s1: float = 0.01
    densify_opacity_threshold_s2: float = 0.01
    init_type: str = "ag"
    init_ratio: float = 1.0       # configs/train_config.yaml:118
    num_pts_per_cpt: int = 200    # Gaussians spawned around every control point at the start of stage s2 (init_type "ag")
    FPS_iter: int = 1000  # stage s1: farthest-point down-sampling to num_cpts every FPS_iter steps
    # exactly the reference's schedule: render at 128 / 256 / 512 WHATEVER the targets' size and resample the targets
    # (main_train_dimo.py:261-313).  Default off: a run configured with smaller targets (the tests, the CPU oracle)
    # keeps rendering at most at the targets' size -- identical for the reference's own ref_size = 512 configuration
    progressive_upsample: bool = False
    # MI355X layout (no reference counterpart): keep the canonical Gaussians in Morton order (densify.py)
    spatial_sort: bool = True
    # regularisers (configs/train_config.yaml:57-65).  Off by default: BASELINE.json's metric is quoted without them
    use_lpips: bool = False   # main_train_dimo.py:339-341 (needs the metric's weights: dimo_amd/lpips_vgg.py)
    lambda_lpips: float = 1000.0  # configs/train_config.yaml:44
    use_arap: bool = False
    arap_start_iter_s1: int = 1000
    arap_end_iter_s2: int = 2000
    lambda_arap: float = 10.0
    add_ga: bool = False
    ga_chamfer: bool = True
    lambda_ga1: float = 10.0
    lambda_ga2: float = 10000.0
    seed: int = 0
    stage: str = "s2"


def enumerate_triples(motions, views, frames):
    """The reference's loop order: motion outermost, then view, then frame (main_train_dimo.py:276-281)."""
    return [(m, v, f) for m in motions for v in views for f in frames]


def shard(items, rank, world):
    """Contiguous slice of `items` for `rank` (sizes differ by at most one)."""
    n = len(items)
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    return items[lo:hi]


class _LazyLoss:
    """Loss of a direct-pipeline step, kept as the device scalars the kernels produced."""

    def __init__(self, loss_accum, ssim_terms, extra=None):
        self.loss_accum, self.ssim_terms, self.extra = loss_accum, ssim_terms, extra

    def value(self):
        loss = self.loss_accum.sum()
        if self.extra is not None:  # KL / ARAP / GA scalars (main stream; kept apart from the kernels' atomic adds)
            loss = loss + self.extra
        for ssum, lam, numel in self.ssim_terms:
            loss = loss + lam * (1 - ssum[0] / numel)
        return loss


class Trainer:
    def __init__(self, cfg: TrainConfig, renderer, rank=0, world_size=1, process_group=None,
                 ssim_fn: Optional[Callable] = None, knn_fn: Optional[Callable] = None, targets=None, direct=None,
                 fps_fn: Optional[Callable] = None):
        self.cfg, self.renderer = cfg, renderer
        self.rank, self.world, self.pg = rank, world_size, process_group
        self.device = renderer.device
        self.step = 0
        self.stage = cfg.stage
        if ssim_fn is None:
            from .fused_ssim import ssim as ssim_fn  # HIP, GPU only
        if knn_fn is None:
            from .knn_cuda import knn_points as knn_fn  # HIP, GPU only
        self.ssim, self.knn = ssim_fn, knn_fn
        self._knn_seeded = knn_fn.__module__ == "dimo_amd.knn_cuda"  # (a test's CPU stand-in takes no seeds)
        self._fps_fn = fps_fn  # stage s1 down-sampling; default: the HIP kernel behind regularizers.sample_farthest_points
        self.cpts_s1 = None    # [motions, frames, M, 3]: control-point trajectories cached at stage-s2 step 0 (GA term)
        self._resampled = {}   # targets at the reduced render sizes of the first 450 steps
        self.cams = CameraCache(fovy_deg=cfg.fovy, device=self.device)
        self.azimuths = default_azimuths(cfg.num_views)
        self.source_time = frame_times(cfg.num_frames)
        self.targets = targets or SyntheticTargets(cfg.resolution, self.device, seed=cfg.seed)
        # rank-identical sampling: all three RNGs the reference uses are seeded (it leaves `random` unseeded)
        self._py_rng = random.Random(cfg.seed)
        self._np_rng = np.random.default_rng(cfg.seed)
        if cfg.spatial_sort:  # Morton order of the canonical Gaussians (densify.py: sort_spatially), kept by densify_schedule
            renderer.gaussians.sort_spatially()
        renderer.gaussians.training_setup(cfg)
        if cfg.stage == "s1":  # prepare_train_s1 (main_train_dimo.py:464-469): the control points do not train in s1
            for grp in renderer.gaussians.optimizer.param_groups:
                if grp["name"] in ("c_radius", "c_xyz"):
                    grp["lr"] = 0.0
        self._last_loss = None
        self._consts = {}
        self._ext_streams = {}
        self._deform_batch = None
        self._exec = None
        import os
        # TimeNet as two native calls (dimo_amd/csrc/timenet.hip) in the direct pipeline; False: PyTorch autograd MLP
        self.fused_timenet = True
        self._fused_tn = None
        self.marks = None  # set to [] to collect (name, torch.cuda.Event) phase marks on the main stream
        self.skipped_steps = 0
        self.time_allreduce = False  # bench.py: event pairs around the step's collective (exposed time)
        self.allreduce_events = []
        self._flat_adam = type(self.optimizer).__name__ == "FlatAdam"
        # per-motion backward (default since round 4): every motion's chain -- forward, losses, rasterizer backward --
        # runs in order on ONE private stream, so a motion's backward overlaps the other motion's losses: 1.6-4 % more
        # frames/s than the joint launch.  DIMO_JOINT_BWD=1: ONE blend / projection
        # backward launch over all the step's renders on this stream (the kernel then runs alone on the device: bench.py
        # switches to it for the pass its roofline clock is taken in)
        self._joint_bwd = os.environ.get("DIMO_JOINT_BWD", "0") == "1"
        # The three below are attributes, not switches: their "False" paths are what the stage-s1 / single-stream /
        # several-rank schedules run anyway (the tests flip them to compare schedules).
        # per-motion backward: the motion's SKINNING backward too in order on its stream (control-point sums staged,
        # one fold at the end) instead of one skinning backward per motion on this stream
        self._skin_in_order = True
        self._split_adam = True  # the fold + Adam's per-Gaussian head on a private stream under the TimeNet backward
        # SSIM and the other image terms of a motion in ONE tile pass (csrc/ssim.hip: dimo_ssim_image_loss); "0": the
        # SSIM kernel followed by the loss kernel (also what runs with LPIPS on, or above 64 images per motion)
        self._fused_loss = os.environ.get("DIMO_FUSED_LOSS", "1") == "1"
        self._side_knn = True  # KNN on a private stream next to the TimeNet forward
        self._direct_wanted = direct
        self._decide_direct()

    def _decide_direct(self):
        """Direct HIP pipeline: GPU, degree-0 colour (DIMO's configuration), product rasterizer; stage s2 (skinning by
        <= 1800 control points, `_r` retired) or stage s1 (the TimeNet moves the Gaussians, shared (1, 1) radius `_r`)."""
        cfg, renderer, direct = self.cfg, self.renderer, self._direct_wanted
        g0 = renderer.gaussians
        stage_ok = (self.stage >= "s2" and len(g0._r) == 0 and g0._c_xyz.shape[0] <= 1800) or \
                   (self.stage == "s1" and tuple(g0._r.shape) == (1, 1))
        self.direct = (direct if direct is not None else True) and self.device.type == "cuda" \
            and stage_ok and cfg.sh_degree == 0 and renderer._rasterizer_factory is None
        if self.direct and not renderer.capacity:
            from .rasterizer import CapacityPolicy
            renderer.capacity = CapacityPolicy(initial=max(1 << 20, 40 * cfg.num_pts))

    @property
    def optimizer(self):
        """Always the model's CURRENT optimizer (densification / pruning rebuild it)."""
        return self.renderer.gaussians.optimizer

    @property
    def last_loss(self):
        """Loss of the last step (0-d tensor) or None."""
        loss = self._last_loss
        if isinstance(loss, _LazyLoss):
            loss = self._last_loss = loss.value()
        return loss.detach() if loss is not None else None

    # ------------------------------------------------------------------ the two-stage schedule
    def finish_stage_s1(self):
        """End of stage s1 (main_train_dimo.py:199-200): Gaussians and control points below opacity 0.01 go."""
        self.renderer.gaussians.prune_s1_end(min_opacity=0.01, extent=4, max_screen_size=1)

    def prepare_train_s2(self, iters_s2=None):
        """`GUI.prepare_train_s2` (main_train_dimo.py:471-500): the stage-s1 Gaussians BECOME the control points (their
        positions, the shared radius exp(_r) as every control radius), the canonical Gaussians are re-initialised --
        `init_type` "ag": `num_pts_per_cpt` around every control point, "normal": `num_pts` in the unit ball --, the
        optimizer starts afresh (TimeNet and latents keep their values), `_r` retires, and the position learning-rate
        schedule becomes 2e-4 -> 2e-6 over `iters_s2` steps (:497-500, applied by the `lr_setup` of :209)."""
        c, rd, g = self.cfg, self.renderer, self.renderer.gaussians
        g.flush_pending_renders()
        if g._c_xyz.shape[0] != g._xyz.shape[0]:
            raise ValueError("prepare_train_s2: stage s1 must end with one Gaussian per control point "
                             f"({g._xyz.shape[0]} Gaussians, {g._c_xyz.shape[0]} control points): run FPS / "
                             "prune_s1_end first")
        self.stage, self.step, c.stage = "s2", 0, "s2"
        with torch.no_grad():
            g._c_xyz.copy_(g._xyz)
            g._c_radius.copy_(g._r.detach().reshape(1, -1)[:, :1].expand_as(g._c_radius))
        g.optimizer = None  # (stage s1's: the model below is a new one)
        rd._np_rng = np.random.default_rng(c.seed + 7919)  # (rank-identical re-initialisation; the reference draws from numpy's global generator)
        if c.init_type == "normal":
            rd.initialize(num_pts=c.num_pts, only_init_gaussians=True)
        elif c.init_type == "ag":
            rd.initialize_ag(g._c_xyz, g.get_c_radius(stage="s2"), num_cpts=g._c_xyz.shape[0],
                             num_pts_per_cpt=c.num_pts_per_cpt, init_ratio=c.init_ratio)
        else:
            raise ValueError("Unsupported init type!!!")
        # (the reference retires `_r` right AFTER training_setup and pops its Adam group; retiring it first leaves the
        # same optimizer: the flat bucket has no slot for a parameter nobody trains)
        g._r = torch.empty(0, device=self.device)
        g.neighbor_dists = g.neighbor_indices = None
        if c.spatial_sort:
            g.sort_spatially()
        g.training_setup(c)
        g.active_sh_degree = g.max_sh_degree
        if iters_s2 is not None:
            c.position_lr_max_steps = int(iters_s2)
        c.position_lr_init, c.position_lr_final = 0.0002, 0.000002
        g.lr_setup(c)
        self.cpts_s1 = None
        self._resampled.clear()
        self._fused_tn = None
        self._last_stats = None
        if self._exec is not None:
            torch.cuda.synchronize()
            self._exec.destroy()
            self._exec = None
        self._flat_adam = type(self.optimizer).__name__ == "FlatAdam"
        self._decide_direct()

    def train_dynamic(self, iters_s1, iters_s2, on_step=None):
        """`GUI.train_dynamic` (main_train_dimo.py:170-218) without its file I/O: stage s1, the end-of-stage prune, the
        hand-over, stage s2.  `on_step(trainer)` is called after every step (logging / evaluation hooks)."""
        if self.stage == "s1":
            for _ in range(iters_s1):
                self.train_step()
                if on_step is not None:
                    on_step(self)
            if iters_s1 > 0:
                self.finish_stage_s1()
            self.prepare_train_s2(iters_s2)
        for _ in range(iters_s2):
            self.train_step()
            if on_step is not None:
                on_step(self)

    # ------------------------------------------------------------------ pieces of train_step
    def find_knn(self, k=4, stream=None):
        """`stream`: raw handle of the stream to search on (the direct pipeline runs the search on one of the
        executor's private streams, next to the TimeNet forward)."""
        g = self.renderer.gaussians
        kw = {} if stream is None else {"stream": stream}
        if stream is not None:
            # the search READS last step's neighbours (its seeds) on a stream the caching allocator knows nothing about:
            # they stay alive until the next search (by then this step's streams have been joined), instead of going
            # back to the pool -- and under a kernel of this stream -- the moment they are replaced below
            self._knn_keep = (g.neighbor_dists, g.neighbor_indices)
        if self._knn_seeded and g.neighbor_indices is not None:  # last step's neighbours prune this step's search
            d, i = self.knn(g._c_xyz.detach(), g._xyz.detach(), k, seed=g.neighbor_indices, **kw)
        else:
            d, i = self.knn(g._c_xyz.detach(), g._xyz.detach(), k, **kw)
        g.neighbor_dists, g.neighbor_indices = d, i

    def fps(self, num_pts):
        """`GUI.FPS` (main_train_dimo.py:511-515): stage-s1 down-sampling of the Gaussians to `num_pts` by farthest
        point sampling.  Quirk kept: the reference hands the INDEX tensor to `prune_points(mask)`, whose `~mask` on
        int64 is a bitwise not (-i-1), so the rows that survive are N-1-idx, in sampling order."""
        g = self.renderer.gaussians
        if g._xyz.shape[0] == 0:
            return
        if self._fps_fn is not None:
            idxs = self._fps_fn(g._xyz.detach(), num_pts)
        else:
            from .regularizers import sample_farthest_points
            idxs = sample_farthest_points(g._xyz.detach()[None], num_pts)[1][0]
        g.prune_points(idxs.to(torch.int64))

    def render_resolution(self):
        """Render size of the current step (main_train_dimo.py:261)."""
        c = self.cfg
        if not c.progressive_resolution:
            return c.resolution
        r = 128 if self.step < 300 else (256 if self.step < 450 else 512)
        return r if c.progressive_upsample else min(c.resolution, r)

    def target(self, m, v, f):
        """(image [3, r, r], mask [1, r, r]) of a triple at the step's render size (bilinear, align_corners=False:
        main_train_dimo.py:307-313)."""
        img, mask = self.targets.get(m, v, f)
        r = self.render_resolution()
        if img.shape[-1] != r or img.shape[-2] != r:
            key = (m, v, f, r)
            hit = self._resampled.get(key)
            if hit is None:
                F = torch.nn.functional
                hit = (F.interpolate(img[None], (r, r), mode="bilinear", align_corners=False)[0],
                       F.interpolate(mask[None], (r, r), mode="bilinear", align_corners=False)[0])
                if len(self._resampled) > 4096:
                    self._resampled.clear()
                self._resampled[key] = hit
            return hit
        return img, mask

    def sample(self) -> List[Tuple[int, int, int]]:
        c = self.cfg
        frames = self._py_rng.sample(range(c.num_frames), c.frames_per_step)
        views = self._py_rng.sample(range(c.num_views), c.views_per_step)
        motions = self._np_rng.choice(c.num_motions, min(c.motions_per_step, c.num_motions), replace=False)
        return enumerate_triples([int(m) for m in motions], views, frames)

    def render_triple(self, m, v, f, deform=None):
        c = self.cfg
        r = self.render_resolution()
        cam = self.cams.get(c.elevation, self.azimuths[v], c.radius, r, r)
        return self.renderer.render(cam, time=self.source_time[f], stage=self.stage, latent_index=m, deform=deform)

    def batched_deform(self, triples):
        """TimeNet for all DISTINCT (motion, frame) pairs of the step in one MLP call (stage s2).

        The reference evaluates the MLP once per render (M = 512 rows each: launch-bound, and views of the
        same (motion, frame) repeat identical work); the control points, times and latents of a step are all
        known up front, so one [pairs*M, 104] batch replaces 2b^3 small ones.  Returns {triple: (dxyz, dquat)}."""
        g = self.renderer.gaussians
        # VAE latents are re-sampled per render in the reference, so only the plain-latent flavour dedupes views
        key = (lambda m, v, f: (m, v, f)) if g.vae_latent else (lambda m, v, f: (m, f))
        pairs = list(dict.fromkeys(key(m, v, f) for (m, v, f) in triples))
        if not pairs:
            return {}
        M = g._c_xyz.shape[0]
        from .batched_render import timenet_apply, timenet_fusable
        if self.device.type == "cuda" and self.fused_timenet and timenet_fusable(g._timenet, g._c_xyz):
            # one fused forward for the step's distinct pairs, one autograd node (dimo_amd/batched_render.py)
            tl = [self.source_time[p[-1]] for p in pairs]
            if g.vae_latent:
                dxyz, dquat = timenet_apply(g._timenet, g._c_xyz, torch.stack([g.latent_code(p[0]) for p in pairs]), tl)
            else:
                dxyz, dquat = timenet_apply(g._timenet, g._c_xyz, g._latent_codes, tl, [p[0] for p in pairs])
        else:
            times = torch.tensor([self.source_time[p[-1]] for p in pairs], dtype=torch.float32, device=self.device)
            times = times[:, None, None].expand(-1, M, 1)
            lat = torch.stack([g.latent_code(p[0]) for p in pairs])[:, None, :].expand(-1, M, -1)
            dxyz, dquat = self._timenet_batched(g._c_xyz[None], times, lat)
        self._deform_batch = (dxyz, dquat, {(m, v, f): pairs.index(key(m, v, f)) for (m, v, f) in triples})
        out = {p: (dxyz[i], dquat[i]) for i, p in enumerate(pairs)}
        return {(m, v, f): out[key(m, v, f)] for (m, v, f) in triples}

    def _fused_deform(self, triples):
        """`batched_deform` on the HIP library: one dimo_timenet_forward for the step's distinct pairs."""
        from .fused_timenet import FusedTimeNet
        g = self.renderer.gaussians
        key = (lambda m, v, f: (m, v, f)) if g.vae_latent else (lambda m, v, f: (m, f))
        pairs = list(dict.fromkeys(key(m, v, f) for (m, v, f) in triples))
        times = [self.source_time[p[-1]] for p in pairs]
        if self._fused_tn is None:
            self._fused_tn = FusedTimeNet(g._timenet)
        if g.vae_latent:
            lat = torch.stack([g.latent_code(p[0]) for p in pairs]) if pairs else None
            table, rows = (lat.detach().contiguous() if pairs else g._mu), None
        else:
            lat, table, rows = None, g._latent_codes, [p[0] for p in pairs]
        pts = g._xyz if self.stage == "s1" else g._c_xyz  # s1: the MLP is evaluated on the Gaussians themselves
        dxyz, dquat = self._fused_tn.forward(pts.detach().contiguous(), times, table, rows)
        return dxyz, dquat, {(m, v, f): pairs.index(key(m, v, f)) for (m, v, f) in triples}, lat

    def _timenet_batched(self, pts, times, lat):
        """TimeNet on the whole step's batch (the autograd module; the direct pipeline uses dimo_timenet_*)."""
        return self.renderer.gaussians._timenet(pts, times, lat, t_apply=True)

    def motion_loss(self, outs, gts, masks, weights, n_img):
        """Loss of one motion's local images; mean-type terms carry the share len(outs)/n_img."""
        c = self.cfg
        from .batched_render import materialize  # (stand-ins of queued renders -> the batch's tensors, zero-copy)
        img = materialize(torch.stack([o["image"] for o in outs]))  # (the clamped render: values in [0, 1])
        share = len(outs) / n_img
        depth_on, normal_on = self._reg_on()
        alpha = materialize(torch.stack([o["alpha"] for o in outs]))
        if img.is_cuda and self.ssim.__module__ == "dimo_amd.fused_ssim" and not c.use_lpips:
            # every image term of the motion as ONE autograd node on the fused kernels (dimo_amd/fused_losses.py)
            from .fused_losses import motion_loss as fused_motion_loss
            from .image_loss import loss_weights
            B, _, H, W = img.shape
            depth = materialize(torch.stack([o["depth"] for o in outs])) if depth_on else None
            normal = materialize(torch.stack([o["normal"] for o in outs])) if normal_on else None
            return fused_motion_loss(img, depth, normal, alpha, list(gts), list(masks),
                                     [c.lambda_mse * w / (3 * H * W) for w in weights],
                                     loss_weights(c, B, n_img, H, W, depth_on, normal_on), c.lambda_ssim * share)
        gt = torch.stack(gts)
        w = torch.tensor(weights, dtype=img.dtype, device=img.device)
        per_img_mse = ((img - gt) ** 2).mean(dim=(1, 2, 3))
        loss = c.lambda_mse * (w * per_img_mse).sum()
        loss = loss + c.lambda_ssim * share * (1 - self.ssim(img, gt))
        loss = loss + c.lambda_mask * share * ((alpha - torch.stack(masks)) ** 2).mean()
        img_hwc = img.permute(0, 2, 3, 1)
        if depth_on:
            depth = materialize(torch.stack([o["depth"] for o in outs])).permute(0, 2, 3, 1)
            loss = loss + c.lambda_smooth * share * compute_edge_aware_smoothness_loss(depth, img_hwc, img.is_cuda)
        if normal_on:
            normal = materialize(torch.stack([o["normal"] for o in outs])).permute(0, 2, 3, 1)
            loss = loss + c.lambda_bilateral * share * compute_bilateral_normal_smoothness_loss(normal, img_hwc, img.is_cuda)
        if c.use_lpips:
            loss = loss + c.lambda_lpips * share * self.lpips_metric()(img, gt).mean()
        return loss

    def _reg_on(self):
        """(depth smoothness on, normal smoothness on) at the current step (main_train_dimo.py:362,368)."""
        c = self.cfg
        return (c.add_depth and self.step > c.depth_reg_start_iter,
                c.add_normal and self.renderer.add_normal and self.step > c.normal_reg_start_iter)

    # ------------------------------------------------------------------ geometry-anchor term (stage s2)
    @torch.no_grad()
    def cache_cpts_s1(self):
        """main_train_dimo.py:231-244: at stage-s2 step 0 the control-point trajectory of every (motion, frame) is
        frozen as the anchor of the geometry-anchor ("GA") term.  One batched TimeNet call instead of 51 x 21."""
        g, c = self.renderer.gaussians, self.cfg
        M = g._c_xyz.shape[0]
        out = torch.empty(c.num_motions, c.num_frames, M, 3, dtype=torch.float32, device=self.device)
        times = torch.tensor(self.source_time, dtype=torch.float32, device=self.device)[:, None, None].expand(-1, M, 1)
        for m in range(c.num_motions):
            lat = g.latent_code(m)[None, None, :].expand(c.num_frames, M, -1)
            dxyz, _ = g._timenet(g._c_xyz[None], times, lat, t_apply=True)
            out[m] = g._c_xyz[None] + dxyz
        self.cpts_s1 = out

    def _ga_active(self):
        return self.cfg.add_ga and self.stage == "s2" and self.cpts_s1 is not None

    def ga_loss(self, cpts, m, f):
        """GA term of ONE render (the reference adds it per (motion, view, frame) render, main_train_dimo.py:295-303)."""
        from .regularizers import geometry_anchor_loss
        c = self.cfg
        return geometry_anchor_loss(cpts, self.cpts_s1[m, f], c.ga_chamfer, c.lambda_ga1, c.lambda_ga2)

    def _ga_direct(self, mine, pair_of, dxyz_c, g_dxyz):
        """GA term of the direct pipeline: closed-form gradient for all (motion, frame) pairs of the step at once
        (chamfer: 2 (p - nn(p)) per control point; L1: sign / numel), weighted by the number of local renders that
        show the pair, added to the TimeNet output gradient and to `_c_xyz.grad`.  Returns the loss scalar."""
        g, c = self.renderer.gaussians, self.cfg
        pairs, counts = {}, {}
        for (m, v, f) in mine:
            p = pair_of[(m, v, f)]
            pairs[p] = (m, f)
            counts[p] = counts.get(p, 0) + 1
        rows = sorted(pairs)
        idx = torch.tensor(rows, device=self.device)
        cnt = torch.tensor([float(counts[p]) for p in rows], device=self.device)
        ori = torch.stack([self.cpts_s1[pairs[p][0], pairs[p][1]] for p in rows])  # [P, M, 3]
        cp = g._c_xyz.detach()[None] + dxyz_c[idx]
        if c.ga_chamfer:
            d2 = torch.cdist(cp, ori).square()
            dmin, nn = d2.min(dim=2)
            near = torch.gather(ori, 1, nn[..., None].expand(-1, -1, 3))
            # the minimum is recomputed from the gathered points (cdist's expansion loses digits for close pairs)
            diff = cp - near
            loss = c.lambda_ga1 * (cnt * diff.square().sum(dim=(1, 2))).sum()
            grad = (2.0 * c.lambda_ga1) * cnt[:, None, None] * diff
        else:
            diff = cp - ori
            loss = c.lambda_ga2 * (cnt * diff.abs().mean(dim=(1, 2))).sum()
            grad = (c.lambda_ga2 / diff[0].numel()) * cnt[:, None, None] * torch.sign(diff)
        g_dxyz.index_add_(0, idx, grad)
        g._c_xyz.grad.add_(grad.sum(dim=0))
        return loss

    def lpips_metric(self):
        """The LPIPS module of `use_lpips` (main_train_dimo.py:150).  Assign `trainer.lpips` a module carrying the
        published weights; without one a fixed random-weight instance stands in (the wiring, not the metric)."""
        if getattr(self, "lpips", None) is None:
            from .lpips_vgg import LPIPS
            gen = torch.random.get_rng_state()
            torch.manual_seed(1234)
            self.lpips = LPIPS().to(self.device)
            torch.random.set_rng_state(gen)
        return self.lpips

    def regularizer_loss(self, m):
        """ARAP term of one motion (main_train_dimo.py:374-384); None when switched off / outside its window."""
        c = self.cfg
        if not c.use_arap:
            return None
        if (self.stage == "s1" and self.step > c.arap_start_iter_s1) or \
                (self.stage == "s2" and self.step < c.arap_end_iter_s2):
            from .regularizers import arap_loss_v2
            err, _ = arap_loss_v2(self.renderer.gaussians, stage=self.stage, latent_index=m)
            return c.lambda_arap * err
        return None

    def all_reduce_grads(self):
        """The step's collective: the flat gradient bucket behind its 4 flag floats (the overflow flag), SUM.  When the
        per-Gaussian head of the bucket (5.6 of 8.2 MB at N = 100 k) went out early -- `all_reduce_point_grads_async`,
        or the side-stream fold of `_forward_backward_direct` -- only the tail is left here."""
        if self.world > 1:
            g = self.renderer.gaussians
            ext = g.flat_grads_ext
            pending, self._pending_ar = getattr(self, "_pending_ar", None), None
            if pending is not None:
                work, split = pending
                dist.all_reduce(ext[g.FLAG_WORDS + split:], op=dist.ReduceOp.SUM, group=self.pg)
                if work is not None:
                    work.wait()
            else:
                dist.all_reduce(ext, op=dist.ReduceOp.SUM, group=self.pg)

    def _set_grad_flag(self, tot):
        """The overflow words of this rank's renders -> the flag float that leads the bucket (SUM over the ranks: any
        non-zero word skips the update everywhere)."""
        g = self.renderer.gaussians
        if tot is not None and tot.numel() > 0:
            g.grad_flag.copy_(tot[:, 1].max().to(torch.float32))
        else:
            g.grad_flag.zero_()

    def all_reduce_point_grads_async(self, tot=None):
        """Starts the all-reduce of the flag + the per-Gaussian gradients (the head of the bucket) as soon as the last
        skinning backward has accumulated into them; it runs on the collective library's stream while this stream goes
        on with the TimeNet backward.  `all_reduce_grads` completes the step's reduction."""
        g = self.renderer.gaussians
        split = getattr(g, "flat_split", 0)
        if self.world > 1 and self._flat_adam and split > 0:
            self._set_grad_flag(tot)
            work = dist.all_reduce(g.flat_grads_ext[:g.FLAG_WORDS + split], op=dist.ReduceOp.SUM, group=self.pg,
                                   async_op=True)
            self._pending_ar = (work, split)

    # ------------------------------------------------------------------ forward + backward, two ways
    def _forward_backward_autograd(self, mine, n_img):
        """Reference-shaped path: Renderer.render per triple, torch losses, ONE autograd backward."""
        c, g = self.cfg, self.renderer.gaussians
        loss = None
        by_motion = {}
        deforms = self.batched_deform(mine) if self.stage >= "s2" else {}
        self._last_out = None
        for (m, v, f) in mine:
            out = self.render_triple(m, v, f, deform=deforms.get((m, v, f)))
            self._last_out = out
            if self._ga_active():
                ga = self.ga_loss(out["cpts_t"], m, f)
                loss = ga if loss is None else loss + ga
            gt, mask = self.target(m, v, f)
            w = 1.0 if (v == 0 or f == 0) else 0.5  # reference view / frame weighting (main_train_dimo.py:334)
            rec = by_motion.setdefault(m, ([], [], [], []))
            rec[0].append(out), rec[1].append(gt), rec[2].append(mask), rec[3].append(w)
        for m, (outs, gts, masks, ws) in by_motion.items():
            lm = self.motion_loss(outs, gts, masks, ws, n_img)
            # per-motion terms (KL, ARAP) carry this rank's share of the motion's images, like the mean-type image
            # terms: a motion whose renders are split over ranks is then counted once by the SUM all-reduce
            share = len(outs) / n_img
            if g.vae_latent:
                mu, lv = g._mu[m], g._log_var[m]
                lm = lm + share * c.lambda_kl * (-0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp()))
            reg = self.regularizer_loss(m)
            if reg is not None:
                lm = lm + share * reg
            loss = lm if loss is None else loss + lm
        if loss is not None:
            loss.backward()
        out = self._last_out
        self._last_stats = None
        if out is not None and out["viewspace_points"].grad is not None:
            self._last_stats = (out["radii"], out["viewspace_points"].grad)
        return loss

    def _collect_counts(self):
        """(report, device words) of the renders since the last optimizer step: the (R, overflow) instance counts that
        become the update's skip flag; `report` = (words, pinned host slot, number) when the optimizer's launch hands
        them to the host (CapacityPolicy.collect_report), else None (a device-to-host copy behind the optimizer)."""
        cap = self.renderer.capacity_policy()
        rep = cap.collect_report() if cap is not None else None
        tot = rep[0] if rep is not None else (cap.collect_async(defer_copy=True) if cap is not None else None)
        return rep, tot

    def _knn_on_side(self):
        return self.direct and self._side_knn and self._knn_seeded

    def _executor(self, n_renders):
        from .executor import StepExecutor
        g, c = self.renderer.gaussians, self.cfg
        cap = self.renderer.capacity.next_capacity()
        res = self.render_resolution()
        if self._exec is None or self._exec.max_renders < n_renders or self._exec.N != g._xyz.shape[0] \
                or self._exec.H != res:
            import os
            if self._exec is not None:  # explicitly, before the new one takes the pool's streams (not at some later GC)
                torch.cuda.synchronize()
                self._exec.destroy()
                self._exec = None
            self._exec = StepExecutor(g._xyz.shape[0], g._c_xyz.shape[0], res, res,
                                      max(n_renders, 8), cap, self.device,
                                      # 0 = batched: every stage is one launch over all renders of the step.
                                      # k > 0 = per-render chains on k private streams (3 + the caller's = the 4
                                      # hardware queues HIP exposes; 2 -> 1377, 3 -> 1513, 4 -> 1237 frames/s
                                      # before the batched mode existed)
                                      n_streams=int(os.environ.get("DIMO_EXEC_STREAMS", "-2")))
        self._exec.resize_capacity(cap)
        return self._exec

    def _forward_backward_direct(self, mine, n_img):
        """MI355X pipeline: explicit HIP kernel chains instead of autograd.

        skinning -> project/bin/sort/blend of every render (native executor, renders overlap on private streams and
        write straight into their motion's batch buffers) -> per motion: fused SSIM + fused image losses (emit the
        four gradient images) -> blend/projection backward of every render -> skinning backward ACCUMULATING into the
        flat gradient bucket.  Only the (batched) TimeNet MLP uses autograd.  Same math as
        `_forward_backward_autograd` (tests compare the two)."""
        from . import _lib
        from .image_loss import fused_image_loss, fused_ssim_image_loss, loss_weights
        c, g, L = self.cfg, self.renderer.gaussians, _lib.lib()
        dev, stream = self.device, _lib.current_stream()
        H = W = self.render_resolution()
        f32 = dict(dtype=torch.float32, device=dev)
        n = len(mine)
        ex = self._executor(n)
        s1 = self.stage == "s1"
        if not s1 and self._knn_on_side():
            # the step's KNN (main_train_dimo.py:257-258) on a private stream of the executor, NEXT TO the weight packing
            # and the TimeNet forward on this stream (independent of each other: both feed the skinning; 25 us of the
            # step's serial head).  The motion whose chain runs on that stream follows in order, the others wait for it.
            if ex.ranged:
                self.find_knn(k=4, stream=ex.side_stream(0))
                ex.side_done(0)
            else:
                self.find_knn(k=4)
        ex.set_common(g, self.renderer.bg_color, self.renderer.add_normal, stage1=s1)
        self._mark("start")
        fused_tn = (self.fused_timenet or s1) and len(g._timenet.skips) <= 1
        if s1 and not fused_tn:
            raise RuntimeError("the stage-s1 direct pipeline needs the fused TimeNet (one skip connection)")
        if fused_tn:
            dxyz_c, dquat_c, pair_of, lat = self._fused_deform(mine)
        else:
            self.batched_deform(mine)
            dxyz_all, dquat_all, pair_of = self._deform_batch  # [P,M,3], [P,M,4], triple -> row
            dxyz_c, dquat_c = dxyz_all.detach().contiguous(), dquat_all.detach().contiguous()
        self._mark("timenet_fwd")
        # accumulated by the skinning backward; one zero-fill for both and for the loss accumulator
        o_q = (dxyz_c.numel() + 3) // 4 * 4  # 16-byte aligned start of the quaternion rows
        n_motions = len({t[0] for t in mine})
        # the step's accumulators (TimeNet output gradients, loss words, SSIM sums).  Two buffers alternate: the
        # optimizer's launch of THIS step clears the one the next step will use (FlatAdam zero_extra) -- a fill launch
        # between the TimeNet forward and the renders otherwise.  (`last_loss` of a step reads its buffer: valid until
        # the NEXT step's optimizer has run.)
        need = o_q + dquat_c.numel() + _LOSS_WORDS + n_motions
        if self._flat_adam:
            pair = getattr(self, "_acc_bufs", None)
            if pair is None or pair[0].numel() != need:
                pair = self._acc_bufs = [torch.zeros(need, **f32), torch.zeros(need, **f32)]
                self._acc_at = 0
            self._acc_at ^= 1
            zeroed, self._zero_next = pair[self._acc_at], pair[self._acc_at ^ 1]
        else:
            zeroed = torch.zeros(need, **f32)
        g_dxyz = zeroed[:dxyz_c.numel()].view_as(dxyz_c)
        g_dquat = zeroed[o_q:o_q + dquat_c.numel()].view_as(dquat_c)
        by_motion = {}
        for t in mine:
            by_motion.setdefault(t[0], []).append(t)
        M3, M4 = dxyz_c.shape[1] * 3 * 4, dquat_c.shape[1] * 4 * 4  # row strides in bytes
        HW4 = H * W * 4
        bufs, first = {}, {}
        i = 0
        # one allocation per output for ALL renders of the step (a motion's batch is a slice): the joint loss launches
        # below take the step's images as one tensor
        img_all, depth_all = torch.empty(n, 3, H, W, **f32), torch.empty(n, 1, H, W, **f32)
        normal_all = torch.empty(n, 3, H, W, **f32) if self.renderer.add_normal else None
        alpha_all = torch.empty(n, 1, H, W, **f32)
        for m, trs in by_motion.items():
            B = len(trs)
            img, depth = img_all[i:i + B], depth_all[i:i + B]
            normal = normal_all[i:i + B] if normal_all is not None else None
            alpha = alpha_all[i:i + B]
            bufs[m], first[m] = (img, depth, normal, alpha), i
            for b, (_m, v, f) in enumerate(trs):
                d = ex.descs[i]
                cam = self.cams.get(c.elevation, self.azimuths[v], c.radius, W, H)
                d.view, d.proj, d.campos = (cam.world_view_transform.data_ptr(), cam.full_proj_transform.data_ptr(),
                                            cam.camera_center.data_ptr())
                d.tanfovx, d.tanfovy = cam.tanfovx, cam.tanfovy
                p = pair_of[(m, v, f)]
                d.d_xyz, d.d_rot = dxyz_c.data_ptr() + p * M3, dquat_c.data_ptr() + p * M4
                d.g_d_xyz, d.g_d_rot = g_dxyz.data_ptr() + p * M3, g_dquat.data_ptr() + p * M4
                d.out_color, d.out_depth = img.data_ptr() + b * 3 * HW4, depth.data_ptr() + b * HW4
                d.out_normal = (normal.data_ptr() + b * 3 * HW4) if normal is not None else None
                d.out_alpha = alpha.data_ptr() + b * HW4
                i += 1
        # the target batches are gathered before the renders are launched: the private streams fork from this
        # stream, so whatever is enqueued here so far is visible to the loss kernels they run
        gathered = {}
        for m, trs in by_motion.items():
            gts = [self.target(*t) for t in trs]
            # one target and one mask per image (source_masks[motion][view][frame], main_train_dimo.py:284), handed to
            # the loss kernels as pointer lists: the pool's tensors are not stacked (4 copy kernels per step before)
            gathered[m] = ([x[0].contiguous() for x in gts], [x[1].contiguous() for x in gts])
            self._const(-c.lambda_ssim * (len(trs) / n_img))
        # ... and every buffer the loss kernels write is allocated here, BEFORE the forks: memory handed out later could
        # be a block whose last use is a kernel still pending on this stream, which a private stream would not wait for
        depth_on, normal_on = self._reg_on()
        loss_all = (torch.empty_like(img_all), torch.empty_like(img_all),
                    torch.empty_like(depth_all) if depth_on else None,
                    torch.empty_like(normal_all) if (normal_on and normal_all is not None) else None,
                    torch.empty_like(alpha_all), torch.empty_like(alpha_all))  # (last: the backward's per-pixel S)
        loss_bufs = {m: tuple(None if t is None else t[first[m]:first[m] + len(trs)] for t in loss_all)
                     for m, trs in by_motion.items()}
        # Every motion's chain -- forward, losses, rasterizer and skinning backward -- runs in order on its own private
        # stream (the default: a motion's backward overlaps the other motion's losses).  `_joint_bwd` (bench.py's roofline
        # pass): the forwards and losses as above, then THIS stream waits for all of them and runs the rasterizer backward
        # as launches over up to 8 of the step's renders -- the kernel the roofline is quoted on then runs alone on the
        # device.  (Measured and removed: one SSIM / loss launch over all the step's images, 5860 against 6000 frames/s;
        # the step's last motion on this stream itself, +1.2 % with the joint backward only.)
        in_order = bool(ex.ranged and not c.use_lpips)
        joint_bwd = bool(in_order and self._joint_bwd and n > 0)
        # Scalars produced on THIS stream (GA, KL, ARAP; LPIPS further down): summed apart from `loss_accum`, which the
        # private streams' kernels add to atomically.  They run BEFORE the forward forks: their gradient writes -- GA into
        # the TimeNet-row gradients `g_dxyz` and `_c_xyz.grad`, ARAP through autograd into the control points, the
        # TimeNet and the latents, KL into `_mu` / `_log_var` -- are plain read-modify-writes, and the private streams'
        # skinning backward (atomic adds into the same `g_d_xyz` rows and into `_c_xyz.grad`) must be ordered behind
        # them; every private stream forks from this one below.
        extra = None
        if self._ga_active():
            extra = self._ga_direct(mine, pair_of, dxyz_c, g_dxyz)
        for m, trs in by_motion.items():
            share = len(trs) / n_img
            if g.vae_latent:  # KL term of this motion (main_train_dimo.py:355-360): tiny, autograd
                mu, lv = g._mu[m], g._log_var[m]
                kl = share * c.lambda_kl * (-0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp()))
                kl.backward()
                extra = kl.detach() if extra is None else extra + kl.detach()
            reg = self.regularizer_loss(m)  # ARAP on the control points: its own small autograd graph
            if reg is not None:
                reg = share * reg
                reg.backward()
                extra = reg.detach() if extra is None else extra + reg.detach()
        if ex.ranged:  # one batch per motion, on alternating private streams
            for m, trs in by_motion.items():
                ex.forward_range(first[m], len(trs))
        else:
            ex.forward(n)
        self.renderer.capacity.track(ex.total_words(n))

        loss_accum = zeroed[o_q + dquat_c.numel():o_q + dquat_c.numel() + _LOSS_WORDS]
        ssums = zeroed[o_q + dquat_c.numel() + _LOSS_WORDS:]
        ssim_terms, keep = [], []
        skinned = 0
        for m, trs in by_motion.items():
            B = len(trs)
            img, depth, normal, alpha = bufs[m]
            # batched ranges: this motion's losses and rasterizer backward continue ON ITS OWN STREAM, in order behind
            # its renders (no cross-stream event until the skinning backward); otherwise join this stream
            own = ex.range_stream(first[m]) if in_order else None
            if own is None:
                ex.join(first[m], B)  # only this motion's renders: the other motions keep rendering underneath
            stream_m = own if own is not None else stream
            gt, mask = gathered[m]
            share = B / n_img
            # SSIM on the clamped render; its gradient image feeds the loss kernel
            ssum = ssums[len(ssim_terms):len(ssim_terms) + 1]  # zeroed with the step's other accumulators
            coef = self._const(-c.lambda_ssim * share)
            ssim_grad, *grad_out, g_dot = loss_bufs[m]
            if c.use_lpips:  # the LPIPS gradient is added to g_image afterwards: S would be stale
                g_dot = None
            if self._fused_loss and B <= 64 and not c.use_lpips:
                pass  # (the SSIM term runs inside the loss launch below)
            elif B <= 32:
                _lib.check(L.dimo_ssim_forward_backward_images(B, 3, H, W, 1 | 2, _lib.ptr(img), _lib.ptr_array(gt),
                                                               _lib.ptr(coef), _lib.ptr(ssum), _lib.ptr(ssim_grad),
                                                               stream_m), "dimo_ssim_forward_backward_images")
            else:
                _lib.check(L.dimo_ssim_forward_backward(B, 3, H, W, 1 | 2, _lib.ptr(img), _lib.ptr(torch.stack(gt)),
                                                        _lib.ptr(coef), _lib.ptr(ssum), _lib.ptr(ssim_grad), stream_m),
                           "dimo_ssim_forward_backward")
            ssim_terms.append((ssum, c.lambda_ssim * share, float(B * 3 * H * W)))
            w_mse = [c.lambda_mse * (1.0 if (v == 0 or f == 0) else 0.5) / (3 * H * W) for (_m, v, f) in trs]
            if self._fused_loss and B <= 64 and not c.use_lpips:  # SSIM + every other image term: one tile pass
                gi, gd, gn, ga = fused_ssim_image_loss(img, depth if depth_on else None, normal if normal_on else None,
                                                       alpha, gt, mask, w_mse,
                                                       loss_weights(c, B, n_img, H, W, depth_on, normal_on), coef, ssum,
                                                       loss_accum, out=tuple(grad_out), stream=stream_m, g_dot=g_dot)
            else:
                gi, gd, gn, ga = fused_image_loss(img, depth if depth_on else None, normal if normal_on else None,
                                                  alpha, gt, mask, w_mse,
                                                  loss_weights(c, B, n_img, H, W, depth_on, normal_on), ssim_grad,
                                                  loss_accum, out=tuple(grad_out), stream=stream_m, g_dot=g_dot)
            keep.append((gi, gd, gn, ga, ssim_grad, g_dot))
            if c.use_lpips:  # torch (MIOpen) on this stream, on the clamped render; its gradient joins the image's
                x = img.detach().clamp(0.0, 1.0).requires_grad_(True)
                lp = c.lambda_lpips * share * self.lpips_metric()(x, torch.stack(gt)).mean()
                (g_lp,) = torch.autograd.grad(lp, x)
                gi.add_(g_lp * ((img >= 0.0) & (img <= 1.0)))
                extra = lp.detach() if extra is None else extra + lp.detach()
            for b in range(B):
                d = ex.descs[first[m] + b]
                d.g_color, d.g_alpha = gi.data_ptr() + b * 3 * HW4, ga.data_ptr() + b * HW4
                d.g_depth = (gd.data_ptr() + b * HW4) if gd is not None else None
                d.g_normal = (gn.data_ptr() + b * 3 * HW4) if gn is not None else None
                d.g_dot = (g_dot.data_ptr() + b * HW4) if g_dot is not None else None
            if joint_bwd:
                pass  # one launch chain over all the step's renders, below
            elif own is not None:
                ex.backward_launch_in_order(first[m], B)  # (on the stream the motion's chain runs on)
                if self._skin_in_order:  # ... and its skinning backward behind it (writes nothing shared)
                    ex.backward_skinning_in_order(first[m], B)
                    skinned += 1
            elif ex.ranged or not ex.batched:
                ex.backward_launch(first[m], B)  # overlaps with the next motion's losses on this stream
        self._mark("losses+launch")
        if joint_bwd:
            ex.backward_launch_joint(0, n)
            ex.backward_accumulate(0, n)
        elif ex.batched and not ex.ranged:
            ex.backward_launch(0, n)
            ex.backward_accumulate(0, n)
        elif skinned and skinned == len(by_motion):
            # every motion is skinned already: ONE fold over the step's renders
            side = ex.private_stream(0) if (self._split_adam and not s1 and self._flat_adam and g.flat_split > 0) else None
            if side is not None:
                # ... on private stream 0, followed there by the optimizer's update of the per-Gaussian head of the
                # bucket (its gradients are final with the fold), NEXT TO the TimeNet backward on this stream, which only
                # needs the motions' TimeNet-row gradients; the tail of the bucket is updated after it (train_step).
                # With several ranks the head's all-reduce (flag words + per-Gaussian gradients) sits between the two on
                # that stream: RCCL's stream waits for the fold, the optimizer's launch for RCCL.
                ex.join_ranges(0, n)
                ex.backward_accumulate(0, n, stream=side)
                counts = self._collect_counts()
                skip = counts[1]
                if self.world > 1:
                    with torch.cuda.stream(self._external_stream(side)):
                        self._set_grad_flag(counts[1])
                        work = dist.all_reduce(g.flat_grads_ext[:g.FLAG_WORDS + g.flat_split], op=dist.ReduceOp.SUM,
                                               group=self.pg, async_op=True)
                        work.wait()  # (the side stream waits; the host only where the backend has no streams: gloo)
                    self._pending_ar = (None, g.flat_split)
                    skip = g.grad_flag.view(torch.int32)
                self.optimizer.step(skip_flags=skip, zero_grad=True, part=("head", g.flat_split), stream=side)
                ex.side_done(0)
                self._adam_head = counts
            else:
                ex.backward_accumulate(0, n)
        else:
            for m, trs in by_motion.items():
                ex.backward_accumulate(first[m], len(trs))
        self._mark("raster_bwd+skinning_bwd")
        if not s1 and getattr(self, "_pending_ar", None) is None and self.world > 1:
            # (stage s1: the TimeNet backward below still adds its INPUT gradient to `_xyz.grad`, the head of
            # the bucket -- found by the two-replica schedule test: the ranks' Gaussian counts drifted apart)
            self._early_counts = self._collect_counts() if self._flat_adam else None
            self.all_reduce_point_grads_async(self._early_counts[1] if self._early_counts else None)
        # the s1 densification statistics come from the step's last render (main_train_dimo.py:429-431)
        self._last_stats = None
        if s1 and n > 0:
            last_slot = ex.slots[n - 1]
            self._last_stats = (last_slot["radii"], last_slot["g_means2D"])
        # TimeNet backward for all renders at once; the gradient w.r.t. its input points goes to the control points
        # (s2) or to the Gaussians themselves (s1)
        g_pts = g._xyz.grad if s1 else g._c_xyz.grad
        if mine and fused_tn:
            if lat is not None:  # VAE latents: re-parameterised rows, their gradient continues through autograd
                g_lat = torch.zeros_like(lat)
                self._fused_tn.backward(g_dxyz, g_dquat, g_pts, g_lat)
                lat.backward(g_lat)
            else:
                self._fused_tn.backward(g_dxyz, g_dquat, g_pts, g._latent_codes.grad)
        elif mine:
            torch.autograd.backward([dxyz_all, dquat_all], [g_dxyz, g_dquat])
        self._mark("timenet_bwd")
        # the scalar loss is only read for logging: it is assembled from these parts when `last_loss` is looked at
        # (a dozen 4-us elementwise launches per step otherwise)
        return _LazyLoss(loss_accum, ssim_terms, extra)

    def _external_stream(self, handle):
        """torch's view of one of the executor's private streams (for the collective the step enqueues there)."""
        st = self._ext_streams.get(handle)
        if st is None:
            st = self._ext_streams[handle] = torch.cuda.ExternalStream(handle)
        return st

    def _mark(self, name):
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))

    def _const(self, value):
        """Cached 1-element device tensors for scalar kernel arguments."""
        t = self._consts.get(value)
        if t is None:
            t = self._consts[value] = torch.tensor([value], dtype=torch.float32, device=self.device)
        return t

    # ------------------------------------------------------------------ the step
    def train_step(self, triples=None):
        """Runs one optimisation step; returns the number of renders THIS rank performed."""
        g = self.renderer.gaussians
        cap = self.renderer.capacity_policy()
        c = self.cfg
        if self._flat_adam and cap is not None:
            bad = cap.poll()  # last step's instance counts (copied asynchronously)
            if bad:  # that update was skipped on the device; the capacity bound has been raised.  The optimizer's
                # bias corrections do not depend on this read-back: the device counts its own skipped launches
                # (adam.hip), identically on every replica because the flag travels through the all-reduce
                self.skipped_steps += bad
                self.optimizer.skipped_host += bad
        if self.stage == "s1" and self.step % c.FPS_iter == 0:  # main_train_dimo.py:227-228
            self.fps(c.num_cpts)
        if self.stage == "s2" and self.step == 0 and c.add_ga and self.cpts_s1 is None:  # main_train_dimo.py:231-244
            self.cache_cpts_s1()
        self.step += 1
        g.update_learning_rate(self.step, self.stage)
        if self.stage == "s2" and self.step < 1000:  # main_train_dimo.py:250-253
            for grp in self.optimizer.param_groups:
                if grp["name"] == "xyz":
                    grp["lr"] = 0.0002
        if self.stage >= "s2" and not self._knn_on_side():
            self.find_knn(k=4)
        if triples is None:
            triples = self.sample()
        mine = shard(triples, self.rank, self.world)
        # the rank whose slice ends with the step's LAST triple: its last render feeds the s1 densification statistics
        n_t = len(triples)
        self._stats_owner = max((r for r in range(self.world) if (n_t * (r + 1)) // self.world > (n_t * r) // self.world),
                                default=0)
        n_img = max(1, len(triples) // max(1, len({t[0] for t in triples})))  # images per motion (b^2)

        if self.direct:
            loss = self._forward_backward_direct(mine, n_img)
        else:
            loss = self._forward_backward_autograd(mine, n_img)
        if self._flat_adam:
            # no host sync at all: the overflow words of this step's renders become a device-side skip flag that
            # travels through the all-reduce; the host looks at them one step later (CapacityPolicy.poll)
            head, self._adam_head = getattr(self, "_adam_head", None), None
            early, self._early_counts = getattr(self, "_early_counts", None), None
            rep, tot = head if head is not None else (early if early is not None else self._collect_counts())
            # (device scalar, no read-back: a step whose renders overflowed must not feed the densification statistics --
            # stage s1 only, where they are gathered)
            self._step_overflow = tot[:, 1].max() if (tot is not None and self.stage == "s1") else None
            zero_next, self._zero_next = getattr(self, "_zero_next", None), None
            finals_before = getattr(self.optimizer, "finals", None)
            if self.world > 1:  # the flag rides at the head of the gradient bucket through the all-reduce
                if getattr(self, "_pending_ar", None) is None:  # (nothing went out early: the whole bucket now)
                    self._set_grad_flag(tot)
                timed = getattr(self, "time_allreduce", False) and self.device.type == "cuda"
                if timed:  # exposed time of the collective on this stream (bench.py reports the mean)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                self.all_reduce_grads()
                if timed:
                    e1.record()
                    self.allreduce_events.append((e0, e1))
                flag = g.grad_flag.view(torch.int32)
                if head is not None:  # the head of the bucket has been updated under the TimeNet backward: the tail now
                    self._exec.wait_side(0)
                    self.optimizer.step(skip_flags=flag, zero_grad=True, report=rep, zero_extra=zero_next,
                                        part=("tail", g.flat_split))
                else:
                    self.optimizer.step(skip_flags=flag, zero_grad=True, report=rep, zero_extra=zero_next)
            elif head is not None:  # the head of the bucket has been updated under the TimeNet backward: the tail now
                self._exec.wait_side(0)
                self.optimizer.step(skip_flags=tot, zero_grad=True, report=rep, zero_extra=zero_next,
                                    part=("tail", g.flat_split))
            else:  # one rank: Adam reads the renders' (R, overflow) words directly
                self.optimizer.step(skip_flags=tot, zero_grad=True, report=rep, zero_extra=zero_next)
            if zero_next is not None and getattr(self.optimizer, "finals", None) == finals_before:
                zero_next.zero_()  # (the optimizer's launch did not run -- a caller replaced `step`: clear them here)
            if cap is not None and rep is None:
                cap.start_copy()  # the words' copy for the host (poll, next step), behind the optimizer
            self._mark("allreduce+adam")
        else:
            if cap is not None and not cap.check():  # host sync; parameters are still untouched
                g.zero_grad()
                raise RuntimeError("instance capacity overflow: CapacityPolicy grew its bound, redo the step")
            self.all_reduce_grads()
            self.optimizer.step()
            g.zero_grad()
        self._last_loss = loss
        self.densify_schedule()
        return len(mine)

    def _densification_stats(self):
        """main_train_dimo.py:429-431: the statistics come from `out` of the step's LAST render only (the reference's
        loop variable), so under data parallelism the rank that owns the last triple contributes and the others add
        zeros: all-reduce(SUM) of (gradient norm, count), all-reduce(MAX) of the radii -- every replica
        then holds exactly the single-process statistics and densifies identically."""
        g = self.renderer.gaussians
        N = g._xyz.shape[0]
        stats = torch.zeros(N, 2, dtype=torch.float32, device=self.device)
        radii = torch.zeros(N, dtype=torch.float32, device=self.device)
        last = getattr(self, "_last_stats", None)
        if last is not None and self.rank == getattr(self, "_stats_owner", 0):
            r, g2d = last  # radii [N] int32, gradient of the screen-space means [N, 3]
            vis = r > 0
            ovf = getattr(self, "_step_overflow", None)
            if ovf is not None:  # the update of an overflowed step is skipped on the device: so are its statistics
                vis = vis & (ovf == 0)
            stats[vis, 0] = torch.norm(g2d[vis, :2], dim=-1)
            stats[vis, 1] = 1.0
            radii[vis] = r[vis].to(radii.dtype)
        if self.world > 1:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.pg)
            dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=self.pg)
        g.xyz_gradient_accum += stats[:, :1]
        g.denom += stats[:, 1:2]
        g.max_radii2D = torch.max(g.max_radii2D, radii)
        self._last_stats = None

    def densify_schedule(self):
        """Densification / pruning of the reference's two stages (main_train_dimo.py:426-443), after the optimizer
        step.  Deterministic in the parameters (and the shared torch seed for the split draws): replicas stay equal."""
        c, g = self.cfg, self.renderer.gaussians
        n_before, opt_before = g._xyz.shape[0], g.optimizer
        self._densify_schedule(c, g)
        if c.spatial_sort and g.optimizer is not opt_before and g._xyz.shape[0] != n_before:
            g.sort_spatially()  # new rows were appended / rows removed: restore the Morton order (one more rebuild)

    def _densify_schedule(self, c, g):
        if self.stage == "s1":
            fps_iter = c.FPS_iter
            if self.step % fps_iter >= c.density_start_iter and self.step <= c.density_end_iter:
                self._densification_stats()
                if self.step % c.densification_interval == 0:
                    if self.world > 1:  # rank-identical split draws, whatever else consumed the global generator
                        g.split_generator = torch.Generator(device=self.device).manual_seed(
                            (c.seed * 1_000_003 + self.step) & 0x7FFFFFFF)
                    g.densify_and_prune(c.densify_grad_threshold, min_opacity=c.densify_opacity_threshold_s1, extent=4,
                                        max_screen_size=1)
                if self.step % c.opacity_reset_interval == 0:
                    g.reset_opacity()
        elif self.stage == "s2" and self.step < c.density_end_iter_s2:
            if self.step % c.densification_interval_s2 == 0 and c.init_type == "ag":
                g.prune(min_opacity=c.densify_opacity_threshold_s2, extent=4, max_screen_size=1)
