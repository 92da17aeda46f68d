"""`GUI.train_step`'s loop body restated against the drop-in surface -- what a maintainer runs who ONLY swaps the
imports and keeps the reference's trainer as it is.

Own code, same ORDER OF OPERATIONS as main_train_dimo.py:246-417: learning rates, `find_knn` through the `KNN` module,
sampling, then per (motion, view, frame) triple `orbit_camera` + `MiniCam`, `renderer.render(cam, time=, stage=,
latent_index=)`, [geometry-anchor term on out["cpts_t"] with its logged `.item()` (:295-303)], `out[...].unsqueeze(0)`
(:305-318), the targets' `F.interpolate`, `torch.cat` per motion (:320-325); then per motion the per-image
`F.mse_loss` on `batch[...][i * n_frames + j]` (:331-337), SSIM (:343), mask MSE (:350), the two smoothness terms on
`.permute(0, 2, 3, 1)` (:362-372), every `tb_writer.add_scalar(..., x.item(), ...)` read (:345-390); ONE
`loss.backward()`, `optimizer.step()`, `optimizer.zero_grad()` (:415-417).  LPIPS / ARAP / KL are left out exactly as
bench.py's headline leaves them out.

It exists for two consumers: the `-m gpu` test that holds the batching behind `render()` to "one launch chain per
step, same numbers as immediate rendering" for THIS loop, and bench.py's `dropin_frames_per_s`.
"""
import random

import numpy as np
import torch
import torch.nn.functional as F

from dimo_amd.camera import MiniCam, OrbitCamera, orbit_camera
from dimo_amd.knn_cuda import KNN
from dimo_amd.losses import compute_bilateral_normal_smoothness_loss, compute_edge_aware_smoothness_loss, ssim
from dimo_amd.regularizers import chamfer_forward
from dimo_amd.synth import default_azimuths, frame_times


class ReferenceLoop:
    """`opt`: a `TrainConfig` (the reference's option names; `frames_per_step` / `views_per_step` / `motions_per_step`
    stand for batch_size, batch_size, 2 * batch_size).  `targets.get(motion, view, frame)` -> (image [3,H,W], mask
    [1,H,W]) plays `self.source_images[name][view][frame]` / `self.source_masks[...]`.  `log_scalars`: make the
    `.item()` reads the reference's TensorBoard logging makes (each one a host sync)."""

    def __init__(self, opt, renderer, targets, log_scalars=True, cpts_s1=None):
        self.opt, self.renderer, self.targets = opt, renderer, targets
        self.device = renderer.device
        self.stage, self.step = opt.stage, 0
        self.num_frames, self.num_views = opt.num_frames, opt.num_views
        self.azimuths = default_azimuths(opt.num_views)          # main_train_dimo.py:80
        self.source_time = frame_times(opt.num_frames)           # :104
        self.input_videos = list(range(opt.num_motions))
        self.cam = OrbitCamera(opt.resolution, opt.resolution, r=opt.radius, fovy=opt.fovy)  # :60
        self.optimizer = renderer.gaussians.optimizer
        self.log_scalars = log_scalars
        self.scalars = {}            # tb_writer stand-in: name -> last value
        self.cpts_s1 = cpts_s1       # [motions][frames] -> [M, 3] (:231-244), needed for add_ga
        self._py, self._np = random.Random(opt.seed), np.random.default_rng(opt.seed)

    def _log(self, name, value):
        if self.log_scalars:
            self.scalars[name] = value.item()

    def find_knn(self, g, k=4):  # main_train_dimo.py:502-509
        key_pts, gaussian_pts = g._c_xyz.detach(), g._xyz.detach()
        dist, indx = KNN(k=k, transpose_mode=True)(key_pts.unsqueeze(0), gaussian_pts.unsqueeze(0))
        g.neighbor_dists, g.neighbor_indices = dist[0], indx[0]

    def sample(self):
        o = self.opt
        frames = self._py.sample(range(self.num_frames), o.frames_per_step)
        views = self._py.sample(range(self.num_views), o.views_per_step)
        motions = self._np.choice(len(self.input_videos), min(o.motions_per_step, len(self.input_videos)), replace=False)
        return [int(m) for m in motions], views, frames

    def train_step(self, sample=None):
        """One optimisation step; returns the number of renders."""
        o, g = self.opt, self.renderer.gaussians
        self.optimizer = g.optimizer
        self.step += 1
        g.update_learning_rate(self.step, self.stage)
        if self.stage == "s2" and self.step < 1000:
            for grp in self.optimizer.param_groups:
                if grp["name"] == "xyz":
                    grp["lr"] = 0.0002
        if self.stage >= "s2":
            self.find_knn(g, k=4)
        loss = 0
        res = 128 if self.step < 300 else (256 if self.step < 450 else 512)
        if not o.progressive_resolution:
            res = o.resolution
        elif not o.progressive_upsample:
            res = min(res, o.resolution)
        motions, batch_views, batch_frames = sample if sample is not None else self.sample()

        render_images, gt_images, render_masks, gt_masks, render_depths, render_normals = {}, {}, {}, {}, {}, {}
        n_renders = 0
        for m in motions:
            r_img, t_img, r_mask, t_mask, r_depth, r_normal = [], [], [], [], [], []
            for v in batch_views:
                for f in batch_frames:
                    gt_image, gt_mask = self.targets.get(m, v, f)
                    gt_image, gt_mask = gt_image[None].to(self.device), gt_mask[None].to(self.device)
                    pose = orbit_camera(o.elevation, self.azimuths[v], o.radius)
                    cur_cam = MiniCam(pose, res, res, self.cam.fovy, self.cam.fovx, self.cam.near, self.cam.far,
                                      device=self.device)
                    timestamp = self.source_time[f]
                    out = self.renderer.render(cur_cam, time=timestamp, stage=self.stage, latent_index=m)
                    n_renders += 1
                    if o.add_ga and self.stage == "s2" and self.cpts_s1 is not None:
                        cpts_ori = self.cpts_s1[m][f].detach()
                        cpts = out["cpts_t"]
                        if o.ga_chamfer:
                            loss = loss + o.lambda_ga1 * chamfer_forward(cpts[None, ...], cpts_ori[None, ...])
                        else:
                            loss = loss + o.lambda_ga2 * (cpts - cpts_ori).abs().mean()
                        self._log(f"{m}/loss_ga", loss)
                    r_img.append(out["image"].unsqueeze(0))
                    t_img.append(F.interpolate(gt_image, (res, res), mode="bilinear", align_corners=False))
                    r_mask.append(out["alpha"].unsqueeze(0))
                    t_mask.append(F.interpolate(gt_mask, (res, res), mode="bilinear", align_corners=False))
                    r_depth.append(out["depth"].unsqueeze(0))
                    r_normal.append(out["normal"].unsqueeze(0))
            render_images[m], gt_images[m] = torch.cat(r_img, dim=0), torch.cat(t_img, dim=0)
            render_masks[m], gt_masks[m] = torch.cat(r_mask, dim=0), torch.cat(t_mask, dim=0)
            render_depths[m], render_normals[m] = torch.cat(r_depth, dim=0), torch.cat(r_normal, dim=0)

        for m in motions:
            nf = len(batch_frames)
            for i, v in enumerate(batch_views):
                for j, f in enumerate(batch_frames):
                    mse_loss = F.mse_loss(render_images[m][i * nf + j], gt_images[m][i * nf + j])
                    loss = loss + o.lambda_mse * (1.0 if (v == 0 or f == 0) else 0.5) * mse_loss
            ssim_loss = 1 - ssim(render_images[m], gt_images[m])
            loss = loss + o.lambda_ssim * ssim_loss
            self._log(f"{m}/loss_ssim", ssim_loss)
            self._log(f"{m}/loss_mse", mse_loss)
            mask_loss = F.mse_loss(render_masks[m], gt_masks[m])
            loss = loss + o.lambda_mask * mask_loss
            self._log(f"{m}/loss_mask", mask_loss)
            if o.add_depth and self.step > o.depth_reg_start_iter:
                smooth = compute_edge_aware_smoothness_loss(render_depths[m].permute(0, 2, 3, 1),
                                                            render_images[m].permute(0, 2, 3, 1))
                loss = loss + o.lambda_smooth * smooth
                self._log(f"{m}/loss_edge_aware_smooth", smooth)
            if o.add_normal and self.step > o.normal_reg_start_iter:
                bilat = compute_bilateral_normal_smoothness_loss(render_normals[m].permute(0, 2, 3, 1),
                                                                 render_images[m].permute(0, 2, 3, 1))
                loss = loss + o.lambda_bilateral * bilat
                self._log(f"{m}/loss_bilateral_normal_smooth", bilat)
            self._log(f"{m}/loss_total", loss)
            self._log(f"{m}/psnr", 10 * torch.log10(1 / mse_loss))

        loss.backward()
        self.optimizer.step()
        self.optimizer.zero_grad()
        self.last_loss = loss.detach()
        self.last_out = out
        return n_renders
