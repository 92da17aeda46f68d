#!/usr/bin/env python
"""Does the step TRAIN?  A hidden seeded teacher model renders self-consistent targets (dimo_amd.synth.TeacherTargets);
a student runs the reference's two-stage schedule shape on them -- stage s1 (TimeNet on a few Gaussians, FPS,
densify / prune), `prune_s1_end`, `prepare_train_s2` (adaptive re-initialisation around the control points), stage s2
(skinning, opacity prunes): main_train_dimo.py:170-218, 426-443 -- on the HIP direct pipeline, and PSNR against held
targets is measured at the four corners of the schedule.

    python tools/teacher_student.py [--iters-s1 240 --iters-s2 300 --res 64]

Used by tests/test_gpu_trains.py (thresholds) and, at C3 scale, by bench.py's `sustained_teacher` figure."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def student_config(res=64, num_cpts=48, pts_per_cpt=40, motions=3, views=9, frames=8, seed=0):
    from dimo_amd.trainer import TrainConfig
    return TrainConfig(
        stage="s1", num_pts=num_cpts * pts_per_cpt, num_cpts=num_cpts, num_pts_per_cpt=pts_per_cpt,
        num_motions=motions, num_views=views, num_frames=frames, motions_per_step=2, views_per_step=2, frames_per_step=2,
        resolution=res, seed=seed,
        # the reference's schedule (FPS every 1000 steps, density window [100, 1000], densify every 100, s2 prune every
        # 1000: configs/train_config.yaml:79-88, run_train_latent.sh) compressed ~12x
        FPS_iter=80, density_start_iter=10, density_end_iter=140, densification_interval=20,
        densification_interval_s2=100, position_lr_max_steps=500)


def psnr(tr, triples):
    """Mean PSNR of the student's renders against the targets of `triples` (no gradient, current stage)."""
    g = tr.renderer.gaussians
    if tr.stage >= "s2":
        g.neighbor_dists = g.neighbor_indices = None
        tr.find_knn(4)
    vals = []
    with torch.no_grad():
        for (m, v, f) in triples:
            out = tr.render_triple(m, v, f)
            gt, _ = tr.target(m, v, f)
            mse = ((out["image"] + 0 - gt) ** 2).mean()
            vals.append(10.0 * torch.log10(1.0 / mse.clamp_min(1e-10)))
    tr.renderer.flush()
    return float(torch.stack(vals).mean())


def run(device="cuda", rank=0, world=1, pg=None, iters_s1=240, iters_s2=300, res=64, seed=0, trace_every=0,
        teacher_kw=None, cfg=None):
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import TeacherTargets, make_teacher
    from dimo_amd.trainer import Trainer
    cfg = cfg or student_config(res=res, seed=seed)
    teacher = make_teacher(device, num_cpts=cfg.num_cpts, pts_per_cpt=cfg.num_pts_per_cpt, num_motions=cfg.num_motions,
                           **(teacher_kw or {}))
    targets = TeacherTargets(teacher, cfg.num_motions, cfg.num_views, cfg.num_frames, cfg.resolution,
                             radius=cfg.radius, fovy=cfg.fovy, elevation=cfg.elevation)
    tg = teacher.gaussians
    with torch.no_grad():
        dx, _ = tg._timenet(tg._c_xyz, 0.5, tg._latent_codes[0])
    info = {"teacher_motion_rms": float(dx.square().sum(-1).mean().sqrt()),
            "teacher_alpha_mean": float(targets.masks.mean())}
    del teacher
    np.random.seed(seed)  # `Renderer.initialize` draws from numpy's global generator, like the reference
    torch.manual_seed(seed)
    rd = Renderer(sh_degree=0, white_background=True, num_latent_code=cfg.num_motions, add_normal=True, device=device)
    rd.initialize(num_pts=cfg.num_cpts, num_cpts=cfg.num_cpts)  # main_train_dimo.py:146: a blob of num_cpts Gaussians
    tr = Trainer(cfg, rd, rank=rank, world_size=world, process_group=pg, targets=targets)
    held = [(m, v, f) for m in range(cfg.num_motions) for v in (0, 3, 6) for f in (1, cfg.num_frames - 2)]
    log = {"psnr_s1_start": psnr(tr, held), "direct_s1": bool(tr.direct)}
    sizes, losses, trace = [], [], []

    def on_step(t):
        sizes.append(int(t.renderer.gaussians._xyz.shape[0]))
        if trace_every and t.step % trace_every == 0:
            losses.append(float(t.last_loss))
            trace.append((t.stage, t.step, psnr(t, held[:6])))

    for _ in range(iters_s1):
        tr.train_step()
        on_step(tr)
    log["psnr_s1_end"] = psnr(tr, held)
    log["gaussians_s1_max"], log["gaussians_s1_end"] = (max(sizes) if sizes else None), (sizes[-1] if sizes else None)
    log["loss_s1_end"] = float(tr.last_loss) if iters_s1 else None
    tr.finish_stage_s1()
    tr.prepare_train_s2(iters_s2)
    log["control_points_s2"] = int(rd.gaussians._c_xyz.shape[0])
    log["gaussians_s2_start"] = int(rd.gaussians._xyz.shape[0])
    log["psnr_s2_start"], log["direct_s2"] = psnr(tr, held), bool(tr.direct)
    for _ in range(iters_s2):
        tr.train_step()
        on_step(tr)
    log["psnr_s2_end"] = psnr(tr, held)
    log["gaussians_s2_end"] = int(rd.gaussians._xyz.shape[0])
    log["loss_s2_end"] = float(tr.last_loss) if iters_s2 else None
    cap = rd.capacity_policy()
    if cap is not None:
        tr.skipped_steps += cap.poll(lag=0)
    log["skipped_steps"] = int(tr.skipped_steps)
    log["finite"] = bool(torch.isfinite(rd.gaussians.flat_params).all())
    log["trace"] = trace
    log.update(info)
    return log, tr


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters-s1", type=int, default=240)
    ap.add_argument("--iters-s2", type=int, default=300)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--trace-every", type=int, default=40)
    a = ap.parse_args()
    out, _ = run(iters_s1=a.iters_s1, iters_s2=a.iters_s2, res=a.res, trace_every=a.trace_every)
    print(json.dumps(out))
