#!/usr/bin/env python
"""bench.py -- train-step frames/s of the deform-then-render path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 100k synthetic Gaussians
(initialised as the reference initialises them), 512 control points, 512x512, stage s2, 8 renders
(2 motions x 2 views x 2 frames) per GPU per step through deform -> rasterize -> losses -> backward ->
RCCL all-reduce of the flat gradient bucket -> Adam.  "frames" = (motion, view, frame) renders taken through
forward + backward + optimizer.  Weak scaling: per-GPU work is fixed, the step's motion count grows with N.

One JSON line is printed by rank 0.  `roofline` describes the dominant kernel (blend backward): ALGORITHMIC
bytes per launch (B1 = (28+4C) R + (8(C+1)+8) P + (24+4C) V with C = 7 feature channels and
the measured R = tile instances, V = visible Gaussians, P = pixels) divided by its mean duration measured
with HIP events on the launch stream during the timed region.  `cpu_baseline` is the same train step on the
host cores with the CPU oracle injected (a reported baseline, not the optimisation target).
"""
import argparse
import gc
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK = 157.3e12  # MI355X_MICROARCH.md: peak FP32 (vector)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
NFEAT = 7
TIMED = ["preprocess_fwd", "scan", "emit", "sort", "ranges", "blend_fwd", "blend_bwd", "preprocess_bwd", "knn",
         "ssim_fwd", "ssim_bwd", "deform_fwd", "deform_bwd", "image_loss", "adam", "timenet_fwd", "timenet_bwd", "place"]


def make_trainer(device, rank, world, num_pts, resolution, per_gpu=(2, 2, 2), capacity=True, global_batch=None,
                 direct=None, regime="trained"):
    """per_gpu = (motions, views, frames) per GPU and step (weak scaling: the step's motion count grows with the
    world size); global_batch = the reference's `batch_size` b for a FIXED step of 2b x b x b renders sharded over
    the ranks (strong scaling; main_train_dimo.py:266-281: b = 2 -> 16 renders, b = 4 -> 128)."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    if global_batch:
        m, v, f = min(51, 2 * global_batch), global_batch, global_batch
    else:
        m, v, f = min(51, per_gpu[0] * world), per_gpu[1], per_gpu[2]
    cfg = TrainConfig(num_pts=num_pts, resolution=resolution, motions_per_step=m, views_per_step=v, frames_per_step=f)
    # the reference renders at 512^2 from step 450 on whatever the targets' size (main_train_dimo.py:261); a stress
    # shape above that (BASELINE.json configs[4]: 1024^2) is rendered at ITS size
    cfg.progressive_resolution = resolution <= 512
    pol = CapacityPolicy(initial=max(1 << 20, 40 * num_pts)) if capacity else None
    rd = Renderer(sh_degree=0, white_background=True, radius=cfg.radius, num_latent_code=cfg.num_motions,
                  latent_code_dim=cfg.latent_code_dim, add_normal=True, device=device, capacity=pol)
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, regime=regime, num_latent=cfg.num_motions)
    tr = Trainer(cfg, rd, rank=rank, world_size=world, direct=direct)
    # steady state of the schedule: past depth/normal_reg_start_iter (200) every image term is on (10 600 of the
    # reference's 10 000 + 2 800 iterations run that way), past step 1000 the s2 xyz lr rule no longer applies
    tr.step = 1000
    return tr, pol


def kernel_rooflines(timing_iso, N, V, R, P, C=NFEAT, renders_in_window=None, views_per_group=1):
    """name -> algorithmic bytes per render (DESIGN.md section 4 / SURVEY.md 8d formulas), time per launch, GB/s,
    fraction of the HBM peak.  `renders_in_window` = renders this rank took through the pipeline while the timers
    ran: every kernel group gets ITS OWN renders per launch = that / its launch count (the forward stages launch per
    motion batch, the joint backward over all the step's renders); None = one render per launch (isolated run).
    The blend kernels are FP32-VALU bound (see `roofline.note`)."""
    alg = {
        "deform_fwd": 92 * N,
        "preprocess_fwd": 56 * N + 77 * V,
        # binning (dimo_amd/csrc/binning.hip), E = level-1 entries ~ 2.4 N (one per supertile a Gaussian touches):
        # three launches since round 5 (count + scatter are one kernel, the level-2 counts ride in the bucket sort:
        # the "emit" and "ranges" groups have no launches any more and drop out of the table)
        "scan": 20 * N + 16 * N + 16 * 2.4 * N,  # level 1: tiles, depth key, rectangle in (twice); offsets and one 16-byte entry per supertile touched out
        "sort": (16 + 8 + 4) * 2.4 * N,    # per-bucket LDS sort: entries in; (id, tile mask) + a quarter row word per entry out
        "place": (8 + 1) * 2.4 * N + 5 * R,      # level-2 fill: sorted entries + row words in; id + cleared flag per instance out
        "blend_fwd": (28 + 4 * C) * R + 4 * (C + 1) * P + 8 * P,
        "blend_bwd": (28 + 4 * C) * R + (8 * (C + 1) + 8) * P + (24 + 4 * C) * V,
        "preprocess_bwd": (24 + 4 * C) * V + 56 * N + 56 * N + 12 * N,
        "deform_bwd": 200 * N,
        # per image of P pixels: SSIM value + gradient reads image and target (3 channels each), writes the gradient;
        # the fused image losses read 15 planes (image 3, depth, normal 3, alpha, target 3, mask, SSIM gradient 3) and
        # write 9 (the four gradient images + the gradient x value plane)
        "ssim_fwd": 4 * 9 * P,
        "image_loss": 4 * 24 * P,
    }
    if not timing_iso.get("image_loss", (0.0, 0))[1]:
        # the default: SSIM + every image term in ONE tile pass (dimo_ssim_image_loss, timed under "ssim_fwd"): 12
        # planes read (image 3, depth, normal 3, alpha, target 3, mask), 9 written
        alg["ssim_fwd"] = 4 * 21 * P
    out = {}
    for k, b in alg.items():
        ms, n = timing_iso.get(k, (0.0, 0))
        if not n or ms <= 0:
            continue
        t = ms / n * 1e-3
        rpl = (renders_in_window / n) if renders_in_window else 1.0
        # the skinning kernels run once per (motion, frame) GROUP of a batch: the views of a pair share a group
        b = b * (rpl / views_per_group if (k.startswith("deform") and renders_in_window) else rpl)
        out[k] = {"algorithmic_bytes_per_launch": float(b), "renders_per_launch": rpl, "ms": ms / n,
                  "GBps": b / t / 1e9, "frac_of_hbm_peak": b / t / 1e9 / HBM_PEAK_GBS}
    return out


def per_launch(timing, key):
    ms, n = timing.get(key, (0.0, 0))
    return ms / n if n else 0.0


def read_timing():
    from dimo_amd import _lib
    L = _lib.lib()
    out = {}
    for name in TIMED:
        ms, n = C.c_double(0), C.c_int64(0)
        _lib.check(L.dimo_timing_read(name.encode(), C.byref(ms), C.byref(n)), "dimo_timing_read")
        out[name] = (ms.value, n.value)
    return out


def cpu_baseline_c1(steps=20):
    """BASELINE.json configs[0] / BASELINE.md section 2's C1: 1k Gaussians, 1 motion x 1 frame x 1 view at 128^2, deform +
    render + losses + backward + Adam on PyTorch-CPU with the CPU oracle kernels (kind 'port'), `steps` train steps."""
    from dimo_amd.trainer import TrainConfig
    from tests.cpu_backend import make_cpu_trainer
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    cfg = TrainConfig(num_pts=1000, num_cpts=64, resolution=128, motions_per_step=1, views_per_step=1, frames_per_step=1)
    cfg.progressive_resolution = False
    tr = make_cpu_trainer(cfg)
    tr.step = 1000
    tr.train_step()
    t0 = time.time()
    n = sum(tr.train_step() for _ in range(steps))
    dt = time.time() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{steps} train steps of 1 render (C1: 1000 Gaussians, 64 control points, 128x128, 1 motion x 1 "
                      f"view x 1 frame): torch-CPU deform/losses/Adam + C oracle rasterizer; {dt:.1f} s"}


def cpu_baseline(num_pts, resolution, renders=20):
    """Same train step on the host cores: product host logic + CPU oracle kernels (kind 'port')."""
    from dimo_amd.trainer import TrainConfig
    from tests.cpu_backend import make_cpu_trainer
    # 16 threads: beyond that the small torch-CPU ops and the OpenMP rows of the oracle stop scaling
    # (a 256-thread run on the GPU node's host was 35x SLOWER than 8 threads)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    cfg = TrainConfig(num_pts=num_pts, resolution=resolution, motions_per_step=1, views_per_step=1,
                      frames_per_step=renders)
    cfg.progressive_resolution = resolution <= 512
    tr = make_cpu_trainer(cfg)
    tr.step = 1000  # the same point of the schedule as the GPU workload: full render size, every image term on
    t0 = time.time()
    n = tr.train_step()
    dt = time.time() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 train step of {n} renders (1 motion x 1 view x {renders} frames) at {num_pts} Gaussians "
                      f"{resolution}x{resolution}: torch-CPU deform/losses/Adam + C oracle rasterizer "
                      f"(forward OpenMP over {cores} threads, backward single-threaded); {dt:.1f} s"}


def scene_stats(tr, slot=0):
    """What ONE render of the last step looked like to the blend kernels (read from render slot `slot`'s workspaces after a
    device sync): R = tile instances, the mean tile-list length, mean `n_contrib` per pixel (the position of a pixel's
    last contributing entry in its tile's list: how deep the compositing walks), the pixel-entry pairs the walk covers
    (sum over tiles of the deepest pixel's position x 256), the backward's work items (buckets of 64 entries some pixel
    of the tile reaches) and the bytes of blend checkpoints the forward wrote for them (9 planes x 256 pixels x 4 B per
    bucket behind a tile's first)."""
    from dimo_amd.rasterizer import inspect_state
    torch.cuda.synchronize()
    ex = tr._exec
    sl = ex.slots[slot]
    st = inspect_state((sl["geom"], sl["bin"], sl["img"]), ex.N, ex.H, ex.W, ex.r_cap)
    H, W = ex.H, ex.W
    nc = st["n_contrib"].to(torch.int64)
    ty, tx = (H + 15) // 16, (W + 15) // 16
    pad = torch.zeros(ty * 16, tx * 16, dtype=torch.int64, device=nc.device)
    pad[:H, :W] = nc
    deepest = pad.view(ty, 16, tx, 16).amax(dim=(1, 3)).reshape(-1)
    nb = (deepest + 63) // 64
    lens = (st["ranges"][:, 1] - st["ranges"][:, 0]).to(torch.int64).clamp_(min=0)
    # buckets per backward item as the forward picks them per render slot (blend.hip, adaptive chains: the rule below
    # on the PREVIOUS render of the slot -- the same scene in steady state): a checkpoint is written where a chain starts
    chain = 4 if int(nb.sum()) >= 8 * nb.numel() else (2 if int(nb.sum()) >= 4 * nb.numel() else 1)
    return {"R_tile_instances": int(st["total"][0].item()) & 0xFFFFFFFF,
            "mean_tile_list": float(lens.float().mean()), "n_contrib_mean": float(nc.float().mean()),
            "pixel_entry_pairs": int((deepest * 256).sum()), "blend_bwd_items": int(((nb + chain - 1) // chain).sum()),
            "buckets_per_tile_mean": float(nb.float().mean()), "buckets_per_backward_item": chain,
            "checkpoint_bytes_written": int(((nb - 1).clamp_(min=0) // chain).sum()) * 9 * 256 * 4,
            "instance_capacity": int(ex.r_cap)}


def regime_figures(device, num_pts, resolution, per_gpu, regime, steps, pmc=True):
    """The C3 step in one of SURVEY 8d's synthetic regimes, in THIS process: frames/s over `steps` steps (same protocol
    as the headline: set-up steps, then `steps` timed between device syncs), the blend kernels' time per launch (the
    backward as ONE launch over the step's renders, alone on the device: the roofline clock; the forward per motion
    batch inside the schedule), what a render looks like to them (`scene_stats`) and -- optionally -- the rocprofv3
    counters of the backward on that regime."""
    from dimo_amd import _lib
    L = _lib.lib()
    tr, pol = make_trainer(device, 0, 1, num_pts, resolution, per_gpu=per_gpu, regime=regime)
    for _ in range(30):
        tr.train_step()
    torch.cuda.synchronize()
    sk0 = tr.skipped_steps
    t0 = time.perf_counter()
    n = sum(tr.train_step() for _ in range(steps))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    skipped = tr.skipped_steps - sk0
    stats = scene_stats(tr)
    V = None
    with torch.no_grad():
        cam = tr.cams.get(0, tr.azimuths[0], tr.cfg.radius, resolution, resolution)
        tr.find_knn()
        out = tr.renderer.render(cam, time=tr.source_time[3], stage="s2", latent_index=0)
        V = int((out["radii"] > 0).sum())
    if pol is not None:
        pol.check()
    L.dimo_timing_select(None)
    L.dimo_timing_enable(1)
    for _ in range(3):
        tr.train_step()
    torch.cuda.synchronize()
    L.dimo_timing_enable(0)
    t_all = read_timing()
    tr._joint_bwd = True
    for _ in range(2):
        tr.train_step()
    torch.cuda.synchronize()
    L.dimo_timing_select(b"blend_bwd")
    L.dimo_timing_enable(1)
    js, jr = 10, 0
    for _ in range(js):
        jr += tr.train_step()
    torch.cuda.synchronize()
    L.dimo_timing_enable(0)
    bwd_ms, bwd_n = read_timing()["blend_bwd"]
    tr._joint_bwd = False
    rpl = jr / max(bwd_n, 1)
    P = resolution * resolution
    R = int(getattr(pol, "last_r_mean", 0) or stats["R_tile_instances"])
    alg = ((28 + 4 * NFEAT) * R + (8 * (NFEAT + 1) + 8) * P + (24 + 4 * NFEAT) * V) * rpl
    avg = bwd_ms / max(bwd_n, 1)
    fwd_ms, fwd_n = t_all["blend_fwd"]
    rps = n / steps
    res = {"frames_per_s": n / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "skipped_steps": skipped,
           "V_visible": V, **stats,
           "blend_bwd_ms_per_launch": avg, "blend_bwd_renders_per_launch": rpl,
           "blend_fwd_ms_per_launch": fwd_ms / max(fwd_n, 1), "blend_fwd_renders_per_launch": 3 * rps / max(fwd_n, 1),
           "kernels_ms_per_launch": {k: (v[0] / v[1] if v[1] else None) for k, v in t_all.items() if v[1]},
           "roofline": {"kernel": "blend_bwd_batched_kernel<true, true>", "algorithmic_bytes_per_launch": alg,
                        "achieved": alg / (avg * 1e-3) / 1e9 if avg > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS if avg > 0 else None,
                        "pixel_entry_pairs_per_s": stats["pixel_entry_pairs"] * rpl / (avg * 1e-3) if avg > 0 else None}}
    del tr
    torch.cuda.empty_cache()
    if pmc:
        live, note = live_pmc(regime=regime)
        if live is not None:
            res["roofline"]["traffic"] = live["hbm_bytes_per_launch"] * rpl / live["renders_per_launch"]
            res["roofline"]["valu"] = live.get("valu")
        res["roofline"]["traffic_source"] = note
    return res


def render_fps(device, num_pts, resolution, rounds=500):
    """The reference's one benchmark-like harness, `GUI.test_fps` (main_test_dimo.py:872-894): one warm-up render, then
    `rounds` calls of `Renderer.render(test_cam, time=0, stage="s2")` on the azimuth-0 orbit camera at 512^2, nothing
    read back -- here under `torch.no_grad()` and with a device sync before the clock stops (the reference's loop has
    none: it times the enqueue).  Two figures: the calls as the drop-in surface runs them (queued behind `render()`, a
    launch chain per 8 calls: dimo_amd/batched_render.py) and one by one (`batch_renders = False`: every call runs its
    own chain before it returns its tensors)."""
    from dimo_amd.camera import MiniCam, OrbitCamera, orbit_camera
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=num_pts, resolution=resolution)
    rd = Renderer(sh_degree=0, white_background=True, radius=cfg.radius, num_latent_code=cfg.num_motions,
                  latent_code_dim=cfg.latent_code_dim, add_normal=True, device=device,
                  capacity=CapacityPolicy(initial=max(1 << 20, 40 * num_pts)))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, regime="trained", num_latent=cfg.num_motions)
    tr = Trainer(cfg, rd)  # (Morton order + the KNN `find_knn` of the harness)
    tr.find_knn(k=4)
    oc = OrbitCamera(resolution, resolution, r=cfg.radius, fovy=cfg.fovy)
    out = {}
    for name, batched in (("through_the_batcher", True), ("one_by_one", False)):
        rd.batch_renders = batched
        with torch.no_grad():
            cam = MiniCam(orbit_camera(cfg.elevation, 0, cfg.radius), resolution, resolution, oc.fovy, oc.fovx, oc.near,
                          oc.far, device=device)
            rd.render(cam, time=0, stage="s2")
            rd.flush()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(rounds):
                rd.render(cam, time=0, stage="s2")
            t_enq = time.perf_counter() - t0
            rd.flush()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[name] = {"fps": rounds / dt, "fps_enqueue_only_as_the_reference_times_it": rounds / t_enq}
    rd.batch_renders = True
    out["what"] = ("GUI.test_fps (main_test_dimo.py:872-894): 1 warm-up + %d Renderer.render(test_cam, time=0, "
                   "stage='s2') calls, %d Gaussians, %d^2, torch.no_grad(), device sync before the clock stops"
                   % (rounds, num_pts, resolution))
    del tr, rd
    torch.cuda.empty_cache()
    return out


def dropin_figures(device, num_pts, resolution, per_gpu, steps=12):
    """frames/s of the C3 step through the drop-in surface ONLY.  Headline (`dropin_frames_per_s`): the reference's own
    loop body, verbatim order of operations including its TensorBoard `.item()` reads (tests/reference_step.py
    restating main_train_dimo.py:246-417; Renderer.render + MiniCam per triple, torch.cat per motion, F.mse_loss per
    image, the ssim / smoothness drop-ins, ONE backward, optimizer.step).  Beside it: the same loop without the logging
    reads, with the geometry-anchor term and its per-render `.item()` (:295-303), and the repo's own reference-shaped
    trainer (`Trainer(direct=False)`: collects the outputs, one fused loss node per motion)."""
    from tests.reference_step import ReferenceLoop
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import SyntheticTargets, init_synthetic_model
    from dimo_amd.trainer import TrainConfig
    out, detail = {}, {}

    def run_loop(log, ga):
        cfg = TrainConfig(num_pts=num_pts, resolution=resolution, motions_per_step=per_gpu[0], views_per_step=per_gpu[1],
                          frames_per_step=per_gpu[2], add_ga=ga)
        cfg.progressive_resolution = resolution <= 512
        rd = Renderer(sh_degree=0, white_background=True, radius=cfg.radius, num_latent_code=cfg.num_motions,
                      latent_code_dim=cfg.latent_code_dim, add_normal=True, device=device,
                      capacity=CapacityPolicy(initial=max(1 << 20, 40 * num_pts)))
        init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, regime="trained", num_latent=cfg.num_motions)
        rd.gaussians.sort_spatially()
        rd.gaussians.training_setup(cfg)
        cpts = None
        if ga:
            c = rd.gaussians._c_xyz.detach()
            cpts = [[c] * cfg.num_frames for _ in range(cfg.num_motions)]
        loop = ReferenceLoop(cfg, rd, SyntheticTargets(resolution, device, seed=0), log_scalars=log, cpts_s1=cpts)
        loop.step = 1000
        for _ in range(4):
            loop.train_step()
        torch.cuda.synchronize()
        b = rd._batcher
        f0, r0 = (b.flushes, b.rendered) if b is not None else (0, 0)
        t0 = time.perf_counter()
        n = sum(loop.train_step() for _ in range(steps))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fl = ((b.flushes - f0), (b.rendered - r0)) if b is not None else (None, None)
        del loop, rd
        torch.cuda.empty_cache()
        return n / dt, fl

    v, fl = run_loop(True, False)
    out["dropin_frames_per_s"] = v
    detail["literal_loop_with_logging_reads"] = v
    detail["launch_chains_per_step"] = (fl[0] / steps) if fl[0] is not None else None
    detail["renders_per_launch_chain"] = (fl[1] / fl[0]) if fl[0] else None
    detail["literal_loop_without_logging_reads"], _ = run_loop(False, False)
    v_ga, fl_ga = run_loop(True, True)
    detail["literal_loop_with_ga_term_and_its_item"] = v_ga
    detail["launch_chains_per_step_with_ga"] = (fl_ga[0] / steps) if fl_ga[0] is not None else None
    tr2, _ = make_trainer(device, 0, 1, num_pts, resolution, per_gpu=per_gpu, capacity=True, direct=False)
    for _ in range(3):
        tr2.train_step()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n2 = sum(tr2.train_step() for _ in range(steps))
    torch.cuda.synchronize()
    detail["collected_outputs_trainer_direct_false"] = n2 / (time.perf_counter() - t2)
    del tr2
    out["dropin_detail"] = detail
    out["dropin_what"] = ("C3 step through the drop-in surface only; headline = the reference's loop body in its own "
                          "order of operations INCLUDING its per-motion tb_writer .item() reads "
                          "(tests/reference_step.py restates main_train_dimo.py:246-417): Renderer.render + MiniCam "
                          "per (motion, view, frame) triple, out[...].unsqueeze(0), torch.cat per motion, F.mse_loss per "
                          "image, ssim / smoothness drop-ins, ONE loss.backward(), FlatAdam.step; the step's renders run "
                          "as %.2f launch chain(s) per step behind render() (lazy outputs with deferred view / cat ops, "
                          "dimo_amd/batched_render.py); %d steps per figure"
                          % (detail["launch_chains_per_step"] or 0.0, steps))
    return out


def teacher_sustained(device, num_pts, resolution, per_gpu, steps):
    """The sustained figure on LEARNABLE targets: a hidden seeded teacher model of the same size (512 control points x
    num_pts / 512 Gaussians, its own TimeNet weights and latents) renders the targets of 8 motions x 9 views x 21
    frames once (resident in HBM); the benchmark's student trains on them for `steps` consecutive steps.  The noise
    targets of the headline workload cannot be learned -- a long run on them drifts (Gaussians blow up,
    instance counts creep) -- these can: frames/s and the PSNR on held renders before / after are reported."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import TeacherTargets, init_synthetic_model, make_teacher
    from dimo_amd.trainer import TrainConfig, Trainer
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import teacher_student as ts
    motions = 8
    cfg = TrainConfig(num_pts=num_pts, resolution=resolution, num_motions=motions, motions_per_step=per_gpu[0],
                      views_per_step=per_gpu[1], frames_per_step=per_gpu[2])
    cfg.progressive_resolution = resolution <= 512
    teacher = make_teacher(device, num_cpts=cfg.num_cpts, pts_per_cpt=max(1, num_pts // cfg.num_cpts),
                           num_motions=motions, scale="dist2")
    t0 = time.perf_counter()
    targets = TeacherTargets(teacher, motions, cfg.num_views, cfg.num_frames, resolution, radius=cfg.radius,
                             fovy=cfg.fovy, elevation=cfg.elevation)
    torch.cuda.synchronize()
    t_targets = time.perf_counter() - t0
    del teacher
    rd = Renderer(sh_degree=0, white_background=True, radius=cfg.radius, num_latent_code=motions,
                  latent_code_dim=cfg.latent_code_dim, add_normal=True, device=device,
                  capacity=CapacityPolicy(initial=max(1 << 20, 40 * num_pts)))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, regime="trained", num_latent=motions)
    tr = Trainer(cfg, rd, targets=targets)
    tr.step = 1000
    held = [(m, v, f) for m in (0, 3, 6) for v in (0, 4) for f in (2, 11)]
    p0 = ts.psnr(tr, held)
    for _ in range(30):
        tr.train_step()
    torch.cuda.synchronize()
    sk0 = tr.skipped_steps
    t1 = time.perf_counter()
    n = sum(tr.train_step() for _ in range(steps))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    cap = rd.capacity_policy()
    if cap is not None:
        tr.skipped_steps += cap.poll(lag=0)
    p1 = ts.psnr(tr, held)
    out = {"frames_per_s": n / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "psnr_start_end": [p0, p1],
           "skipped_steps": tr.skipped_steps - sk0, "gaussians_start_end": [num_pts, int(rd.gaussians._xyz.shape[0])],
           "targets": f"{motions} motions x {cfg.num_views} views x {cfg.num_frames} frames rendered by a hidden teacher "
                      f"model ({num_pts} Gaussians) in {t_targets:.1f} s, resident in HBM",
           "what": "the same training step as the headline on self-consistent (learnable) targets, consecutive steps "
                   "from schedule step 1030 across the stage-s2 prune of step 2000; PSNR of 12 held renders"}
    del tr, rd, targets
    torch.cuda.empty_cache()
    return out


def live_pmc(timeout=150, regime="trained"):
    """HBM traffic and VALU occupancy of the dominant kernel from rocprofv3 PMC counters, collected DURING this run in
    child processes: three passes over tools/pmc_probe.py (the same C3 workload), one counter set each, never combined
    with a trace -- FETCH_SIZE, WRITE_SIZE (corrected with the in-run calibration on a 1 GiB copy, as
    MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE reads half of a wide streaming read on gfx950) and one SQ
    set.  Returns (dict or None, note)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    if any("ROCPROF" in k or k.startswith("ROCP_") for k in os.environ):
        return None, "not collected: this run is itself under rocprofv3"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "not collected: rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_summarise import per_kernel
    tmp = tempfile.mkdtemp(prefix="dimo_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    env["DIMO_XSTREAM"] = "event"  # (counter collection serialises dispatches: no stream-memory waits, see pmc_probe.py)
    sets = {"fetch": "FETCH_SIZE", "write": "WRITE_SIZE",
            "sq": "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"}
    files = {}
    try:
        for name, ctrs in sets.items():
            d = os.path.join(tmp, name)
            cmd = [exe, "--pmc", *ctrs.split(), "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "tools", "pmc_probe.py"), "--steps", "3", "--regime", regime] \
                + (["--no-calibration"] if name == "sq" else [])
            # (its own process group: a pass that does not come back is killed WITH the workload it started)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                proc.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, f"not collected: the {name} pass did not finish within {timeout} s"
            hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not hits:
                return None, f"not collected: the {name} pass left no counter file"
            files[name] = hits[0]
        pat = "blend_bwd_batched_kernel"
        find = lambda dct: next((v for k, v in dct.items() if pat in k), [])
        mean = lambda xs: sum(xs) / len(xs) if xs else None
        fetch, write = per_kernel(files["fetch"], "FETCH_SIZE"), per_kernel(files["write"], "WRITE_SIZE")
        big = lambda dct: [v for k, vs in dct.items() if "__amd_rocclr_copyBuffer" in k for v in vs if v > 100000.0]
        cal_f, cal_w = mean(big(fetch)), mean(big(write))
        kf = (float(1 << 20) / cal_f) if cal_f else 2.0
        kw = (float(1 << 20) / cal_w) if cal_w else 1.0
        fk, wk = mean(find(fetch)), mean(find(write))
        if fk is None or wk is None:
            return None, "not collected: the kernel is not in the counter files"
        out = {"hbm_bytes_per_launch": (fk * kf + wk * kw) * 1024.0, "renders_per_launch": 8.0,
               "fetch_factor": kf, "write_factor": kw, "launches": len(find(fetch))}
        sq = {c: mean(find(per_kernel(files["sq"], c))) for c in sets["sq"].split()}
        g = (sq.get("GRBM_GUI_ACTIVE") or 0.0) / 8.0  # (summed over the 8 XCDs)
        if g and sq.get("SQ_WAVE_CYCLES"):
            out["valu"] = {"valu_busy_at_4_cycles_per_inst": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (1024.0 * g),
                           "valu_busy_at_2_cycles_per_inst": 2.0 * sq["SQ_INSTS_VALU"] / (1024.0 * g),
                           "waves_per_simd": 4.0 * sq["SQ_WAVE_CYCLES"] / (1024.0 * g),
                           "valu_insts_per_wave": sq["SQ_INSTS_VALU"] / max(sq["SQ_WAVES"], 1.0),
                           "wave_wait_any": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                           "source": "rocprofv3 --pmc SQ pass of tools/pmc_probe.py, collected during this run"}
        return out, ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes over tools/pmc_probe.py = this workload, "
                     "child processes of THIS run, after the timed region), FETCH_SIZE x %.3f / WRITE_SIZE x %.3f from "
                     "the in-run 1 GiB copy calibration; %d launches of 8 renders" % (kf, kw, out["launches"]))
    except Exception as e:  # the counters must never take the throughput number down with them
        return None, f"not collected: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def self_launch(n):
    """Re-executes this script as n ranks under torch.distributed.run (127.0.0.1, a free port); returns its exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--num-pts", type=int, default=100000)
    ap.add_argument("--resolution", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in (autograd) path's frames/s")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for the tests)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: the reference's batch_size b -> a FIXED step of 2b x b x b renders (b = 2: 16, "
                         "b = 4: 128) sharded over the ranks, instead of 8 renders per GPU")
    ap.add_argument("--per-gpu", default="2,2,2", help="weak scaling: motions,views,frames per GPU and step")
    ap.add_argument("--sustained-steps", type=int, default=1200,
                    help="consecutive steps of the `sustained` figure (0: skip); they cross a stage-s2 prune")
    ap.add_argument("--no-teacher", action="store_true", help="skip the teacher-target sustained figure")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="take roofline.traffic / roofline.valu from profiles/ instead of collecting them during the run")
    ap.add_argument("--sync-exact", action="store_true", help="size sort buffers by reading R back (one sync per render)")
    ap.add_argument("--regime", default="trained", choices=("trained", "init", "low"),
                    help="opacities of the synthetic Gaussians the HEADLINE is quoted on (SURVEY 8d): trained = "
                         "sigmoid(U(-2, 4)), lists saturate early (every round's headline); init = every opacity 0.05, "
                         "the reference's own initial state; low = U(0.01, 0.1)")
    ap.add_argument("--main-stream-allreduce", action="store_true",
                    help="several ranks: the plain schedule -- fold, ONE all-reduce of the whole gradient bucket and one "
                         "Adam launch on the caller's stream (Trainer._split_adam = False) -- instead of the head's "
                         "all-reduce on a private stream under the TimeNet backward (the fallback, never needed so far)")
    ap.add_argument("--no-regimes", action="store_true",
                    help="skip the `regimes` block (the same step in the init regime, measured in this process)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, rendezvous on a free
        # loopback port) through the same torch.distributed.run form the driver uses; rank 0's JSON line is this
        # process's output, the exit code is the launcher's
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    local = local % torch.cuda.device_count()  # (several ranks may share a device under --backend gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from dimo_amd import _lib
    L = _lib.lib()
    per_gpu = tuple(int(x) for x in args.per_gpu.split(","))
    tr, pol = make_trainer(device, rank, world, args.num_pts, args.resolution, per_gpu=per_gpu,
                           capacity=not args.sync_exact, global_batch=args.global_batch or None, regime=args.regime)
    tr.time_allreduce = world > 1  # event pairs around the step's collective: its EXPOSED time on this stream
    if args.main_stream_allreduce:
        tr._split_adam = False

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, before the W warm-up steps of the contract: the first ~15 steps of a fresh process are not steady state
    # (code objects load on first launch, the caching allocator and the instance-capacity policy settle, the first
    # KNN runs unseeded: 2.7 / 2.0 / 1.6 ms for steps 2-4 and host hiccups of ~1 ms up to step 14 against 1.27 ms
    # afterwards -- tools/steps_probe.py), and the driver's 5 + 20 steps would time exactly those
    PRESTEPS = 20
    for _ in range(PRESTEPS):
        tr.train_step()
    # a full (generation-2) Python GC pass walks every object torch created at import time: ~45 ms, i.e. ten
    # training steps, whenever it happens to fall into the timed region.  Long-running training loops park the
    # start-up objects in the permanent generation for the same reason.  It runs HERE, in front of the warm-up steps,
    # not between them and the timed region: the device sits idle for those 45 ms, its clocks drop, and the first ~15
    # steps behind the gap ran 1.08-1.10 ms against 0.98 later (step-end times of `DIMO_BENCH_TRACE=1`, round 5:
    # `--steps 20` read 7300-7470 frames/s where `--steps 100` read 7960-8030 on the same box) -- the warm-up steps
    # exist to hand the timed region a device in steady state.
    gc.collect()
    gc.freeze()
    barrier()
    # ... and the device is taken back to its steady state before the contract's W warm-up steps: with W = 5 (the
    # driver's flags) five steps behind the collector's gap still left the first ten timed steps at 1.02-1.03 ms
    # against 0.97 later
    # (round-5 advice: the figure WITHOUT these settle steps is reported too -- the first 5 of them stand in for the
    # warm-up of the old protocol, the other 15 are timed the way its timed region was: `value_behind_idle_gap`)
    SETTLE = 20
    for _ in range(5):
        tr.train_step()
    barrier()
    t_gap = time.perf_counter()
    gap_renders = 0
    for _ in range(SETTLE - 5):
        gap_renders += tr.train_step()
    barrier()
    t_gap = time.perf_counter() - t_gap
    gc.collect()  # (what the settle steps left behind: a few hundred objects, well under a millisecond)
    gc.freeze()
    for _ in range(args.warmup):
        tr.train_step()
    skipped_warmup = tr.skipped_steps
    tr.allreduce_events = []
    barrier()
    # the timed region carries event pairs around the dominant kernel only (two event records per launch are
    # not free); the other kernel groups are timed in three extra steps after it
    L.dimo_timing_select(b"blend_bwd")
    L.dimo_timing_enable(1)
    t0 = time.perf_counter()
    renders = 0
    trace = [] if os.environ.get("DIMO_BENCH_TRACE") else None
    for _ in range(args.steps):
        renders += tr.train_step()
        if trace is not None:
            trace.append(time.perf_counter() - t0)
    barrier()
    elapsed = time.perf_counter() - t0
    skipped_timed = tr.skipped_steps - skipped_warmup  # (read with a lag of one step: the last step is not in yet)
    R_step = getattr(pol, "last_r_mean", None) if pol is not None else None
    tr_views = 1 if tr.cfg.vae_latent else tr.cfg.views_per_step
    ar_ms = [a.elapsed_time(b) for a, b in tr.allreduce_events] if getattr(tr, "allreduce_events", None) else []
    tr.time_allreduce = False
    if trace is not None:
        print("step-end host times (ms):", " ".join(f"{1e3 * x:.2f}" for x in trace), f"| total {1e3 * elapsed:.2f}",
              f"skipped={tr.skipped_steps}", file=sys.stderr)
    L.dimo_timing_enable(0)
    tt = torch.tensor([elapsed, float(renders), 1.0], dtype=torch.float64, device=device)
    ranks_seen = 1
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed, renders_total, ranks_seen = float(tmax[0]), float(tt[1]), int(round(float(tt[2])))
    else:
        renders_total = float(renders)
    timing = read_timing()
    # The roofline clock.  In the timed schedule every motion's rasterizer backward (4 renders) runs on its own stream
    # and OVERLAPS the other motion's kernels: its live duration is not the kernel's own.  The same steps are therefore
    # taken once more with the backward as ONE launch over the step's renders behind everything else
    # (`Trainer._joint_bwd`: the kernel runs alone on the device, which is also what rocprofv3 sees of it): that
    # duration is the one `roofline.achieved` is computed from; the timed region's is reported beside it.
    joint0 = tr._joint_bwd
    tr._joint_bwd = True
    for _ in range(3):
        tr.train_step()
    barrier()
    L.dimo_timing_select(b"blend_bwd")
    L.dimo_timing_enable(1)
    serial_steps, serial_renders = max(10, min(args.steps, 30)), 0
    for _ in range(serial_steps):
        serial_renders += tr.train_step()
    barrier()
    L.dimo_timing_enable(0)
    timing_serial = read_timing()
    tr._joint_bwd = joint0
    for _ in range(2):
        tr.train_step()
    barrier()
    torch.cuda.reset_peak_memory_stats(device)
    # step latency distribution (median + p10/p90), one device sync per step, outside the timed region
    lat = []
    for _ in range(min(args.steps, 50)):
        barrier()
        t1 = time.perf_counter()
        tr.train_step()
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t1))
    lat.sort()
    pct = lambda q: lat[min(len(lat) - 1, int(q * len(lat)))]
    L.dimo_timing_select(None)
    L.dimo_timing_enable(1)
    for _ in range(3):
        tr.train_step()
    barrier()
    L.dimo_timing_enable(0)
    timing_all = read_timing()
    skipped_total = tr.skipped_steps
    peak_mem = torch.cuda.max_memory_allocated(device)

    R = V = cam = None
    timing_iso = {}
    iso_ms = iso_n = 0
    if rank == 0:
        # measured R (tile instances) and V (visible Gaussians) of this workload, outside the timed region
        cam = tr.cams.get(0, tr.azimuths[0], tr.cfg.radius, args.resolution, args.resolution)
        tr.find_knn()
        with torch.no_grad():
            out = tr.renderer.render(cam, time=tr.source_time[3], stage="s2", latent_index=0)
        V = int((out["radii"] > 0).sum())
        if pol is not None:
            pol.check()
            R = int(getattr(pol, "last_r_mean", 0)) or None
        else:
            R = None
        if R is None:
            from dimo_amd.rasterizer import CapacityPolicy
            p2 = CapacityPolicy(initial=pol.capacity if pol else 1 << 24)
            tr.renderer.capacity = p2
            with torch.no_grad():
                tr.renderer.render(cam, time=tr.source_time[3], stage="s2", latent_index=0)
            tot = torch.cat(p2._pending).cpu()
            R = int(tot[:, 0].max())
        # the same kernel alone on the device (the timed region overlaps 3-4 renders, which stretches each
        # individual launch): 5 isolated fwd+bwd renders on one stream, outside the timed region
        L.dimo_timing_enable(1)
        for _ in range(5):
            o = tr.renderer.render(cam, time=tr.source_time[3], stage="s2", latent_index=0)
            (o["image"].sum() + o["depth"].sum() + o["normal"].sum() + o["alpha"].sum()).backward()
        torch.cuda.synchronize()
        L.dimo_timing_enable(0)
        timing_iso = read_timing()
        iso_ms, iso_n = timing_iso["blend_bwd"]
        tr.renderer.gaussians.zero_grad()
        if pol is not None:
            pol.check()

    # ---- per-rank phases of a step and the replicas' agreement (every rank; gathered on rank 0).  Phase marks are events
    # on the caller's stream: head = weight packing + TimeNet forward (+ the KNN beside it) up to the forks; chains = the
    # motions' chains on their private streams up to the join (render, losses, rasterizer + skinning backward); tail =
    # TimeNet backward, collectives, optimizer.  The checksum is the parameter bucket's bit pattern.
    phases = None
    try:
        tr.marks = []
        for _ in range(6):
            tr.train_step()
        barrier()
        marks, tr.marks = tr.marks, None
        steps_m, cur = [], None
        for name, ev in marks:
            if name == "start":
                cur = {}
                steps_m.append(cur)
            if cur is not None:
                cur[name] = ev
        acc, cnt = {"head_ms": 0.0, "chains_ms": 0.0, "tail_ms": 0.0}, 0
        for k_, m_ in enumerate(steps_m[1:], 1):  # (the first marked step also absorbs the switch-over)
            if not all(x in m_ for x in ("start", "timenet_fwd", "raster_bwd+skinning_bwd", "allreduce+adam")):
                continue
            prev_end = steps_m[k_ - 1].get("allreduce+adam")
            acc["head_ms"] += (prev_end.elapsed_time(m_["timenet_fwd"]) if prev_end is not None
                               else m_["start"].elapsed_time(m_["timenet_fwd"]))
            acc["chains_ms"] += m_["timenet_fwd"].elapsed_time(m_["raster_bwd+skinning_bwd"])
            acc["tail_ms"] += m_["raster_bwd+skinning_bwd"].elapsed_time(m_["allreduce+adam"])
            cnt += 1
        phases = {k_: v_ / max(cnt, 1) for k_, v_ in acc.items()}
        phases["renders_per_step_this_rank"] = renders / max(args.steps, 1)
    except Exception as e:  # (diagnostics only)
        tr.marks = None
        phases = {"failed": repr(e)}
    fp = tr.renderer.gaussians.flat_params.detach()
    chk = fp.view(torch.int32).to(torch.int64)
    chk = torch.stack([chk.sum(), (chk * torch.arange(1, chk.numel() + 1, device=device, dtype=torch.int64) % 1000003).sum()])
    rank_phases, replicas_identical = [phases], True
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, phases)
        rank_phases = gathered
        cmax, cmin = chk.clone(), chk.clone()
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
        replicas_identical = bool(torch.equal(cmax, cmin))

    trained_stats = None
    if rank == 0 and world == 1:
        try:  # (what a render looks like to the blend kernels NOW: the sustained run below drifts away from it)
            trained_stats = scene_stats(tr)
        except Exception:
            trained_stats = None
    # sustained rate: the fresh-state figure above is steps ~26-45 of a process; a long run drifts (random targets blow
    # a few Gaussians up, R grows) and crosses the schedule's stage-s2 opacity prune (step % 1000 == 0: the Gaussian
    # count changes, every workspace is rebuilt).  >= 1000 consecutive steps, every rank, collectives included.
    sustained = None
    if args.sustained_steps > 0:
        n0 = tr.renderer.gaussians._xyz.shape[0]
        r0 = getattr(pol, "last_r_mean", None) if pol is not None else None
        step0, skipped0 = tr.step, tr.skipped_steps
        barrier()
        ts = time.perf_counter()
        ns = 0
        for _ in range(args.sustained_steps):
            ns += tr.train_step()
        barrier()
        dts = time.perf_counter() - ts
        st = torch.tensor([dts, float(ns)], dtype=torch.float64, device=device)
        if world > 1:
            smax = st.clone()
            dist.all_reduce(smax, op=dist.ReduceOp.MAX)
            dist.all_reduce(st, op=dist.ReduceOp.SUM)
            dts, ns = float(smax[0]), float(st[1])
        if pol is not None:
            pol.poll(lag=0)
        sustained = {"frames_per_s": ns / dts, "ms_per_step": 1e3 * dts / args.sustained_steps,
                     "steps": args.sustained_steps, "schedule_steps": [step0 + 1, tr.step],
                     "s2_prunes_crossed": sum(1 for q in range(step0 + 1, tr.step + 1)
                                              if q % tr.cfg.densification_interval_s2 == 0
                                              and q < tr.cfg.density_end_iter_s2),
                     "gaussians_start_end": [int(n0), int(tr.renderer.gaussians._xyz.shape[0])],
                     "R_mean_per_render_start_end": [r0, getattr(pol, "last_r_mean", None) if pol is not None else None],
                     "skipped_steps": tr.skipped_steps - skipped0,
                     "what": "consecutive training steps right after the measurements above, same process and "
                             "trainer, wall clock with a barrier + device sync on both sides, max over ranks"}
    sustained_teacher = None
    if args.sustained_steps > 0 and world == 1 and not args.no_teacher:
        try:
            sustained_teacher = teacher_sustained(device, args.num_pts, args.resolution, per_gpu, args.sustained_steps)
        except Exception as e:  # (never takes the headline down with it)
            sustained_teacher = {"frames_per_s": None, "what": f"failed: {e!r}"}
    if rank == 0:
        P = args.resolution * args.resolution
        sched_ms, sched_n = timing["blend_bwd"]            # in the timed schedule (overlapped launches per motion)
        sched_rpl = renders / max(sched_n, 1)
        bwd_ms, bwd_n = timing_serial["blend_bwd"]          # alone on the device: one launch over the step's renders
        # a launch moves the algorithmic bytes of all the renders it covers
        rpl = serial_renders / max(bwd_n, 1)
        rps_local = renders / max(args.steps, 1)  # renders this rank takes through one step
        win = 3 * rps_local                       # ... and through the three steps every kernel group was timed in
        rpl_of = lambda k: (win / timing_all[k][1]) if timing_all.get(k, (0, 0))[1] else 1.0
        if R_step:  # mean instance count per render of the last timed steps (the capacity policy's lagged read-back)
            R = int(R_step)
        alg_render = (28 + 4 * NFEAT) * R + (8 * (NFEAT + 1) + 8) * P + (24 + 4 * NFEAT) * V
        alg_bytes = alg_render * rpl
        avg_s = (bwd_ms / max(bwd_n, 1)) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic = None
        valu = None
        sqf = os.path.join(ROOT, "profiles", "sq_blend.json")
        if os.path.exists(sqf):  # SQ counters of a rocprofv3 --pmc run of this workload (tools/pmc_sq.sh), not live
            try:
                d = json.load(open(sqf)).get("blend_bwd_batched", {}).get("derived", {})
                valu = {"valu_busy_at_4_cycles_per_inst": d.get("valu_busy_quad"),
                        "valu_busy_at_2_cycles_per_inst": d.get("valu_issue_frac"),
                        "waves_per_simd": d.get("waves_per_simd"), "source": "profiles/sq_blend.json (rocprofv3 --pmc, "
                        "separate run of tools/pmc_probe.py)"}
            except Exception:
                valu = None
        traffic_source = None
        c3_default = (args.num_pts, args.resolution, args.per_gpu, args.global_batch) == (100000, 512, "2,2,2", 0)
        if world == 1 and c3_default and not args.no_live_pmc:
            live, traffic_source = live_pmc(regime=args.regime)
            if live is not None:
                traffic = live["hbm_bytes_per_launch"] * rpl / live["renders_per_launch"]
                valu = live.get("valu", valu)
        pmc = os.path.join(ROOT, "profiles", "pmc_blend_bwd.json")
        if traffic is None and os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get("hbm_bytes_per_launch")
                if traffic is not None and pj.get("renders_per_launch"):  # scale to this run's batch size
                    traffic = traffic * rpl / pj["renders_per_launch"]
            except Exception:
                traffic = None
        shape_name = {(100000, 512): "C3", (50000, 256): "C2 shape", (200000, 1024): "C5 shape"}.get(
            (args.num_pts, args.resolution), "custom")
        res = {
            "metric": f"train-step frames/sec @{args.num_pts // 1000}k Gaussians {args.resolution}^2 (renders through "
                      f"deform+raster fwd+bwd+losses+Adam)",
            "value": renders_total / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f32",
            # every rank adds 1 through the same all-reduce the elapsed time goes through
            "ranks_seen": ranks_seen,
            "backend": (("rccl (torch.distributed 'nccl')" if args.backend == "nccl" else args.backend)
                        if world > 1 else "none (one rank)"),
            "data": "synthetic (seeded Gaussians initialised as the reference does, random-init TimeNet, "
                    "random targets; no dataset/LPIPS weights offline; LPIPS/ARAP/GA/KL terms excluded)",
            "config": {"workload": f"{shape_name}: {args.num_pts} Gaussians, 512 control points, {args.resolution}^2, stage s2, "
                                   f"diff_gauss flavour (rgb+depth+normal+alpha), "
                                   + (f"a fixed step of {tr.cfg.motions_per_step} x {tr.cfg.views_per_step} x "
                                      f"{tr.cfg.frames_per_step} renders sharded over the ranks (batch_size "
                                      f"{args.global_batch})" if args.global_batch else
                                      f"{per_gpu[0] * per_gpu[1] * per_gpu[2]} renders/GPU/step ({per_gpu[0]} motions x "
                                      f"{per_gpu[1]} views x {per_gpu[2]} frames per GPU)")
                                   + ", schedule step 1000+ (every image term on)"
                                   + f", opacity regime '{args.regime}'",
                       "renders_per_step": int(renders_total / args.steps), "parallelism": f"dp{world}",
                       "R_tile_instances": R, "V_visible": V,
                       "instance_capacity": int(pol.capacity) if pol is not None else None,
                       "hbm_peak_allocated_GB": peak_mem / 1e9},
            # a step whose renders overflowed the instance capacity is skipped on the device (Adam no-op) but its
            # renders are still counted above: this must read 0 for `value` to be a training rate
            "setup_steps_before_warmup": PRESTEPS + SETTLE,
            "skipped_steps": {"timed_region": skipped_timed, "whole_run": skipped_total},
            "allreduce_exposed_ms_per_step": (sum(ar_ms) / len(ar_ms)) if ar_ms else (0.0 if world == 1 else None),
            # every rank's step by phase (events on its caller's stream, 5 steps) and whether the replicas' parameter
            # buckets agree bit for bit after everything above
            "rank_phases_ms": rank_phases, "replicas_bit_identical": replicas_identical,
            "roofline": {"bound": "valu", "frac_is_of": "hbm peak (as the metric asks)",
                         "kernel": "blend_bwd_batched_kernel<true, true> (the joint launch over the step's renders)",
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "renders_per_launch": rpl,
                         "avg_ms": bwd_ms / max(bwd_n, 1), "launches": bwd_n,
                         "clock": "live HIP events on the launch stream over %d steps taken right after the timed region "
                                  "with the rasterizer backward as ONE launch over the step's renders (the kernel alone "
                                  "on the device; rocprofv3 of the same steps agrees: profiles/)" % serial_steps,
                         "timed_region": {
                             "avg_ms": sched_ms / max(sched_n, 1), "launches": sched_n, "renders_per_launch": sched_rpl,
                             "achieved": (alg_render * sched_rpl / (sched_ms / max(sched_n, 1) * 1e-3) / 1e9)
                             if sched_ms else None,
                             "frac": (alg_render * sched_rpl / (sched_ms / max(sched_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS)
                             if sched_ms else None,
                             "what": "the same kernel inside the timed schedule: one launch per motion on the motion's "
                                     "own stream, overlapping the other motion's loss / binning kernels (the schedule "
                                     "is faster, the launch itself reads longer)"},
                         "traffic_source": (traffic_source if (traffic_source and not traffic_source.startswith("not"))
                                            else ((traffic_source + "; " if traffic_source else "") +
                                                  "profiles/pmc_blend_bwd.json: FETCH_SIZE / WRITE_SIZE of separate "
                                                  "rocprofv3 --pmc passes over tools/pmc_probe.py in an earlier run, "
                                                  "calibrated on a 1 GiB copy; scaled to this run's renders per launch"
                                                  if traffic is not None else traffic_source)),
                         "valu": valu,
                         "isolated": {"avg_ms": iso_ms / max(iso_n, 1), "launches": iso_n, "renders_per_launch": 1,
                                      "achieved": alg_render / (iso_ms / max(iso_n, 1) * 1e-3) / 1e9 if iso_ms else None,
                                      "frac": alg_render / (iso_ms / max(iso_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS
                                      if iso_ms else None,
                                      "what": "single-render kernel (blend_bwd_kernel, the C-ABI path) with nothing "
                                              "else on the device; in the timed region two batches overlap on two "
                                              "streams"},
                         "note": "tile blend is FP32-VALU bound, not HBM bound (each 64-B record is reused by 256 "
                                 "pixels); the HBM fraction is reported as required, see DESIGN.md"},
            "synced_step_ms": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "steps": len(lat),
                               "what": "one step at a time with a device sync after each (no overlap of the host "
                                       "enqueue with the previous step), outside the timed region"},
            # pixel-Gaussian interactions I = sum over tiles of len * 256, against the FP32 vector peak
            "interactions": {"per_render": R * 256, "blend_bwd_per_s": R * 256 / (avg_s / rpl) if avg_s > 0 else None,
                             "blend_fwd_per_s": (R * 256 / (timing_all["blend_fwd"][0] / timing_all["blend_fwd"][1]
                                                            * 1e-3 / rpl_of("blend_fwd"))
                                                 if timing_all["blend_fwd"][1] else None),
                             "fp32_vector_peak_flops": FP32_VALU_PEAK,
                             "note": "list entries x 256 pixels; culling and saturation skip most of them, so the "
                                     "rate is an upper-bound style figure, not executed FLOPs"},
            # achieved HBM GB/s per kernel against the 8 TB/s peak, from the ALGORITHMIC bytes of
            # DESIGN.md section 4 and (a) the BATCHED launches the step really runs, (b) single-render launches
            # measured alone on the device
            "kernel_rooflines": kernel_rooflines(timing_all, args.num_pts, V, R, P, renders_in_window=win,
                                                 views_per_group=tr_views),
            "kernel_rooflines_isolated": kernel_rooflines(timing_iso, args.num_pts, V, R, P),
            # BASELINE.json metric (ii): rasterizer forward / backward device time of ONE render alone on the device
            # (sum of its kernels' HIP-event times: project, scan, depth sort, placement, blend | blend, projection)
            "raster_ms_per_render_isolated": {
                "forward": sum(per_launch(timing_iso, k) for k in
                               ("preprocess_fwd", "scan", "sort", "emit", "ranges", "place", "blend_fwd")),
                "backward": sum(per_launch(timing_iso, k) for k in ("blend_bwd", "preprocess_bwd"))},
            "raster_ms_per_render_batched": {
                "forward": sum(per_launch(timing_all, k) / max(rpl_of(k), 1) for k in
                               ("preprocess_fwd", "scan", "sort", "emit", "ranges", "place", "blend_fwd")),
                "backward": sum(per_launch(timing_all, k) / max(rpl_of(k), 1) for k in ("blend_bwd", "preprocess_bwd")),
                "what": "per render of a batched launch in the timed schedule (two motions' batches share the chip)"},
            "sustained": sustained,
            "sustained_teacher": sustained_teacher,
            "kernels_ms_per_launch": {k: (v[0] / v[1] if v[1] else None) for k, v in timing_all.items()},
            "kernels_ms_per_launch_isolated": {k: (v[0] / v[1] if v[1] else None) for k, v in timing_iso.items()},
        }
        # ---- the protocol's other clocks as first-class fields (round-5 review item 6).  `value` = K steps enqueued back
        # to back between two device syncs (the driver's contract); BASELINE.md section 2's protocol syncs after EVERY
        # step and quotes the median; `sustained` is >= 1000 consecutive steps of the same trainer.
        res["synced_frames_per_s"] = rps_local * world / (pct(0.5) * 1e-3)
        res["sustained_frames_per_s"] = sustained["frames_per_s"] if sustained else None
        res["value_behind_idle_gap"] = gap_renders * world / t_gap
        res["bench_protocol"] = (
            f"{PRESTEPS} set-up steps, GC pass + freeze, {SETTLE} settle steps (the last {SETTLE - 5} of them timed: "
            f"value_behind_idle_gap = the rate the pre-round-5 protocol read, right behind the collector's idle gap), "
            f"W = {args.warmup} warm-up steps, barrier + device sync, K = {args.steps} timed steps enqueued back to back "
            f"with NO per-step sync, barrier + device sync -> value; synced_frames_per_s = BASELINE.md section 2's "
            f"protocol (one device sync per step, median of {len(lat)} steps); sustained_frames_per_s = "
            f"{args.sustained_steps} consecutive steps on the benchmark's noise targets")
        res["config"]["protocol"] = {"value_frames_per_s": res["value"], "synced_frames_per_s": res["synced_frames_per_s"],
                                     "sustained_frames_per_s": res["sustained_frames_per_s"],
                                     "value_behind_idle_gap": res["value_behind_idle_gap"],
                                     "setup_steps_before_warmup": PRESTEPS + SETTLE, "per_step_sync_in_timed_region": False}
        # ---- both synthetic regimes of SURVEY 8d from ONE process (round-5 review item 1): the headline's ("trained":
        # lists saturate after ~85 of ~950 entries) from the measurements above, the reference's own initial state
        # ("init": every opacity 0.05, renderer/latent_gs_renderer.py:431 -- every pixel walks its whole list) from a
        # second trainer built here
        if world == 1 and c3_default and not args.no_regimes and args.regime == "trained":
            try:
                st = trained_stats if trained_stats is not None else scene_stats(tr)
                fm, fn = timing_all["blend_fwd"]
                regimes = {"trained": {
                    "frames_per_s": res["value"], "ms_per_step": res["ms_per_step"], "steps": args.steps,
                    "skipped_steps": skipped_timed, "V_visible": V, **st,
                    "blend_bwd_ms_per_launch": bwd_ms / max(bwd_n, 1), "blend_bwd_renders_per_launch": rpl,
                    "blend_fwd_ms_per_launch": fm / max(fn, 1), "blend_fwd_renders_per_launch": rpl_of("blend_fwd"),
                    "roofline": {"kernel": "blend_bwd_batched_kernel<true, true>", "algorithmic_bytes_per_launch": alg_bytes,
                                 "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "traffic": traffic, "valu": valu,
                                 "pixel_entry_pairs_per_s": st["pixel_entry_pairs"] * rpl / avg_s if avg_s > 0 else None}}}
                del tr
                tr = None
                torch.cuda.empty_cache()
                regimes["init"] = regime_figures(device, args.num_pts, args.resolution, per_gpu, "init",
                                                 max(args.steps, 20), pmc=not args.no_live_pmc)
                regimes["what"] = ("the same C3 step in SURVEY 8d's two synthetic regimes, one process: 'trained' = "
                                   "opacity sigmoid(U(-2, 4)) (the headline `value`), 'init' = every opacity 0.05 = the state "
                                   "the reference creates its Gaussians in and starts stage s2 from "
                                   "(renderer/latent_gs_renderer.py:431,1038-1058); real training walks from the second to "
                                   "the first")
                res["regimes"] = regimes
                brief = lambda d: {k: d[k] for k in ("frames_per_s", "ms_per_step", "n_contrib_mean", "R_tile_instances",
                                                     "blend_fwd_ms_per_launch", "blend_bwd_ms_per_launch",
                                                     "checkpoint_bytes_written", "instance_capacity")} \
                    | {"roofline_frac": d["roofline"]["frac"], "valu": d["roofline"].get("valu")}
                res["config"]["regimes"] = {k: brief(regimes[k]) for k in ("trained", "init")}
                res["init_regime_frames_per_s"] = regimes["init"]["frames_per_s"]
            except Exception as e:  # (never takes the headline down with it)
                res["regimes"] = {"what": f"failed: {e!r}"}
        if world == 1 and not args.no_dropin:
            # what a maintainer gets who only swaps the imports and keeps the reference's
            # trainer: tests/reference_step.py restates GUI.train_step's loop body in its order of operations
            try:
                tr = None
                torch.cuda.empty_cache()
                res.update(dropin_figures(device, args.num_pts, args.resolution, per_gpu))
            except Exception as e:
                res["dropin_frames_per_s"] = None
                res["dropin_what"] = f"failed: {e!r}"
        if world == 1 and not args.no_dropin:
            # stage s1 (2 800 of the reference's 12 800 iterations, run_train_latent.sh:12): num_cpts Gaussians (what
            # FPS leaves), the TimeNet evaluated on them, every scale = exp(_r) -- same step shape, HIP pipeline
            try:
                from dimo_amd.rasterizer import CapacityPolicy
                from dimo_amd.renderer import Renderer
                from dimo_amd.synth import init_synthetic_model
                from dimo_amd.trainer import TrainConfig, Trainer
                c1 = TrainConfig(num_pts=512, resolution=args.resolution, motions_per_step=per_gpu[0],
                                 views_per_step=per_gpu[1], frames_per_step=per_gpu[2], stage="s1", FPS_iter=10 ** 9,
                                 position_lr_max_steps=500)
                rd1 = Renderer(sh_degree=0, white_background=True, radius=c1.radius, num_latent_code=c1.num_motions,
                               add_normal=True, device=device, capacity=CapacityPolicy(initial=1 << 22))
                init_synthetic_model(rd1, c1.num_pts, c1.num_cpts, seed=0, regime="trained", num_latent=c1.num_motions)
                g1 = rd1.gaussians
                g1._r = torch.nn.Parameter(torch.full((1, 1), -3.2, device=device))  # exp(-3.2) = 0.04: blobs of a 512-point shape
                t1 = Trainer(c1, rd1)
                t1.step = 1100  # full resolution, past the density window
                for _ in range(5):
                    t1.train_step()
                torch.cuda.synchronize()
                k1 = 20
                ts = time.perf_counter()
                n1 = sum(t1.train_step() for _ in range(k1))
                torch.cuda.synchronize()
                res["s1_frames_per_s"] = n1 / (time.perf_counter() - ts)
                res["s1_what"] = ("stage s1 on the HIP pipeline (direct=%s): 512 Gaussians of radius exp(_r) = 0.04, TimeNet "
                                  "on the Gaussians, %d renders/step at %d^2, %d steps; skipped %d"
                                  % (t1.direct, n1 // k1, args.resolution, k1, t1.skipped_steps))
                del t1, rd1
            except Exception as e:
                res["s1_frames_per_s"] = None
                res["s1_what"] = f"failed: {e!r}"
        if world == 1 and not args.no_dropin:
            # the reference's own benchmark-like harness (main_test_dimo.py:872-894): render-only calls per second
            try:
                res["render_fps"] = render_fps(device, args.num_pts, args.resolution)
            except Exception as e:
                res["render_fps"] = {"what": f"failed: {e!r}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args.num_pts, args.resolution)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
            try:  # the C1 plumbing configuration on the same host cores
                res["cpu_baseline_c1"] = cpu_baseline_c1()
            except Exception as e:
                res["cpu_baseline_c1"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                          "sample": f"failed: {e!r}"}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
