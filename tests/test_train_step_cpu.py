"""Config 1 of BASELINE.json (plumbing, no GPU): the product's host logic (Renderer, Trainer, flat
parameter bucket, sampling, losses, Adam) driven end to end on CPU with the oracle rasterizer injected;
plus the data-parallel path on 2 gloo ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dimo_amd.trainer import TrainConfig, enumerate_triples, shard
from tests.cpu_backend import make_cpu_trainer


def small_cfg(**kw):
    base = dict(num_pts=1000, num_cpts=32, num_motions=3, num_frames=5, num_views=4, motions_per_step=1,
                views_per_step=1, frames_per_step=1, resolution=128)
    base.update(kw)
    return TrainConfig(**base)


def test_config1_one_render_step():
    tr = make_cpu_trainer(small_cfg())
    g = tr.renderer.gaussians
    before = g.flat_params.clone()
    assert g._xyz.data_ptr() == g.flat_params.data_ptr()  # parameters are views of the flat bucket
    assert g._xyz.grad.data_ptr() == g.flat_grads.data_ptr()
    n = tr.train_step()
    assert n == 1 and torch.isfinite(tr.last_loss)
    assert torch.isfinite(g.flat_params).all() and not torch.equal(before, g.flat_params)
    assert torch.count_nonzero(g.flat_grads) == 0  # reset by one memset
    assert g.neighbor_indices.shape == (1000, 4) and g.neighbor_indices.dtype == torch.int64
    names = [pg["name"] for pg in tr.optimizer.param_groups]
    assert names == ["xyz", "f_dc", "opacity", "scaling", "rotation", "latent_code", "deform", "deform_rot",
                     "c_xyz", "c_radius"]  # f_rest / r are empty at sh_degree 0 in stage s2
    assert tr.optimizer.defaults["eps"] == 1e-15


def test_loss_is_partition_invariant_and_sum_reduced():
    """Splitting a step's triples over ranks and summing gradients == the single-process step."""
    cfg = small_cfg(motions_per_step=2, views_per_step=2, frames_per_step=1, resolution=64, num_pts=400)
    tr = make_cpu_trainer(cfg)
    triples = tr.sample()
    assert len(triples) == 4 and triples == enumerate_triples(sorted({t[0] for t in triples}, key=[t[0] for t in triples].index),
                                                              list(dict.fromkeys(t[1] for t in triples)),
                                                              list(dict.fromkeys(t[2] for t in triples)))
    grads = []
    for world in (1, 2):
        acc = None
        for rank in range(world):
            t2 = make_cpu_trainer(cfg, rank=rank, world=world)
            t2.world = world
            t2.all_reduce_grads = lambda: None  # no process group here: sum by hand
            g = t2.renderer.gaussians
            t2.optimizer.step = lambda: None
            g.zero_grad = lambda: None
            t2.train_step(triples)
            acc = g.flat_grads.clone() if acc is None else acc + g.flat_grads
        grads.append(acc)
    rel = (grads[0] - grads[1]).abs().sum() / grads[0].abs().sum()
    assert rel < 1e-5, rel


def test_per_motion_terms_are_counted_once_when_a_motion_is_split_over_ranks():
    """KL (VAE latents) and the geometry-anchor term with ONE motion's four renders split over two ranks: the SUM of
    the ranks' gradients must equal the single-process gradients (the KL term used to be added once per rank)."""
    cfg = small_cfg(motions_per_step=1, views_per_step=2, frames_per_step=2, resolution=48, num_pts=300,
                    vae_latent=True, add_ga=True)
    triples = make_cpu_trainer(cfg).sample()
    assert len({t[0] for t in triples}) == 1 and len(triples) == 4
    grads, losses = [], []
    for world in (1, 2):
        acc, loss = None, 0.0
        for rank in range(world):
            t2 = make_cpu_trainer(cfg, rank=rank, world=world)
            t2.all_reduce_grads = lambda: None
            g = t2.renderer.gaussians
            t2.optimizer.step = lambda: None
            g.zero_grad = lambda: None
            with torch.no_grad():  # std = exp(-15): the per-render eps draws (global generator) do not matter
                g._log_var.fill_(-30.0)
            t2.cache_cpts_s1()
            t2.cpts_s1 += 0.02 * torch.randn(t2.cpts_s1.shape, generator=torch.Generator().manual_seed(3))
            t2.step = 300
            t2.train_step(triples)
            acc = g.flat_grads.clone() if acc is None else acc + g.flat_grads
            loss += float(t2.last_loss)
        grads.append(acc)
        losses.append(loss)
    assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[0]), losses
    rel = (grads[0] - grads[1]).abs().sum() / grads[0].abs().sum()
    assert rel < 1e-5, rel


def test_shard_covers_everything():
    items = list(range(10))
    for world in (1, 2, 3, 4, 8, 16):
        got = [x for r in range(world) for x in shard(items, r, world)]
        assert got == items


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = small_cfg(motions_per_step=2, views_per_step=2, frames_per_step=1, resolution=64, num_pts=400)
    tr = make_cpu_trainer(cfg, rank=rank, world=world)
    counts = [tr.train_step() for _ in range(2)]
    torch.save(dict(params=tr.renderer.gaussians.flat_params.clone(), counts=counts), f"{out}/rank{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_data_parallel_two_ranks_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(f"{tmp_path}/rank0.pt"), torch.load(f"{tmp_path}/rank1.pt")
    assert a["counts"] == [2, 2] and b["counts"] == [2, 2]
    assert torch.equal(a["params"], b["params"]), "replicas diverged"
    cfg = small_cfg(motions_per_step=2, views_per_step=2, frames_per_step=1, resolution=64, num_pts=400)
    single = make_cpu_trainer(cfg)
    for _ in range(2):
        single.train_step()
    p = single.renderer.gaussians.flat_params
    rel = (p - a["params"]).abs().sum() / p.abs().sum()
    assert rel < 1e-4, rel


def _s1_cfg():
    # stage s1: FPS down to num_cpts at step 0, statistics from step 1 on, densify at steps 2 and 4
    return small_cfg(stage="s1", num_pts=300, num_cpts=48, motions_per_step=2, views_per_step=1, frames_per_step=1,
                     resolution=48, density_start_iter=1, densification_interval=2, densify_grad_threshold=1e-9,
                     position_lr_max_steps=500)


def _s1_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tr = make_cpu_trainer(_s1_cfg(), rank=rank, world=world, regime="init")
    g = tr.renderer.gaussians
    torch.manual_seed(100 + rank)  # the ranks' global generators differ: the split draws must not depend on them
    sizes = []
    for _ in range(5):
        tr.train_step()
        sizes.append(g._xyz.shape[0])
    torch.save(dict(params=g.flat_params.clone(), sizes=sizes, accum=g.xyz_gradient_accum.clone(),
                    denom=g.denom.clone(), radii=g.max_radii2D.clone()), f"{out}/rank{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_stage_s1_data_parallel_densification_keeps_replicas_identical(tmp_path):
    """SURVEY 8e / renderer/latent_gs_renderer.py:838,922-924: under data parallelism the densification statistics
    are all-reduced (SUM for the gradient norms and counts, MAX for the radii) and the split draws come from a
    rank-identical generator, so two gloo ranks train stage s1 ACROSS two densify_and_prune calls with equal Gaussian
    counts and bit-identical parameters -- and the counts equal the single-process run on the same triples."""
    port = 25500 + (os.getpid() % 2000)
    mp.spawn(_s1_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(f"{tmp_path}/rank0.pt"), torch.load(f"{tmp_path}/rank1.pt")
    assert a["sizes"] == b["sizes"], (a["sizes"], b["sizes"])
    assert a["sizes"][0] == 48 and a["sizes"][-1] != 48, a["sizes"]  # FPS at step 0, then densified
    assert torch.equal(a["params"], b["params"]), "replicas diverged"
    for k in ("accum", "denom", "radii"):
        assert torch.equal(a[k], b[k]), k
    single = make_cpu_trainer(_s1_cfg(), regime="init")
    sizes = []
    for _ in range(5):
        single.train_step()
        sizes.append(single.renderer.gaussians._xyz.shape[0])
    assert sizes[:2] == a["sizes"][:2]  # identical up to the first densification (the split draws differ after it)


def test_stage_s1_fps_and_lr_rules():
    tr = make_cpu_trainer(_s1_cfg(), regime="init")
    g = tr.renderer.gaussians
    lrs = {grp["name"]: grp["lr"] for grp in tr.optimizer.param_groups}
    assert lrs["c_xyz"] == 0.0 and lrs["c_radius"] == 0.0 and "r" in lrs  # prepare_train_s1
    xyz0 = g._xyz.detach().clone()
    from oracle.regularizers_ref import farthest_point_sample_ref
    idx = farthest_point_sample_ref(xyz0.numpy(), 48)
    tr.fps(48)
    # the reference's `prune_points(idxs)` keeps rows N-1-idx (bitwise not of an index tensor), in sampling order
    assert torch.equal(g._xyz.detach(), xyz0[300 - 1 - torch.from_numpy(idx)])


def test_stage_s2_schedule_rules():
    """main_train_dimo.py:250-253 (xyz lr pinned to 2e-4 while step < 1000 in s2) and :362,368 (the depth / normal
    smoothness terms start after step 200)."""
    cfg = small_cfg(motions_per_step=1, resolution=32, num_pts=200)
    tr = make_cpu_trainer(cfg)
    tr.train_step()
    lr = {grp["name"]: grp["lr"] for grp in tr.optimizer.param_groups}
    assert lr["xyz"] == 0.0002 and tr._reg_on() == (False, False)
    tr.step = 200
    tr.train_step()
    assert tr._reg_on() == (True, True)
    tr.step = 999
    tr.train_step()
    lr = {grp["name"]: grp["lr"] for grp in tr.optimizer.param_groups}
    assert lr["xyz"] != 0.0002 and abs(lr["xyz"] - tr.renderer.gaussians.xyz_scheduler_args(1000)) < 1e-12


def test_progressive_render_resolution_rule():
    """main_train_dimo.py:261,307-313: 128^2 while step < 300, 256^2 while step < 450, then full size; the targets are
    resampled bilinearly to the render size."""
    cfg = small_cfg(motions_per_step=1, resolution=512, num_pts=50)
    tr = make_cpu_trainer(cfg)
    got = []
    for step in (1, 299, 300, 449, 450, 5000):
        tr.step = step
        got.append(tr.render_resolution())
    assert got == [128, 128, 256, 256, 512, 512]
    tr.step = 10
    img, mask = tr.target(0, 1, 2)
    full, fmask = tr.targets.get(0, 1, 2)
    assert img.shape == (3, 128, 128) and mask.shape == (1, 128, 128) and full.shape == (3, 512, 512)
    want = torch.nn.functional.interpolate(full[None], (128, 128), mode="bilinear", align_corners=False)[0]
    assert torch.equal(img, want)
    cfg2 = small_cfg(motions_per_step=1, resolution=96, num_pts=50)
    assert make_cpu_trainer(cfg2).render_resolution() == 96  # never above the configured size


# ---------------------------------------------------------------------------- world sizes 3 / 4 / 8 (round 4)
def _dp_cfg(kind):
    if kind == "uneven":     # 4 renders: world 3 -> shards of 1, 1, 2; world 8 -> four ranks with an EMPTY shard
        return small_cfg(motions_per_step=2, views_per_step=2, frames_per_step=1, resolution=48, num_pts=300)
    if kind == "strong_b2":  # the reference's batch_size 2 step: 4 motions x 2 views x 2 frames = 16 renders
        return small_cfg(num_motions=4, motions_per_step=4, views_per_step=2, frames_per_step=2, resolution=32,
                         num_pts=200, num_cpts=16)
    raise ValueError(kind)


def _dp_worker_n(rank, world, port, out, kind, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    tr = make_cpu_trainer(_dp_cfg(kind), rank=rank, world=world)
    tr.step = 250  # every image term on
    counts = [tr.train_step() for _ in range(steps)]
    torch.save(dict(params=tr.renderer.gaussians.flat_params.clone(), counts=counts), f"{out}/rank{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,kind,steps", [(3, "uneven", 2), (8, "uneven", 2), (4, "strong_b2", 1), (8, "strong_b2", 1)])
def test_data_parallel_world_sizes_uneven_and_empty_shards(tmp_path, world, kind, steps):
    """SURVEY section 4 asks for 2-8 ranks: uneven shards (world 3), ranks with NO render at all (4 renders on 8
    ranks: they still take part in every collective and apply the same update) and the strong-scaling shape of
    DESIGN section 6 (batch_size 2 = 16 renders: 4 / 2 per rank at world 4 / 8).  Replicas bit-identical, render
    counts = the contiguous shard sizes, parameters equal to the single-process step."""
    port = 21500 + (os.getpid() * 7 + world * 131 + len(kind)) % 6000
    mp.spawn(_dp_worker_n, args=(world, port, str(tmp_path), kind, steps), nprocs=world, join=True)
    outs = [torch.load(f"{tmp_path}/rank{r}.pt") for r in range(world)]
    n = len(enumerate_triples(range(_dp_cfg(kind).motions_per_step), range(_dp_cfg(kind).views_per_step),
                              range(_dp_cfg(kind).frames_per_step)))
    want = [len(shard(list(range(n)), r, world)) for r in range(world)]
    assert [o["counts"][0] for o in outs] == want and sum(want) == n
    if kind == "uneven" and world == 8:
        assert want.count(0) == 4
    for o in outs[1:]:
        assert torch.equal(outs[0]["params"], o["params"]), "replicas diverged"
    single = make_cpu_trainer(_dp_cfg(kind))
    single.step = 250
    for _ in range(steps):
        single.train_step()
    p = single.renderer.gaussians.flat_params
    rel = (p - outs[0]["params"]).abs().sum() / p.abs().sum()
    assert rel < 1e-4, rel


def _s1_worker_n(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    tr = make_cpu_trainer(_s1_cfg(), rank=rank, world=world, regime="init")
    g = tr.renderer.gaussians
    torch.manual_seed(100 + rank)
    sizes, counts = [], []
    for _ in range(4):
        counts.append(tr.train_step())
        sizes.append(g._xyz.shape[0])
    torch.save(dict(params=g.flat_params.clone(), sizes=sizes, counts=counts, accum=g.xyz_gradient_accum.clone(),
                    denom=g.denom.clone(), radii=g.max_radii2D.clone()), f"{out}/rank{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_stage_s1_densification_at_world_4_with_idle_ranks(tmp_path):
    """Two renders per step on FOUR ranks: two ranks render nothing, the rank that owns the step's LAST triple feeds the
    densification statistics (main_train_dimo.py:429-431), every rank densifies identically."""
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_s1_worker_n, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    outs = [torch.load(f"{tmp_path}/rank{r}.pt") for r in range(4)]
    assert [o["counts"][0] for o in outs] == [0, 1, 0, 1]
    assert outs[0]["sizes"][0] == 48 and outs[0]["sizes"][-1] != 48, outs[0]["sizes"]
    for o in outs[1:]:
        assert o["sizes"] == outs[0]["sizes"] and torch.equal(o["params"], outs[0]["params"]), "replicas diverged"
        for k in ("accum", "denom", "radii"):
            assert torch.equal(o[k], outs[0][k]), k


def test_two_stage_schedule_hand_over():
    """`Trainer.train_dynamic` = GUI.train_dynamic's shape (main_train_dimo.py:170-218): stage s1 with FPS and the
    density window, `prune_s1_end`, `prepare_train_s2` (:471-500: Gaussians -> control points, shared radius -> control
    radii, `num_pts_per_cpt` fresh Gaussians per control point, a new optimizer without the `r` group, position lr
    2e-4 -> 2e-6 over iters_s2), stage s2."""
    mk = lambda: small_cfg(stage="s1", num_pts=60, num_cpts=24, num_pts_per_cpt=5, motions_per_step=1, views_per_step=1,
                           frames_per_step=1, resolution=32, FPS_iter=3, density_start_iter=1, density_end_iter=2,
                           densification_interval=2, densify_grad_threshold=1e-9, position_lr_max_steps=500)
    tr = make_cpu_trainer(mk(), regime="trained")
    g = tr.renderer.gaussians
    seen = []
    tr.train_dynamic(4, 2, on_step=lambda t: seen.append((t.stage, t.step, t.renderer.gaussians._xyz.shape[0])))
    assert [s[0] for s in seen] == ["s1"] * 4 + ["s2"] * 2 and [s[1] for s in seen] == [1, 2, 3, 4, 1, 2]
    assert seen[0][2] == 24 and seen[3][2] == 24          # FPS at steps 0 and 3 (densified in between)
    m = g._c_xyz.shape[0]
    assert m <= 24 and g._xyz.shape[0] == m * 5 and seen[-1][2] == m * 5
    assert tr.stage == "s2" and len(g._r) == 0
    names = [grp["name"] for grp in tr.optimizer.param_groups]
    assert "r" not in names and names[0] == "xyz"
    assert g._xyz.data_ptr() == g.flat_params.data_ptr()
    lr = {grp["name"]: grp["lr"] for grp in tr.optimizer.param_groups}
    assert lr["xyz"] == 0.0002                                # main_train_dimo.py:250-253
    assert abs(g.xyz_scheduler_args(2) - 0.000002) < 1e-12   # ... and the schedule ends at 2e-6 after iters_s2 = 2
    assert torch.isfinite(g.flat_params).all() and torch.isfinite(tr.last_loss)
    # right after the hand-over (no s2 step yet): the control points are where the stage-s1 Gaussians were, every
    # control radius is the shared exp(_r), and the fresh Gaussians sit within that radius of their control point
    tr2 = make_cpu_trainer(mk(), regime="trained")  # (the hand-over switches its config to stage s2: a fresh one)
    g2 = tr2.renderer.gaussians
    for _ in range(4):
        tr2.train_step()
    s1_xyz, s1_r = g2._xyz.detach().clone(), float(g2._r.detach().reshape(-1)[0])
    tr2.train_dynamic(0, 0)  # (already past stage s1's steps: prune_s1_end is skipped, the hand-over runs)
    assert torch.equal(g2._c_xyz.detach(), s1_xyz) and torch.all(g2._c_radius.detach() == s1_r)
    d = (g2._xyz.detach()[:, None, :] - g2._c_xyz.detach()[None]).norm(dim=-1).min(dim=1).values
    assert float(d.max()) <= float(np.exp(s1_r)) * (1 + 1e-5)


def test_prune_s1_end_prunes_gaussians_and_control_points_together():
    cfg = small_cfg(stage="s1", num_pts=40, num_cpts=40, resolution=32, FPS_iter=10 ** 9)
    tr = make_cpu_trainer(cfg, regime="trained")
    g = tr.renderer.gaussians
    tr.train_step()
    with torch.no_grad():
        g._opacity[::4] = -10.0  # sigmoid < 0.01
    keep = (torch.sigmoid(g._opacity.detach()) >= 0.01).squeeze(-1)
    xyz, cxyz = g._xyz.detach()[keep].clone(), g._c_xyz.detach()[keep].clone()
    m_c = g.optimizer.state[g._c_xyz]["exp_avg"][keep].clone() if g._c_xyz in g.optimizer.state else None
    g.prune_s1_end(min_opacity=0.01, extent=4, max_screen_size=1)
    assert g._xyz.shape[0] == 30 and g._c_xyz.shape[0] == 30 and g._c_radius.shape[0] == 30
    assert torch.equal(g._xyz.detach(), xyz) and torch.equal(g._c_xyz.detach(), cxyz)
    if m_c is not None:
        assert torch.equal(g.optimizer.state[g._c_xyz]["exp_avg"], m_c)
