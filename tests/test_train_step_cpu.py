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
