"""GPU parity: fused skinning kernel (dimo_deform_* through the C ABI) vs oracle/deform_ref.py on
seeded inputs, and vs the reference-generated fixture; full Renderer.render on the GPU vs the CPU
pipeline with oracle kernels."""
import numpy as np
import pytest
import torch

from oracle.deform_ref import skinning_ref
from tests.test_deform_oracle import fixture_inputs

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rand_inputs(N, M, seed):
    g = torch.Generator().manual_seed(seed)
    c_xyz = (torch.rand(M, 3, generator=g) - 0.5)
    xyz = (torch.rand(N, 3, generator=g) - 0.5)
    d = torch.cdist(xyz, c_xyz)
    nn_dist, nn_idx = torch.topk(d, 4, dim=1, largest=False)
    return dict(xyz=xyz, rotation=torch.randn(N, 4, generator=g), scaling=torch.randn(N, 3, generator=g) - 3,
                opacity=torch.randn(N, 1, generator=g), c_xyz=c_xyz,
                c_log_radius=torch.log(torch.rand(M, 1, generator=g) * 0.2 + 0.05),
                d_xyz=torch.randn(M, 3, generator=g) * 0.05,
                d_rot=torch.tensor([1.0, 0, 0, 0]) + 0.3 * torch.randn(M, 4, generator=g),
                nn_dist=nn_dist.contiguous(), nn_idx=nn_idx.contiguous())


def _compare(a, local_frame=True, seed=0):
    from dimo_amd.deform import fused_skinning
    names = ("xyz", "rotation", "scaling", "opacity", "c_xyz", "c_log_radius", "d_xyz", "d_rot")
    cpu = {k: (v.clone().double().requires_grad_(True) if k in names else v) for k, v in a.items()}
    cpu["nn_dist"] = a["nn_dist"].double()
    gpu = {k: (v.clone().cuda().requires_grad_(True) if k in names else v.cuda()) for k, v in a.items()}
    ref = skinning_ref(**cpu, local_frame=local_frame)
    got = fused_skinning(*[gpu[k] for k in names], gpu["nn_dist"], gpu["nn_idx"], local_frame)
    g = torch.Generator().manual_seed(seed + 1)
    loss_r = loss_g = 0
    for r, o in zip(ref, got):
        err = (o.detach().cpu().double() - r.detach()).abs().mean().item()
        assert err <= TOL * max(1.0, r.detach().abs().mean().item()), err
        w = torch.randn(r.shape, generator=g)
        loss_r = loss_r + (r * w.double()).sum()
        loss_g = loss_g + (o * w.cuda()).sum()
    loss_r.backward()
    loss_g.backward()
    for k in names:
        r, o = cpu[k].grad, gpu[k].grad.cpu().double()
        if r is None:  # input unused by this variant (c_xyz when local_frame=False): the kernel writes zeros
            assert o.abs().max() == 0, k
            continue
        rel = (o - r).abs().sum() / (r.abs().sum() + 1e-12)
        assert rel <= TOL, (k, rel.item())


@pytest.mark.parametrize("N,M", [(1000, 32), (100000, 512), (777, 5), (5000, 1500)])
def test_fused_skinning_vs_oracle(N, M):
    _compare(_rand_inputs(N, M, seed=N + M))


def test_fused_skinning_global_frame_variant():
    _compare(_rand_inputs(3000, 64, seed=3), local_frame=False)


def test_fused_skinning_vs_reference_fixture():
    from dimo_amd.deform import fused_skinning
    z, a = fixture_inputs("deform_latent.npz")
    names = ("xyz", "rotation", "scaling", "opacity", "c_xyz", "c_log_radius", "d_xyz", "d_rot")
    pts, rot, scales, opac = fused_skinning(*[a[k].cuda() for k in names], a["nn_dist"].cuda(), a["nn_idx"].cuda())
    np.testing.assert_allclose(pts.cpu().numpy(), z["s2.in.means3D"], atol=5e-6)
    np.testing.assert_allclose(rot.cpu().numpy(), z["s2.in.rotations"], atol=5e-6)
    np.testing.assert_allclose(scales.cpu().numpy(), z["s2.in.scales"], rtol=2e-6)
    np.testing.assert_allclose(opac.cpu().numpy(), z["s2.in.opacities"], atol=2e-7)


def test_train_step_gpu_matches_cpu_oracle_pipeline():
    """One full stage-s2 train step (batched TimeNet, fused skinning, HIP rasterizer, fused SSIM, Adam) on
    the GPU vs the same host logic on CPU with every native op replaced by its oracle."""
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    from tests.cpu_backend import make_cpu_trainer
    cfg = TrainConfig(num_pts=3000, num_cpts=48, num_motions=4, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=2, resolution=96)
    cpu = make_cpu_trainer(cfg)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda")
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=cfg.seed, num_latent=cfg.num_motions)
    gpu = Trainer(cfg, rd)
    cpu.step = gpu.step = 300  # past depth/normal_reg_start_iter: every image term is on
    p0 = cpu.renderer.gaussians.flat_params.clone()
    assert torch.allclose(p0, rd.gaussians.flat_params.cpu(), atol=2e-6)  # log(sqrt(dist2)) rounds per device
    # gradients of the first step (before Adam's sign-like normalisation amplifies rounding)
    for t in (cpu, gpu):
        t.optimizer.step = lambda *a, **k: None
        t.renderer.gaussians.zero_grad = lambda: None
    triples = cpu.sample()
    cpu.train_step(triples)
    gpu.train_step(triples)
    gc, gg = cpu.renderer.gaussians.flat_grads, rd.gaussians.flat_grads.cpu()
    assert abs(cpu.last_loss.item() - gpu.last_loss.item()) <= 1e-4 * abs(cpu.last_loss.item())
    rel = (gc - gg).abs().sum() / gc.abs().sum()
    assert rel < 2e-4, rel


@pytest.mark.parametrize("regime", ["trained", "init"])
def test_train_step_at_benchmark_size_default_switches_matches_cpu_oracle_pipeline(regime):
    """The same comparison at C3 size (100 k Gaussians, 512 control points, 512^2, 8 renders) with every switch at
    its default -- in the trained regime and in the reference's own initial state (every opacity 0.05: SURVEY 8d's
    "init" regime, 7x the compositing work per render) --: the executor tests hold the KERNELS to the oracle at this size, this one holds the cross-stream
    SCHEDULE (two motions' chains on two private streams, skinning backward in order, fold + Adam head next to the
    TimeNet backward) -- a missed dependency shows as a wrong gradient bucket here, not as a rare flake."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    from tests.cpu_backend import make_cpu_trainer
    cfg = TrainConfig(num_pts=100000, num_cpts=512, num_motions=6, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=2, resolution=512, progressive_resolution=False)
    cpu = make_cpu_trainer(cfg, regime=regime)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                  capacity=CapacityPolicy(initial=1 << 22))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=cfg.seed, num_latent=cfg.num_motions, regime=regime)
    gpu = Trainer(cfg, rd)
    assert gpu.direct
    cpu.step = gpu.step = 300
    # the trainers sort their Gaussians (Morton order) from positions that differ in the last bit between the devices
    # (log(sqrt(dist2))): start both from the CPU model's parameters
    with torch.no_grad():
        rd.gaussians.flat_params.copy_(cpu.renderer.gaussians.flat_params.to("cuda"))
    for t in (cpu, gpu):
        t.optimizer.step = lambda *a, **k: None
        t.renderer.gaussians.zero_grad = lambda: None
    triples = cpu.sample()
    cpu.train_step(triples)
    for _ in range(3):  # the schedule's overlaps differ run to run: every repetition must agree
        rd.gaussians.flat_grads.zero_()
        gpu.step = 300
        gpu.train_step(triples)
        gc, gg = cpu.renderer.gaussians.flat_grads, rd.gaussians.flat_grads.cpu()
        assert abs(cpu.last_loss.item() - gpu.last_loss.item()) <= 1e-4 * abs(cpu.last_loss.item())
        rel = (gc - gg).abs().sum() / gc.abs().sum()
        assert rel < 5e-4, rel
