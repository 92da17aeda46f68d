"""GPU parity of the fused image-loss kernel (+ fused SSIM with clamping) vs the torch restatement of the
reference's loss assembly, and of the trainer's direct HIP pipeline vs its autograd pipeline."""
import numpy as np
import pytest
import torch

from oracle.losses_ref import motion_loss_ref

pytestmark = pytest.mark.gpu


def _lam(cfg):
    return dict(mse=cfg.lambda_mse, ssim=cfg.lambda_ssim, mask=cfg.lambda_mask, smooth=cfg.lambda_smooth,
                bilateral=cfg.lambda_bilateral)


@pytest.mark.parametrize("B,H,W,share,dn", [(4, 64, 48, 1.0, (True, True)), (1, 33, 70, 0.5, (True, False)),
                                            (3, 128, 128, 0.75, (False, True)), (2, 40, 40, 1.0, (False, False)),
                                            # a wave owns 62 columns x 8 rows: one-column last strip, one-row last block
                                            (2, 17, 125, 1.0, (True, True)), (1, 9, 63, 1.0, (True, True)),
                                            (1, 2, 300, 1.0, (True, True)), (1, 70, 2, 1.0, (True, True)),
                                            (2, 512, 512, 1.0, (True, True))])
def test_image_loss_kernel_vs_reference_assembly(B, H, W, share, dn):
    from dimo_amd import _lib
    from dimo_amd.image_loss import fused_image_loss, loss_weights
    from dimo_amd.trainer import TrainConfig
    cfg = TrainConfig(add_depth=dn[0], add_normal=dn[1])
    g = torch.Generator().manual_seed(B * H + W)
    image = torch.rand(B, 3, H, W, generator=g) * 1.4 - 0.2  # values outside [0,1] exercise the clamp mask
    image[:, :, :4] = 1.0  # exact boundary values (white background) must pass gradient like torch.clamp
    image[:, :, 4:6] = 0.0  # ... and the lower boundary (a mutant that closed it went unnoticed: tools/mutate_emulated.py)
    depth = torch.rand(B, 1, H, W, generator=g) * 2
    normal = torch.randn(B, 3, H, W, generator=g)
    alpha = torch.rand(B, 1, H, W, generator=g)
    gt = torch.rand(B, 3, H, W, generator=g)
    mask = (torch.rand(1, H, W, generator=g) > 0.5).float()
    wts = [1.0 if b % 2 == 0 else 0.5 for b in range(B)]
    n_img = round(B / share)
    # ---- reference assembly in float64 on CPU
    leaves = [t.clone().double().requires_grad_(True) for t in (image, depth, normal, alpha)]
    ref = motion_loss_ref(leaves[0], leaves[1] if dn[0] else None, leaves[2] if dn[1] else None, leaves[3],
                          gt.double(), mask.double(), wts, _lam(cfg), share=B / n_img)
    ref.backward()
    # ---- HIP: ssim fwd/bwd (clamped) + loss kernel
    L = _lib.lib()
    d = lambda t: t.cuda().contiguous()
    img_d, dep_d, nrm_d, alp_d, gt_d, mask_d = map(d, (image, depth, normal, alpha, gt, mask))
    ssum = torch.empty(1, device="cuda")
    partials = torch.empty(3, B, 3, H, W, device="cuda")
    st = _lib.current_stream()
    _lib.check(L.dimo_ssim_forward(B, 3, H, W, 1, _lib.ptr(img_d), _lib.ptr(gt_d), _lib.ptr(ssum), _lib.ptr(partials), st), "f")
    coef = torch.tensor([-cfg.lambda_ssim * B / n_img], device="cuda")
    sg = torch.empty(B, 3, H, W, device="cuda")
    _lib.check(L.dimo_ssim_backward(B, 3, H, W, 1, _lib.ptr(img_d), _lib.ptr(gt_d), _lib.ptr(partials), _lib.ptr(coef), _lib.ptr(sg), st), "b")
    acc = torch.zeros(512, device="cuda")  # DIMO_LOSS_WORDS
    w_mse = [cfg.lambda_mse * w / (3 * H * W) for w in wts]
    gdot = torch.full((B, 1, H, W), float("nan"), device="cuda")
    gi, gd, gn, ga = fused_image_loss(img_d, dep_d if dn[0] else None, nrm_d if dn[1] else None, alp_d, gt_d, mask_d,
                                      w_mse, loss_weights(cfg, B, n_img, H, W), sg, acc, g_dot=gdot)
    # the per-pixel plane the rasterizer backward consumes: sum over the channels of gradient x rendered value
    want = (gi * img_d).sum(1, keepdim=True) + ga * alp_d
    if gd is not None:
        want = want + gd * dep_d
    if gn is not None:
        want = want + (gn * nrm_d).sum(1, keepdim=True)
    assert torch.isfinite(gdot).all()
    assert (gdot - want).abs().max().item() <= 1e-6 * max(want.abs().max().item(), 1e-12) + 1e-12
    loss = acc.sum().item() + cfg.lambda_ssim * (B / n_img) * (1 - ssum[0].item() / (B * 3 * H * W))
    assert abs(loss - ref.item()) <= 2e-5 * abs(ref.item()), (loss, ref.item())
    for got, leaf, name in ((gi, leaves[0], "image"), (gd, leaves[1], "depth"), (gn, leaves[2], "normal"),
                            (ga, leaves[3], "alpha")):
        if got is None:
            assert leaf.grad is None
            continue
        r = leaf.grad
        rel = (got.cpu().double() - r).abs().sum() / (r.abs().sum() + 1e-12)
        assert rel <= 1e-4, (name, rel.item())


@pytest.mark.parametrize("B,H,W,share,dn", [(4, 64, 48, 1.0, (True, True)), (1, 33, 70, 0.5, (True, False)),
                                            (3, 128, 128, 0.75, (False, True)), (2, 40, 40, 1.0, (False, False)),
                                            (2, 17, 125, 1.0, (True, True)), (1, 9, 63, 1.0, (True, True)),
                                            (1, 2, 300, 1.0, (True, True)), (1, 70, 2, 1.0, (True, True)),
                                            (2, 512, 512, 1.0, (True, True))])
def test_one_pass_ssim_and_image_losses_vs_reference_assembly(B, H, W, share, dn):
    """dimo_ssim_image_loss (SSIM + every other image term of a motion's batch in ONE tile pass: csrc/ssim.hip) against
    the float64 restatement of the reference's loss assembly (main_train_dimo.py:331-372, src/loss.py:64-106,132-175)
    -- the same inputs and tolerances as the two-kernel test above, per-image targets / masks handed over as pointer
    lists like the trainer does."""
    from dimo_amd.image_loss import fused_ssim_image_loss, loss_weights
    from dimo_amd.trainer import TrainConfig
    cfg = TrainConfig(add_depth=dn[0], add_normal=dn[1])
    g = torch.Generator().manual_seed(B * H + W)
    image = torch.rand(B, 3, H, W, generator=g) * 1.4 - 0.2
    image[:, :, :4] = 1.0
    image[:, :, 4:6] = 0.0
    depth = torch.rand(B, 1, H, W, generator=g) * 2
    normal = torch.randn(B, 3, H, W, generator=g)
    alpha = torch.rand(B, 1, H, W, generator=g)
    gt = torch.rand(B, 3, H, W, generator=g)
    mask = (torch.rand(1, H, W, generator=g) > 0.5).float()
    wts = [1.0 if b % 2 == 0 else 0.5 for b in range(B)]
    n_img = round(B / share)
    leaves = [t.clone().double().requires_grad_(True) for t in (image, depth, normal, alpha)]
    ref = motion_loss_ref(leaves[0], leaves[1] if dn[0] else None, leaves[2] if dn[1] else None, leaves[3],
                          gt.double(), mask.double(), wts, _lam(cfg), share=B / n_img)
    ref.backward()
    d = lambda t: t.cuda().contiguous()
    img_d, dep_d, nrm_d, alp_d, gt_d, mask_d = map(d, (image, depth, normal, alpha, gt, mask))
    ssum = torch.zeros(1, device="cuda")
    coef = torch.tensor([-cfg.lambda_ssim * B / n_img], device="cuda")
    acc = torch.zeros(512, device="cuda")
    w_mse = [cfg.lambda_mse * w / (3 * H * W) for w in wts]
    gdot = torch.full((B, 1, H, W), float("nan"), device="cuda")
    gt_list = [gt_d[b].contiguous() for b in range(B)]
    mask_list = [mask_d.contiguous() for _ in range(B)]
    gi, gd, gn, ga = fused_ssim_image_loss(img_d, dep_d if dn[0] else None, nrm_d if dn[1] else None, alp_d, gt_list,
                                           mask_list, w_mse, loss_weights(cfg, B, n_img, H, W), coef, ssum, acc,
                                           g_dot=gdot)
    want = (gi * img_d).sum(1, keepdim=True) + ga * alp_d
    if gd is not None:
        want = want + gd * dep_d
    if gn is not None:
        want = want + (gn * nrm_d).sum(1, keepdim=True)
    assert torch.isfinite(gdot).all()
    assert (gdot - want).abs().max().item() <= 1e-6 * max(want.abs().max().item(), 1e-12) + 1e-12
    loss = acc.sum().item() + cfg.lambda_ssim * (B / n_img) * (1 - ssum[0].item() / (B * 3 * H * W))
    assert abs(loss - ref.item()) <= 2e-5 * abs(ref.item()), (loss, ref.item())
    for got, leaf, name in ((gi, leaves[0], "image"), (gd, leaves[1], "depth"), (gn, leaves[2], "normal"),
                            (ga, leaves[3], "alpha")):
        if got is None:
            assert leaf.grad is None
            continue
        r = leaf.grad
        rel = (got.cpu().double() - r).abs().sum() / (r.abs().sum() + 1e-12)
        assert rel <= 1e-4, (name, rel.item())


@pytest.mark.parametrize("vae,arap", [(False, False), (True, False), (False, True)])
def test_direct_pipeline_equals_autograd_pipeline(vae, arap):
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=5000, num_cpts=64, num_motions=4, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=2, resolution=128, vae_latent=vae, use_arap=arap)
    res = []
    for direct in (False, True):
        rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda", vae_latent=vae,
                      capacity=CapacityPolicy(initial=1 << 19) if direct else None)
        init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, num_latent=cfg.num_motions)
        tr = Trainer(cfg, rd, direct=direct)
        tr.step = 300  # past depth/normal_reg_start_iter: every image term is on
        assert tr.direct == direct
        tr.optimizer.step = lambda *a, **k: None
        rd.gaussians.zero_grad = lambda: None
        triples = tr.sample()
        torch.manual_seed(5)  # same VAE eps draws in both pipelines
        tr.train_step(triples)
        res.append((tr.last_loss.item(), rd.gaussians.flat_grads.clone()))
    (la, ga), (lb, gb) = res
    assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
    rel = (ga - gb).abs().sum() / ga.abs().sum()
    assert rel < 1e-4, rel
    # and a real step moves the parameters identically enough
    assert torch.isfinite(gb).all()


class _MaskedTargets:
    """Targets whose mask differs per (view, frame), like source_masks[motion][view][frame] (main_train_dimo.py:284)."""

    def __init__(self, res):
        self.res = res
        self._cache = {}

    def get(self, m, v, f):
        key = (m, v, f)
        if key not in self._cache:
            gen = torch.Generator().manual_seed(1000 * m + 10 * v + f)
            img = torch.rand(3, self.res, self.res, generator=gen)
            yy, xx = torch.meshgrid(torch.arange(self.res), torch.arange(self.res), indexing="ij")
            cx, cy = self.res * (0.3 + 0.1 * v), self.res * (0.35 + 0.08 * f)
            mask = (((xx - cx) ** 2 + (yy - cy) ** 2) <= (0.25 * self.res) ** 2).float()[None]
            self._cache[key] = (img.cuda(), mask.cuda())
        return self._cache[key]


@pytest.mark.parametrize("ga", [None, "chamfer", "l1"])
def test_direct_pipeline_per_image_masks_and_geometry_anchor(ga):
    """(a) every image of a motion's batch is compared with ITS OWN mask (a shared synthetic mask hid a bug here);
    (b) the geometry-anchor term of stage s2 (main_train_dimo.py:231-244,295-303), chamfer and L1 flavours, with its
    step-0 control-point cache: direct pipeline == autograd pipeline."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=4000, num_cpts=64, num_motions=4, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=2, resolution=96, add_ga=ga is not None,
                      ga_chamfer=ga == "chamfer")
    res = []
    for direct in (False, True):
        rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                      capacity=CapacityPolicy(initial=1 << 19) if direct else None)
        init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, num_latent=cfg.num_motions)
        tr = Trainer(cfg, rd, direct=direct, targets=_MaskedTargets(cfg.resolution))
        tr.optimizer.step = lambda *a, **k: None
        rd.gaussians.zero_grad = lambda: None
        if ga is not None:  # the cache of step 0, then moved: at step 0 itself the anchors coincide with the points
            tr.cache_cpts_s1()
            assert tr.cpts_s1.shape == (cfg.num_motions, cfg.num_frames, cfg.num_cpts, 3)
            tr.cpts_s1 += 0.02 * torch.randn(tr.cpts_s1.shape, generator=torch.Generator().manual_seed(3)).cuda()
        tr.step = 300
        tr.train_step(tr.sample())
        res.append((tr.last_loss.item(), rd.gaussians.flat_grads.clone(), rd.gaussians._c_xyz.grad.clone()))
    (la, ga_, ca), (lb, gb, cb) = res
    assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
    assert (ga_ - gb).abs().sum() / ga_.abs().sum() < 1e-4
    assert (ca - cb).abs().sum() / ca.abs().sum() < 1e-4


def test_direct_pipeline_ragged_tiles_and_three_view_groups():
    """72^2 images (4.5 tiles per side: the last tile row / column is partial) and three views of ONE (motion,
    frame) pair per motion (a deformation group of three renders): direct pipeline == autograd pipeline."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=3000, num_cpts=40, num_motions=3, num_frames=5, num_views=5, motions_per_step=2,
                      views_per_step=3, frames_per_step=1, resolution=72)
    res = []
    for direct in (False, True):
        rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                      capacity=CapacityPolicy(initial=1 << 18) if direct else None)
        init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=3, num_latent=cfg.num_motions)
        tr = Trainer(cfg, rd, direct=direct)
        tr.step = 300  # past depth/normal_reg_start_iter: every image term is on
        tr.optimizer.step = lambda *a, **k: None
        rd.gaussians.zero_grad = lambda: None
        tr.train_step(tr.sample())
        res.append((tr.last_loss.item(), rd.gaussians.flat_grads.clone()))
    (la, ga), (lb, gb) = res
    assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
    assert (ga - gb).abs().sum() / ga.abs().sum() < 1e-4


def test_capacity_overflow_is_skipped_on_device_and_recovered_a_step_later():
    """An instance capacity that is too small: the overflowing steps must leave the parameters untouched (device-side
    skip flag), the host -- which reads the counts with a lag of one step and never waits for the device -- raises the
    capacity, and training then proceeds."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=3000, num_cpts=32, num_motions=3, num_frames=4, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=1, resolution=64)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                  capacity=CapacityPolicy(initial=64))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=4, num_latent=cfg.num_motions)
    tr = Trainer(cfg, rd)
    p0 = rd.gaussians.flat_params.clone()
    tr.train_step()
    torch.cuda.synchronize()
    assert torch.equal(rd.gaussians.flat_params, p0), "an overflowing step was applied"
    for _ in range(6):
        tr.train_step()
    torch.cuda.synchronize()
    assert tr.skipped_steps >= 1 and rd.capacity.capacity > 64
    assert not torch.equal(rd.gaussians.flat_params, p0) and torch.isfinite(rd.gaussians.flat_params).all()
    skipped = tr.skipped_steps
    tr.train_step(); tr.train_step(); tr.train_step()
    assert tr.skipped_steps == skipped, "still overflowing after the capacity was raised"


def test_direct_pipeline_with_lpips_term():
    """`use_lpips` (main_train_dimo.py:339-341) on the fixed random-weight stand-in: the direct pipeline adds the
    metric's image gradient to the loss kernel's, and must agree with the autograd pipeline."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=2000, num_cpts=32, num_motions=3, num_frames=4, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=1, resolution=64, use_lpips=True, lambda_lpips=50.0)
    res = []
    for direct in (False, True):
        rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                      capacity=CapacityPolicy(initial=1 << 17) if direct else None)
        init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=2, num_latent=cfg.num_motions)
        tr = Trainer(cfg, rd, direct=direct)
        tr.step = 300  # past depth/normal_reg_start_iter: every image term is on
        tr.optimizer.step = lambda *a, **k: None
        rd.gaussians.zero_grad = lambda: None
        tr.train_step(tr.sample())
        res.append((tr.last_loss.item(), rd.gaussians.flat_grads.clone()))
    (la, ga), (lb, gb) = res
    assert abs(la - lb) <= 1e-4 * abs(la), (la, lb)
    assert (ga - gb).abs().sum() / ga.abs().sum() < 1e-3  # MIOpen may pick different algorithms for the two graphs


def test_direct_pipeline_more_renders_per_motion_than_a_batch():
    """10 renders per motion (5 frames x 2 views): a motion's range is split into launches of at most 8 renders,
    deformation groups must not straddle the split incorrectly.  direct pipeline == autograd pipeline."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=2500, num_cpts=48, num_motions=3, num_frames=7, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=5, resolution=64)
    res = []
    for direct in (False, True):
        rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                      capacity=CapacityPolicy(initial=1 << 17) if direct else None)
        init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=5, num_latent=cfg.num_motions)
        tr = Trainer(cfg, rd, direct=direct)
        tr.step = 300  # past depth/normal_reg_start_iter: every image term is on
        tr.optimizer.step = lambda *a, **k: None
        rd.gaussians.zero_grad = lambda: None
        assert tr.train_step(tr.sample()) == 20
        res.append((tr.last_loss.item(), rd.gaussians.flat_grads.clone()))
    (la, ga), (lb, gb) = res
    assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
    assert (ga - gb).abs().sum() / ga.abs().sum() < 1e-4


def _dp_gpu_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = TrainConfig(num_pts=4000, num_cpts=64, num_motions=6, num_frames=6, num_views=4, motions_per_step=2 * world,
                      views_per_step=2, frames_per_step=1, resolution=96)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                  capacity=CapacityPolicy(initial=1 << 18))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, num_latent=cfg.num_motions)
    tr = Trainer(cfg, rd, rank=rank, world_size=world)
    tr.step = 300
    assert tr.direct and tr._flat_adam
    counts = [tr.train_step() for _ in range(3)]
    torch.cuda.synchronize()
    torch.save(dict(params=rd.gaussians.flat_params.cpu(), counts=counts, skipped=tr.skipped_steps,
                    loss=float(tr.last_loss)), f"{out}/rank{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_direct_pipeline_data_parallel_two_ranks_on_one_device(tmp_path):
    """The world > 1 code path of the HIP pipeline (flat bucket + overflow flag through the all-reduce, FlatAdam
    with the device skip flag) with two ranks sharing cuda:0 over gloo: replicas stay identical and match the
    single-process step on the union of the triples."""
    import os
    import torch.multiprocessing as mp
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    port = 23500 + (os.getpid() % 2000)
    mp.spawn(_dp_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(f"{tmp_path}/rank0.pt"), torch.load(f"{tmp_path}/rank1.pt")
    assert a["counts"] == [4, 4, 4] and b["counts"] == [4, 4, 4] and a["skipped"] == b["skipped"] == 0
    assert torch.equal(a["params"], b["params"]), "replicas diverged"
    cfg = TrainConfig(num_pts=4000, num_cpts=64, num_motions=6, num_frames=6, num_views=4, motions_per_step=4,
                      views_per_step=2, frames_per_step=1, resolution=96)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                  capacity=CapacityPolicy(initial=1 << 18))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, num_latent=cfg.num_motions)
    single = Trainer(cfg, rd)
    single.step = 300
    for _ in range(3):
        single.train_step()
    p = rd.gaussians.flat_params.cpu()
    rel = (p - a["params"]).abs().sum() / p.abs().sum()
    assert rel < 1e-4, rel


def test_executor_modes_agree(monkeypatch):
    """Per-render stream chains (3), fully batched on the caller's stream (0) and batched ranges on private streams (-2)
    -- "-2": the DEFAULT schedule, every motion's chain incl. its rasterizer and skinning backward in order on its own
    stream; "-2/joint": the rasterizer backward as ONE launch over the step's renders on the caller's stream, the pass
    bench.py takes its roofline clock in -- are schedules of the same kernels: one step from the same state must give
    the same loss and gradients.  "-2/two-loss-kernels": the default schedule with DIMO_FUSED_LOSS=0, the SSIM kernel
    followed by the loss kernel instead of the one tile pass (same terms, sums in another order)."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=5000, num_cpts=64, num_motions=4, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=2, resolution=128)
    res = {}
    for mode in ("3", "0", "-2", "-2/joint", "-2/two-loss-kernels"):
        monkeypatch.setenv("DIMO_EXEC_STREAMS", mode.split("/")[0])
        monkeypatch.setenv("DIMO_JOINT_BWD", "1" if mode.endswith("joint") else "0")
        monkeypatch.setenv("DIMO_FUSED_LOSS", "0" if mode.endswith("two-loss-kernels") else "1")
        rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                      capacity=CapacityPolicy(initial=1 << 19))
        init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, num_latent=cfg.num_motions)
        tr = Trainer(cfg, rd)
        tr.step = 300
        tr.optimizer.step = lambda *a, **k: None
        tr.train_step(tr.sample())
        torch.cuda.synchronize()
        g = rd.gaussians
        # the TimeNet weight gradients are summed with hardware atomics (order not fixed): compare those loosely
        res[mode] = (tr.last_loss.item(), g.flat_grads.clone(), g._xyz.grad.clone(), g._c_xyz.grad.clone())
    for mode in ("0", "-2", "-2/joint", "-2/two-loss-kernels"):
        assert abs(res[mode][0] - res["3"][0]) <= 1e-6 * abs(res["3"][0])
        # per-Gaussian gradients: the batched modes sum the two views of a (motion, frame) pair before the skinning
        # backward and reduce a tile's pixels in one wave instead of two, so only the summation order differs
        ref = res["3"][2]
        assert torch.allclose(res[mode][2], ref, rtol=1e-3, atol=1e-5 * float(ref.abs().max())), mode
        assert (res[mode][2] - ref).abs().sum() / ref.abs().sum() < 1e-5, mode
        rel = (res[mode][1] - res["3"][1]).abs().sum() / res["3"][1].abs().sum()
        assert rel < 1e-5, (mode, rel)


def _s1_trainer(cfg, direct, seed=0):
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import Trainer
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                  capacity=CapacityPolicy(initial=1 << 20) if direct else None)
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=seed, num_latent=cfg.num_motions)
    g = rd.gaussians
    # the shared log-radius create_from_pcd makes for stage s1 (renderer/latent_gs_renderer.py:449-451)
    g._r = torch.nn.Parameter(g._scaling.detach().mean() * torch.ones(1, 1, device="cuda"))
    tr = Trainer(cfg, rd, direct=direct)
    assert tr.direct == direct
    return tr, rd


def test_stage_s1_direct_pipeline_equals_autograd_pipeline():
    """Stage s1 (renderer/latent_gs_renderer.py:1176-1177, 1211-1212): the TimeNet moves every Gaussian itself and
    every scale is exp(_r).  HIP path (dimo_timenet_forward on the N Gaussians, s1 deformation kernels, batched
    rasterizer, fused losses) == the reference-shaped autograd path, including d loss / d _r, and the densification
    statistics source (radii, d loss / d means2D of the step's last render)."""
    from dimo_amd.trainer import TrainConfig
    cfg = TrainConfig(num_pts=1500, num_cpts=64, num_motions=4, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=2, resolution=96, stage="s1", FPS_iter=10 ** 9,
                      density_end_iter=0)  # (outside the density window: the statistics source stays readable)
    res = []
    for direct in (False, True):
        tr, rd = _s1_trainer(cfg, direct)
        tr.optimizer.step = lambda *a, **k: None
        rd.gaussians.zero_grad = lambda: None
        tr.step = 300
        tr.train_step(tr.sample())
        torch.cuda.synchronize()
        g = rd.gaussians
        radii, g2d = tr._last_stats
        res.append((tr.last_loss.item(), g.flat_grads.clone(), g._r.grad.clone(), g._xyz.grad.clone(),
                    radii.clone(), g2d.clone()))
    (la, ga, ra, xa, rada, g2a), (lb, gb, rb, xb, radb, g2b) = res
    assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
    assert (ga - gb).abs().sum() / ga.abs().sum() < 1e-4
    assert abs(float(ra) - float(rb)) <= 1e-4 * abs(float(ra)) and float(ra) != 0.0
    assert (xa - xb).abs().sum() / xa.abs().sum() < 1e-4
    assert torch.equal(rada, radb)
    assert (g2a - g2b).abs().sum() / g2a.abs().sum() < 1e-4


def test_stage_s1_trains_across_fps_and_densification_on_the_hip_path():
    """FPS at step 0 (num_pts -> num_cpts Gaussians), densification statistics from the executor's last slot, two
    densify_and_prune calls (the flat buckets, FlatAdam moments, executor slots and the fused TimeNet follow the
    changing Gaussian count), all without leaving the HIP pipeline."""
    from dimo_amd.trainer import TrainConfig
    cfg = TrainConfig(num_pts=2000, num_cpts=96, num_motions=3, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=1, resolution=64, stage="s1", density_start_iter=1,
                      densification_interval=2, densify_grad_threshold=1e-9, position_lr_max_steps=500)
    tr, rd = _s1_trainer(cfg, True, seed=1)
    g = rd.gaussians
    sizes, losses = [], []
    for _ in range(6):
        tr.train_step()
        sizes.append(g._xyz.shape[0])
        losses.append(float(tr.last_loss))
    torch.cuda.synchronize()
    assert sizes[0] == 96 and sizes[-1] != 96, sizes
    assert tr.direct and tr._exec.N == sizes[-1] or tr._exec.N == sizes[-2]
    assert all(np.isfinite(losses)) and torch.isfinite(g.flat_params).all()
    assert tr.skipped_steps == 0


def test_direct_pipeline_follows_the_progressive_resolution():
    """main_train_dimo.py:261: 128^2 -> 256^2 at step 300: the executor's slots, the camera cache and the resampled
    targets follow the render size; direct pipeline == autograd pipeline on both sides of the change."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=3000, num_cpts=48, num_motions=3, num_frames=5, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=1, resolution=256)
    res = []
    for direct in (False, True):
        rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                      capacity=CapacityPolicy(initial=1 << 19) if direct else None)
        init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, num_latent=cfg.num_motions)
        tr = Trainer(cfg, rd, direct=direct)
        tr.optimizer.step = lambda *a, **k: None
        sizes = []
        tr.step = 298
        for _ in range(2):  # steps 299 (128^2) and 300 (256^2)
            rd.gaussians.zero_grad()
            tr.train_step(tr.sample())
            sizes.append(tr.render_resolution())
            res.append((direct, tr.last_loss.item(), rd.gaussians.flat_grads.clone()))
        assert sizes == [128, 256]
        if direct:
            assert tr._exec.H == 256
    for i in range(2):
        (_, la, ga), (_, lb, gb) = res[i], res[2 + i]
        assert abs(la - lb) <= 1e-5 * abs(la), (i, la, lb)
        assert (ga - gb).abs().sum() / ga.abs().sum() < 1e-4, i


def test_spatial_sort_is_a_relabelling_of_the_same_training():
    """TrainConfig.spatial_sort only permutes the rows of the canonical Gaussians (densify.py: sort_spatially): the
    same triples give the same loss and, row for row through the permutation, the same gradients (up to the order of
    floating-point sums in the tile lists: equal depths are ordered by row index)."""
    from dimo_amd.densify import morton_order
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    out = []
    for sort in (False, True):
        cfg = TrainConfig(num_pts=6000, num_cpts=64, num_motions=3, num_frames=5, num_views=4, motions_per_step=2,
                          views_per_step=2, frames_per_step=2, resolution=96, spatial_sort=sort,
                          progressive_resolution=False)
        rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                      capacity=CapacityPolicy(initial=1 << 18))
        init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=11, num_latent=cfg.num_motions)
        xyz0 = rd.gaussians._xyz.detach().clone()
        tr = Trainer(cfg, rd)
        tr.step = 300
        tr.optimizer.step = lambda *a, **k: None
        rd.gaussians.zero_grad = lambda: None
        assert tr.train_step(tr.sample()) == 8
        out.append((tr.last_loss.item(), rd.gaussians._xyz.detach().clone(), rd.gaussians._xyz.grad.clone(),
                    rd.gaussians._c_xyz.grad.clone(), xyz0))
    (l0, x0, g0, c0, raw), (l1, x1, g1, c1, _) = out
    perm = morton_order(raw)
    assert torch.equal(x1, x0[perm]) and not torch.equal(perm, torch.arange(len(perm), device=perm.device))
    assert abs(l0 - l1) <= 1e-5 * abs(l0), (l0, l1)
    assert (g1 - g0[perm]).abs().sum() / g0.abs().sum() < 1e-4
    assert (c1 - c0).abs().sum() / c0.abs().sum() < 1e-4
