"""The reference's OWN loop body through the drop-in surface (tests/reference_step.py restates
main_train_dimo.py:246-417 in its order of operations: render -> [GA term on out["cpts_t"] + .item()] ->
out[...].unsqueeze(0) -> torch.cat per motion -> per-image mse on batch[...][k] -> ssim -> mask mse -> smoothness terms
on .permute(0, 2, 3, 1) -> the logged .item() reads -> ONE backward -> optimizer.step).

What must hold for the batching behind `Renderer.render` (dimo_amd/batched_render.py) to serve THAT loop:
  * the whole step is ONE launch chain (`flushes` grows by one per step) -- not one per render, not one per motion --
    with and without the geometry-anchor term, whose per-render `.item()` only needs the TimeNet;
  * numbers equal to rendering every triple immediately (loss, flat gradient bucket);
  * a queued render shows the model of the moment `render()` was called (prune / optimizer step in between),
    keeps the autograd mode it was queued in, and an overflow of the instance capacity in ANY batch is noticed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(batch, N=5000, M=32, res=64, ga=False, chamfer=True, log=True, capacity=None, seed=3):
    from tests.reference_step import ReferenceLoop
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import SyntheticTargets, init_synthetic_model
    from dimo_amd.trainer import TrainConfig
    cfg = TrainConfig(num_pts=N, num_cpts=M, num_motions=4, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=2, resolution=res, add_ga=ga, ga_chamfer=chamfer, seed=seed)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda", batch_renders=batch,
                  capacity=capacity)
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=seed, num_latent=cfg.num_motions)
    rd.gaussians.training_setup(cfg)
    cpts = None
    if ga:
        gen = torch.Generator().manual_seed(11)
        cpts = [[(rd.gaussians._c_xyz.detach() + 0.01 * torch.randn(M, 3, generator=gen).cuda())
                 for _ in range(cfg.num_frames)] for _ in range(cfg.num_motions)]
    loop = ReferenceLoop(cfg, rd, SyntheticTargets(res, "cuda", seed=seed), log_scalars=log, cpts_s1=cpts)
    loop.step = 300  # past depth/normal_reg_start_iter: every image term on
    return loop, rd


@pytest.mark.parametrize("ga", [False, "chamfer", "l1"])
def test_the_reference_loop_is_one_launch_chain_per_step_and_equals_immediate_rendering(ga):
    flat, loss, scal = {}, {}, {}
    for batch in (False, True):
        loop, rd = _setup(batch, ga=bool(ga), chamfer=(ga != "l1"))
        g = rd.gaussians
        opt = g.optimizer
        step, zero = opt.step, opt.zero_grad
        opt.step, opt.zero_grad = (lambda *a, **k: None), (lambda *a, **k: None)  # gradients compared BEFORE Adam
        n = loop.train_step(sample=([1, 3], [0, 2], [1, 4]))
        assert n == 8
        flat[batch], loss[batch], scal[batch] = g.flat_grads.detach().clone(), float(loop.last_loss), dict(loop.scalars)
        opt.step, opt.zero_grad = step, zero
        zero()
        if batch:
            b = rd._batcher
            assert b is not None and b.rendered == 8
            assert b.flushes == 1, f"{b.flushes} launch chains for one step of the reference's loop"
            assert not any(b.in_use), "render slots must be free again after the backward"
        for _ in range(3):  # real steps: slot bookkeeping, optimizer hook, capacity guard
            loop.train_step()
        torch.cuda.synchronize()
        assert torch.isfinite(loop.last_loss) and torch.isfinite(g.flat_params).all()
        if batch:
            assert rd._batcher.flushes == 4 and rd._batcher.rendered == 32 and not any(rd._batcher.in_use)
    assert abs(loss[False] - loss[True]) <= 1e-5 * abs(loss[False])
    err = (flat[False] - flat[True]).abs().sum() / flat[False].abs().sum()
    assert err <= 1e-4, float(err)
    assert set(scal[False]) == set(scal[True]) and len(scal[True]) >= 14
    for k, v in scal[False].items():  # every logged scalar (the reference's tb_writer reads) agrees
        assert abs(v - scal[True][k]) <= 2e-5 * max(1.0, abs(v)), (k, v, scal[True][k])


def test_outputs_are_stand_ins_with_metadata_until_a_value_is_needed():
    from dimo_amd.batched_render import LazyTensor
    loop, rd = _setup(True)
    loop.find_knn(rd.gaussians)
    cam = loop_cam(loop, 64)
    outs = [rd.render(cam, time=0.1 * k, stage="s2", latent_index=k % 4) for k in range(5)]
    b = rd._batcher
    for o in outs:
        assert isinstance(o["image"], LazyTensor) and tuple(o["image"].shape) == (3, 64, 64)
        assert tuple(o["depth"].unsqueeze(0).shape) == (1, 1, 64, 64) and o["alpha"].dtype == torch.float32
        assert o["normal"].device.type == "cuda" and tuple(o["radii"].shape) == (5000,) and o["radii"].dtype == torch.int32
        assert tuple(o["cpts_t"].shape) == (32, 3) and tuple(o["visibility_filter"].shape) == (5000,)
    imgs = torch.cat([o["image"].unsqueeze(0) for o in outs], dim=0)
    hwc = imgs.permute(0, 2, 3, 1)
    assert b.flushes == 0 and b.pending is not None and tuple(hwc.shape) == (5, 64, 64, 3)
    c = outs[3]["cpts_t"][None, ...]                      # the GA term's read: TimeNet only, a real tensor
    assert isinstance(c, torch.Tensor) and b.flushes == 0 and b.pending is not None
    t = hwc.sum()                                         # a value: the batch runs, once
    assert b.flushes == 1 and b.rendered == 5 and torch.isfinite(t)
    full = b_full(outs[0])
    from dimo_amd.batched_render import materialize
    assert materialize(imgs).data_ptr() == full.data_ptr(), "torch.cat of a batch's renders must be a zero-copy view"
    assert materialize(outs[2]["image"]).data_ptr() == full[2].data_ptr()
    assert float((materialize(imgs) - torch.stack([materialize(o["image"]) for o in outs])).abs().max()) == 0.0
    (t + c.sum()).backward()
    assert not any(b.in_use)


def loop_cam(loop, res):
    from dimo_amd.camera import MiniCam, orbit_camera
    o = loop.opt
    return MiniCam(orbit_camera(o.elevation, 40.0, o.radius), res, res, loop.cam.fovy, loop.cam.fovx, loop.cam.near,
                   loop.cam.far, device="cuda")


def b_full(out):
    pend = out["image"]._src[0]
    return pend["full"]["image"]


def test_a_queued_render_shows_the_model_and_the_grad_mode_of_the_call():
    loop, rd = _setup(True)
    g = rd.gaussians
    loop.find_knn(g)
    cam = loop_cam(loop, 64)
    # reference values: immediate rendering of the same model
    loopi, rdi = _setup(False)
    loopi.find_knn(rdi.gaussians)
    with torch.no_grad():
        want = rdi.render(cam, time=0.3, stage="s2", latent_index=2)
    o = rd.render(cam, time=0.3, stage="s2", latent_index=2)          # queued, grad enabled
    n_old = g._xyz.shape[0]
    mask = torch.zeros(n_old, dtype=torch.bool, device="cuda")
    mask[::3] = True
    g.prune_points(mask)                                              # the model changes under the queued render
    assert rd._batcher.flushes == 1 and rd._batcher.pending is None, "a mutator must run what is queued first"
    assert tuple(o["radii"].shape) == (n_old,) and torch.equal(o["radii"] + 0, want["radii"])
    assert float((o["image"] - want["image"]).abs().max()) <= 1e-5
    with torch.no_grad():
        assert (o["image"] * 1.0).requires_grad is False
    loss = o["image"].unsqueeze(0).sum()
    assert loss.requires_grad, "queued with autograd on: the graph must exist"
    del loss, o  # (the graph goes: its render slot is free and the executor can be rebuilt for the new row count;
    #              while a slot is held, renders of a different model take the immediate path)
    assert not any(rd._batcher.in_use)
    # the pruned model renders with its own row count, in a new batch
    loop.find_knn(g)
    with torch.no_grad():
        o2 = rd.render(cam, time=0.3, stage="s2", latent_index=2)
        o3 = rd.render(cam, time=0.3, stage="s2", latent_index=1)
        with torch.enable_grad():                                     # a change of the autograd mode cuts the batch
            o4 = rd.render(cam, time=0.3, stage="s2", latent_index=1)
        assert rd._batcher.flushes == 2
        assert tuple(o2["radii"].shape) == (g._xyz.shape[0],) and not (o3["image"] + 0).requires_grad
    assert (o4["image"] + 0).requires_grad
    # an optimizer step in between: the queued render still shows the parameters of its call
    before = rd.render(cam, time=0.5, stage="s2", latent_index=0)
    g.flat_grads.normal_()
    g.optimizer.step()
    after = rd.render(cam, time=0.5, stage="s2", latent_index=0)
    assert float((before["image"] - after["image"]).abs().max()) > 0.0
    rd.flush()


def test_a_failed_batch_raises_at_every_output_and_frees_its_slots():
    loop, rd = _setup(True)
    loop.find_knn(rd.gaussians)
    cam = loop_cam(loop, 64)
    o = rd.render(cam, time=0.3, stage="s2", latent_index=2)
    b = rd._batcher
    run = b._run
    b._run = lambda pend: (_ for _ in ()).throw(ValueError("injected"))
    with pytest.raises(ValueError):
        o["image"].sum()
    b._run = run
    with pytest.raises(RuntimeError, match="failed"):
        o["depth"].sum()
    assert not any(b.in_use)
    with torch.no_grad():
        assert torch.isfinite(rd.render(cam, time=0.3, stage="s2", latent_index=2)["image"].sum())


def test_an_overflow_in_an_early_batch_is_seen_by_the_guard_many_batches_later():
    """the (R, overflow) words the policy tracks were live views of slot words that the next batch in the
    same slots overwrites; an unpolled renderer looks every 64 renders.  Batch 0 overflows, seven and more batches
    re-use its slots, the guard must still raise."""
    from dimo_amd.rasterizer import CapacityPolicy
    probe = CapacityPolicy(initial=1 << 22)
    loop, rd = _setup(True, capacity=probe)
    loop.find_knn(rd.gaussians)
    cam = loop_cam(loop, 64)
    with torch.no_grad():
        rd.render(cam, time=0.2, stage="s2", latent_index=0)["image"].sum()
        assert probe.check()
        r1 = probe.last_r_max
        rd.render(cam, scaling_modifier=3.0, time=0.2, stage="s2", latent_index=0)["image"].sum()
        assert probe.check()
        r3 = probe.last_r_max
    assert r3 > 1.6 * r1, (r1, r3)
    tight = CapacityPolicy(initial=int(1.3 * r1), margin=1.0)
    loop, rd = _setup(True, capacity=tight)
    loop.find_knn(rd.gaussians)
    with torch.no_grad():
        rd.render(cam, scaling_modifier=3.0, time=0.2, stage="s2", latent_index=0)["image"].sum()  # batch 0: overflows
        with pytest.raises(RuntimeError, match="overflow"):
            for k in range(80):                                                                    # 10 more batches
                o = rd.render(cam, time=0.2, stage="s2", latent_index=k % 4)
                if k % 8 == 7:
                    o["image"].sum()
