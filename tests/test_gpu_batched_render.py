"""The batching behind the reference-shaped call surface (dimo_amd/batched_render.py): `Renderer.render` queues its
renders and runs them as one launch chain at the first use of an output (renderer/latent_gs_renderer.py:1096-1293 is
the surface, main_train_dimo.py:276-318 + :415 the loop it serves).  The batched path must give what the immediate
per-render path gives -- images and every parameter gradient -- however the caller interleaves renders and uses."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(batch, N=6000, M=40, res=96, vae=False, normal=True):
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=N, num_cpts=M, num_motions=4, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=2, resolution=res, vae_latent=vae)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=normal, device="cuda", vae_latent=vae,
                  batch_renders=batch)
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=3, num_latent=cfg.num_motions)
    tr = Trainer(cfg, rd, direct=False)
    tr.step = 300
    tr.find_knn(4)
    return tr, rd


def _grads(rd):
    g = rd.gaussians
    out = {n: getattr(g, n).grad.detach().clone() for n in ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc",
                                                             "_c_xyz", "_c_radius", "_latent_codes")}
    out["timenet"] = torch.cat([p.grad.reshape(-1) for p in g._timenet.parameters()])
    return out


def _loss_of(outs, ws):
    loss = 0
    for o, w in zip(outs, ws):
        loss = loss + (o["image"] * w[0]).sum() + (o["depth"] * w[1]).sum() + (o["alpha"] * w[3]).sum()
        if o["normal"] is not None:
            loss = loss + (o["normal"] * w[2]).sum()
        loss = loss + o["cpts_t"].square().sum()
    return loss


def _render_all(tr, rd, consume_each, deform_mode, cams):
    g = rd.gaussians
    outs = []
    deforms = tr.batched_deform([(m, 0, f) for (m, f, _) in cams]) if deform_mode == "given" else {}
    for (m, f, az) in cams:
        cam = tr.cams.get(0.0, az, tr.cfg.radius, tr.cfg.resolution, tr.cfg.resolution)
        o = rd.render(cam, time=tr.source_time[f], stage="s2", latent_index=m, deform=deforms.get((m, 0, f)))
        if consume_each:
            _ = float(o["alpha"].sum())  # (the reference's per-render .item() calls: main_train_dimo.py:303)
        outs.append(o)
    return outs


CAMS = [(0, 1, 0.0), (0, 1, 90.0), (1, 2, 30.0), (1, 2, 200.0), (2, 0, 10.0), (0, 1, 300.0), (3, 5, 45.0), (1, 4, 120.0),
        (2, 0, 250.0), (3, 3, 10.0)]


@pytest.mark.parametrize("deform_mode", ["given", "lazy"])
@pytest.mark.parametrize("consume_each", [False, True])
def test_batched_equals_immediate(deform_mode, consume_each):
    res = {}
    gen = torch.Generator().manual_seed(5)
    ws = [[torch.randn(c, 96, 96, generator=gen).cuda() for c in (3, 1, 3, 1)] for _ in CAMS]
    for batch in (False, True):
        tr, rd = _setup(batch)
        rd.gaussians.zero_grad()
        outs = _render_all(tr, rd, consume_each, deform_mode, CAMS)
        if batch and not consume_each:
            assert rd._batcher.flushes == 0 and rd._batcher.pending is not None  # nothing has run yet
        loss = _loss_of(outs, ws)
        loss.backward()
        imgs = torch.stack([torch.cat([o["image"], o["depth"], o["normal"], o["alpha"]]) for o in outs])
        radii = torch.stack([o["radii"] + 0 for o in outs])
        vis = torch.stack([o["visibility_filter"] & True for o in outs])
        sink = torch.stack([o["viewspace_points"].grad for o in outs])
        res[batch] = (loss.detach(), imgs.detach(), radii, vis, sink, _grads(rd))
        if batch:
            b = rd._batcher
            assert b.rendered == len(CAMS) and b.flushes == (len(CAMS) if consume_each else 1)
            assert not any(b.in_use), "render slots must be free again after the backward"
    (l0, i0, r0, v0, s0, g0), (l1, i1, r1, v1, s1, g1) = res[False], res[True]
    assert torch.equal(r0, r1) and torch.equal(v0, v1)
    assert (i0 - i1).abs().max() <= 1e-5
    assert abs(float(l0 - l1)) <= 1e-5 * abs(float(l0))
    assert (s0 - s1).abs().sum() <= 1e-4 * s0.abs().sum()
    for k in g0:
        err = (g0[k] - g1[k]).abs().sum() / (g0[k].abs().sum() + 1e-12)
        assert err <= 1e-4, (k, float(err))


def test_no_grad_renders_release_their_slots_and_mixed_calls_stay_ordered():
    tr, rd = _setup(True)
    cam = tr.cams.get(0.0, 40.0, tr.cfg.radius, 96, 96)
    with torch.no_grad():
        for k in range(40):  # more than the 16 render slots: every batch gives its slots back
            o = rd.render(cam, time=0.25, stage="s2", latent_index=k % 4)
            if k % 5 == 4:
                assert torch.isfinite(o["image"]).all() and int((o["radii"] > 0).sum()) > 0
    rd.flush()
    assert not any(rd._batcher.in_use)
    # an unbatchable call (override_color) in between flushes what is pending and renders immediately
    o1 = rd.render(cam, time=0.5, stage="s2", latent_index=1)
    o2 = rd.render(cam, time=0.5, stage="s2", latent_index=1, override_color=torch.rand(6000, 3, device="cuda"))
    assert isinstance(o2["image"], torch.Tensor) and rd._batcher.pending is None
    assert torch.isfinite(o1["image"]).all()
    # pts_t / cpts_t of a batched render
    o3 = rd.render(cam, time=0.5, stage="s2", latent_index=1)
    assert tuple(o3["pts_t"].shape) == (6000, 3) and tuple(o3["cpts_t"].shape) == (40, 3)


def test_reference_shaped_train_steps_batched_equals_unbatched():
    """Two trainers on the autograd (reference-shaped) path, one with the batching switched off: the same loss and the
    same flat gradient bucket for the same step (before Adam, whose first updates lr * g / |g| turn rounding noise
    around zero into sign flips), then a few real steps for the slot bookkeeping."""
    flat, loss = {}, {}
    for batch in (False, True):
        tr, rd = _setup(batch, N=5000, M=32, res=64)
        tr.step = 250
        step, zero = tr.optimizer.step, rd.gaussians.zero_grad
        tr.optimizer.step = lambda *a, **k: None
        rd.gaussians.zero_grad = lambda: None
        triples = tr.sample()
        tr.train_step(triples)
        flat[batch], loss[batch] = rd.gaussians.flat_grads.detach().clone(), float(tr.last_loss)
        tr.optimizer.step, rd.gaussians.zero_grad = step, zero
        zero()
        for _ in range(3):
            tr.train_step()
        torch.cuda.synchronize()
        assert torch.isfinite(tr.last_loss) and torch.isfinite(rd.gaussians.flat_params).all()
        if batch:
            assert rd._batcher.flushes == 4 and rd._batcher.rendered == 32 and not any(rd._batcher.in_use)
    assert abs(loss[False] - loss[True]) <= 1e-5 * abs(loss[False])
    err = (flat[False] - flat[True]).abs().sum() / flat[False].abs().sum()
    assert err <= 1e-4, float(err)


def test_fused_smoothness_drop_ins_match_pytorch():
    from dimo_amd.losses import compute_bilateral_normal_smoothness_loss as bil
    from dimo_amd.losses import compute_edge_aware_smoothness_loss as edge
    gen = torch.Generator().manual_seed(1)
    rgb = torch.rand(3, 40, 56, 3, generator=gen).cuda().requires_grad_(True)
    depth = torch.randn(3, 40, 56, 1, generator=gen).cuda().requires_grad_(True)
    normal = torch.randn(3, 40, 56, 3, generator=gen).cuda().requires_grad_(True)
    for fn, x in ((edge, depth), (bil, normal)):
        grads = []
        for fused in (True, False):
            for t in (rgb, x):
                t.grad = None
            v = fn(x, rgb, assume_unit_range=fused)
            v.backward()
            grads.append((v.detach(), rgb.grad.clone(), x.grad.clone()))
        (v0, a0, b0), (v1, a1, b1) = grads
        assert abs(float(v0 - v1)) <= 1e-5 * abs(float(v1))
        assert (a0 - a1).abs().sum() <= 1e-4 * a1.abs().sum() and (b0 - b1).abs().sum() <= 1e-4 * b1.abs().sum()
