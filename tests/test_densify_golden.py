"""Densification / pruning / optimizer surgery and PLY / pth checkpoint I/O against golden
vectors captured from the reference's own `GaussianModel` (tests/golden/make_golden.py, section 8): parameter values,
Adam moments and step counters after densify_and_prune -> prune -> reset_opacity must match row for row."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def _model(gold, device="cpu"):
    from dimo_amd.gaussian_model import GaussianModel
    from dimo_amd.trainer import TrainConfig
    from torch import nn
    g = GaussianModel(0, num_latent_code=3, latent_code_dim=32, device=device, dist2_fn=lambda x: None)
    P = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float32, device=device).requires_grad_(True))
    g._xyz, g._features_dc, g._features_rest = P(gold["init.xyz"]), P(gold["init.f_dc"]), P(gold["init.f_rest"])
    g._opacity, g._scaling, g._rotation = P(gold["init.opacity"]), P(gold["init.scaling"]), P(gold["init.rotation"])
    gen = torch.Generator().manual_seed(1)
    g._c_xyz = P((torch.rand(16, 3, generator=gen) - 0.5).numpy())
    g._c_radius = P(torch.log(torch.rand(16, 1, generator=gen) * 0.1 + 0.05).numpy())
    g.spatial_lr_scale = 1.0
    g.max_radii2D = torch.zeros(g._xyz.shape[0], device=device)
    return g, TrainConfig()


def _two_steps(g, gold, step_kw=None):
    for it in range(2):
        g.zero_grad()
        for k, p in g.per_gaussian().items():
            if p.numel():
                p.grad.copy_(torch.tensor(gold[f"grad{it}.{k}"], device=p.device))
        g.optimizer.step(**(step_kw or {}))
    dev = g._xyz.device
    g.xyz_gradient_accum = torch.tensor(gold["accum"], device=dev)
    g.denom = torch.tensor(gold["denom"], device=dev)
    g.max_radii2D = torch.tensor(gold["max_radii2D"], device=dev)


def _check(g, gold, tag, atol=1e-6, rtol=1e-5):
    for k, p in g.per_gaussian().items():
        want = gold[f"{tag}.{k}"]
        assert tuple(p.shape) == want.shape, (tag, k, p.shape, want.shape)
        if p.numel() == 0:
            continue
        np.testing.assert_allclose(p.detach().cpu().numpy(), want, rtol=1e-6, atol=atol, err_msg=f"{tag}.{k}")
        m, v, step = g._moments(p)
        np.testing.assert_allclose(m.cpu().numpy(), gold[f"{tag}.exp_avg.{k}"], rtol=rtol, atol=1e-9, err_msg=f"{tag}.m.{k}")
        np.testing.assert_allclose(v.cpu().numpy(), gold[f"{tag}.exp_avg_sq.{k}"], rtol=rtol, atol=1e-12, err_msg=f"{tag}.v.{k}")
        assert float(step) == float(gold[f"{tag}.step.{k}"])
    np.testing.assert_allclose(g.xyz_gradient_accum.cpu().numpy(), gold[f"{tag}.accum"], rtol=1e-6)
    np.testing.assert_allclose(g.denom.cpu().numpy(), gold[f"{tag}.denom"])
    np.testing.assert_allclose(g.max_radii2D.cpu().numpy(), gold[f"{tag}.max_radii2D"], rtol=1e-6)


def test_densify_prune_reset_match_reference():
    gold = np.load(os.path.join(GOLD, "densify.npz"))
    g, cfg = _model(gold)
    g.training_setup(cfg)
    _two_steps(g, gold)
    _check(g, gold, "before")
    thr = gold["thresholds"]
    torch.manual_seed(int(gold["seed_split"]))
    g.densify_and_prune(float(thr[0]), min_opacity=float(thr[1]), extent=float(thr[2]), max_screen_size=float(thr[3]))
    _check(g, gold, "densified")
    # the flat buckets were rebuilt: every parameter (and gradient) is a view into them again
    base, n = g.flat_params.data_ptr(), g.flat_params.numel()
    for grp in g.optimizer.param_groups:
        for p in grp["params"]:
            assert base <= p.data_ptr() < base + 4 * n and p.grad.data_ptr() >= g.flat_grads.data_ptr()
    g.max_radii2D = torch.tensor(gold["prune.max_radii2D"])
    g.prune(min_opacity=float(thr[4]), extent=float(thr[2]), max_screen_size=float(thr[3]))
    _check(g, gold, "pruned")
    g.reset_opacity()
    _check(g, gold, "reset")
    # and training goes on: one more Adam step on the rebuilt optimizer
    g.zero_grad()
    g._xyz.grad.fill_(0.01)
    before = g._xyz.detach().clone()
    g.optimizer.step()
    assert not torch.equal(before, g._xyz.detach())


def _model_r(gold, radius):
    from torch import nn
    g, cfg = _model(gold)
    g._r = nn.Parameter(torch.log(torch.tensor([[radius]], dtype=torch.float32)).requires_grad_(True))
    return g, cfg


@pytest.mark.parametrize("case,radius", [("split", 0.05), ("clone", 0.03)])
def test_stage_s1_densification_uses_the_shared_radius(case, radius):
    """Stage s1 (the only stage that densifies): `_r` is the shared (1, 1) log-radius, get_scaling = exp(_r), and the
    clone / split selection, the split's stds and new_scaling and the world-size prune follow it
    (renderer/latent_gs_renderer.py:341-351,826-890).  Golden: the reference's own GaussianModel with `_r` set."""
    gold = np.load(os.path.join(GOLD, "densify_r.npz"))
    g, cfg = _model_r(gold, radius)
    g.training_setup(cfg)
    assert "r" in [grp["name"] for grp in g.optimizer.param_groups]
    for it in range(2):
        g.zero_grad()
        for k, p in g.per_gaussian().items():
            if p.numel():
                p.grad.copy_(torch.tensor(gold[f"grad{it}.{k}"]))
        g._r.grad.fill_(0.02 * (it + 1))
        g.optimizer.step()
    np.testing.assert_allclose(g._r.detach().numpy(), gold[f"{case}.r_before"], rtol=1e-6)
    g.xyz_gradient_accum, g.denom = torch.tensor(gold["accum"]), torch.tensor(gold["denom"])
    g.max_radii2D = torch.tensor(gold["max_radii2D"])
    thr = gold["thresholds"]
    torch.manual_seed(int(gold["seed_split"]))
    g.densify_and_prune(float(thr[0]), min_opacity=float(thr[1]), extent=float(thr[2]), max_screen_size=float(thr[3]))
    _check(g, gold, f"{case}.densified")
    # the shared radius and its Adam state are untouched by the surgery
    np.testing.assert_allclose(g._r.detach().numpy(), gold[f"{case}.densified.r"], rtol=1e-6)
    m, v, step = g._moments(g._r)
    np.testing.assert_allclose(m.numpy(), gold[f"{case}.densified.exp_avg.r"], rtol=1e-5)
    np.testing.assert_allclose(v.numpy(), gold[f"{case}.densified.exp_avg_sq.r"], rtol=1e-5)
    assert g._xyz.shape[0] != 300
    if case == "split":
        # GUI.FPS (main_train_dimo.py:511-515) hands prune_points an INDEX tensor: ~idx = -idx-1 selects rows N-1-idx
        g.xyz_gradient_accum = torch.arange(g._xyz.shape[0], dtype=torch.float32)[:, None].clone()
        g.prune_points(torch.tensor(gold["fps.idxs"]))
        assert g._xyz.shape[0] == len(gold["fps.idxs"])
        _check(g, gold, "fps")


def test_prune_points_keeps_other_groups_and_their_moments():
    gold = np.load(os.path.join(GOLD, "densify.npz"))
    g, cfg = _model(gold)
    g.training_setup(cfg)
    g.zero_grad()
    g._c_xyz.grad.fill_(0.5)
    g._latent_codes.grad.fill_(-0.25)
    g.optimizer.step()
    m_c = g._moments(g._c_xyz)[0].clone()
    m_l = g._moments(g._latent_codes)[0].clone()
    c_before, lr_before = g._c_xyz.detach().clone(), {x["name"]: x["lr"] for x in g.optimizer.param_groups}
    mask = torch.zeros(g._xyz.shape[0], dtype=torch.bool)
    mask[::3] = True
    keep_xyz = g._xyz.detach()[~mask].clone()
    g.prune_points(mask)
    assert torch.equal(g._xyz.detach(), keep_xyz)
    assert torch.equal(g._c_xyz.detach(), c_before)
    assert torch.equal(g._moments(g._c_xyz)[0], m_c) and torch.equal(g._moments(g._latent_codes)[0], m_l)
    assert {x["name"]: x["lr"] for x in g.optimizer.param_groups} == lr_before


def test_ply_and_model_checkpoint_round_trip(tmp_path):
    from dimo_amd.ply_io import read_ply
    gold = np.load(os.path.join(GOLD, "densify.npz"))
    g, cfg = _model(gold)
    p1, p2 = str(tmp_path / "s2" / "point_cloud_500.ply"), str(tmp_path / "s2" / "point_cloud_c_500.ply")
    g.save_ply(p1, p2)
    head = open(p1, "rb").read(600).split(b"end_header\n")[0].decode().split("\n")
    assert head[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 400"]
    want = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
            "rot_0", "rot_1", "rot_2", "rot_3"]  # construct_list_of_attributes, latent_gs_renderer.py:517-529
    assert [l.split()[-1] for l in head[3:] if l.startswith("property")] == want
    assert all(l.split()[1] == "float" for l in head[3:] if l.startswith("property"))
    raw = open(p1, "rb").read()
    assert len(raw) == raw.index(b"end_header\n") + len(b"end_header\n") + 400 * 17 * 4  # 17 floats per vertex
    v = read_ply(p1)
    assert np.array_equal(v["nx"], np.zeros(400)) and np.allclose(v["f_dc_1"], gold["init.f_dc"][:, 0, 1])
    g2, _ = _model(gold)
    with torch.no_grad():
        g2._xyz.zero_(), g2._c_xyz.zero_()
    g2.load_ply(p1, p2)
    for (k, a), b in zip(g.per_gaussian().items(), g2.per_gaussian().values()):
        assert torch.equal(a.detach(), b.detach()), k
    assert torch.equal(g._c_xyz.detach(), g2._c_xyz.detach()) and torch.equal(g._c_radius.detach(), g2._c_radius.detach())
    assert g2._features_dc.shape == (400, 1, 3) and g2._features_rest.shape == (400, 0, 3)
    # ASCII files (other tools write them) are read too
    asc = tmp_path / "a.ply"
    asc.write_text("ply\nformat ascii 1.0\ncomment x\nelement vertex 2\nproperty float c_x\nproperty float c_y\n"
                   "property float c_z\nproperty float c_radius\nend_header\n1 2 3 4\n5 6 7 8\n")
    assert np.array_equal(read_ply(str(asc))["c_radius"], np.array([4.0, 8.0]))
    # latents + TimeNet state dict
    g.save_model(str(tmp_path / "s2"), step=500)
    assert sorted(os.listdir(tmp_path / "s2"))[:1] == ["latent_codes_500.pth"]
    sd = torch.load(str(tmp_path / "s2" / "timenet_500.pth"))
    assert "deformnet.5.weight" in sd and sd["deformnet.5.weight"].shape == (256, 360)  # the reference's key names
    with torch.no_grad():
        g2._latent_codes.zero_()
        g2._timenet.pts_layers[0].weight.zero_()
    g2.load_model(str(tmp_path / "s2"), step=500)
    assert torch.equal(g2._latent_codes.detach(), g._latent_codes.detach())
    assert torch.equal(g2._timenet.pts_layers[0].weight, g._timenet.pts_layers[0].weight)


@pytest.mark.gpu
def test_densify_on_device_with_flat_adam_then_train():
    """Same golden comparison on the GPU (flat buckets + FlatAdam), then a training step of the HIP pipeline on the
    re-sized model (executor slots, KNN and the flat buckets all follow the new Gaussian count)."""
    gold = np.load(os.path.join(GOLD, "densify.npz"))
    g, cfg = _model(gold, device="cuda")
    g._dist2_fn = None
    g.training_setup(cfg)
    assert type(g.optimizer).__name__ == "FlatAdam"
    _two_steps(g, gold)
    _check(g, gold, "before", atol=2e-6, rtol=1e-4)  # the HIP Adam fuses its multiply-adds
    thr = gold["thresholds"]
    # the split's normal draws come from the device generator: feed the CPU draws of the fixture instead
    torch.manual_seed(int(gold["seed_split"]))
    orig = torch.normal
    torch.normal = lambda mean, std: orig(mean=mean.cpu(), std=std.cpu()).to(std.device)
    try:
        g.densify_and_prune(float(thr[0]), min_opacity=float(thr[1]), extent=float(thr[2]), max_screen_size=float(thr[3]))
    finally:
        torch.normal = orig
    _check(g, gold, "densified", atol=5e-6, rtol=1e-4)


@pytest.mark.gpu
def test_training_goes_on_across_a_prune():
    """Stage-s2 schedule (main_train_dimo.py:439-443): the prune at step 2 removes the Gaussians made transparent
    before it; flat buckets, FlatAdam moments, executor slots, KNN and the HIP TimeNet all follow the new count."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=6000, num_cpts=64, num_motions=4, num_frames=6, num_views=4, motions_per_step=2,
                      views_per_step=2, frames_per_step=1, resolution=96, densification_interval_s2=2)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                  capacity=CapacityPolicy(initial=1 << 18))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, num_latent=cfg.num_motions)
    tr = Trainer(cfg, rd)
    g = rd.gaussians
    tr.train_step()
    with torch.no_grad():
        g._opacity[:1000] = -10.0  # sigmoid -> 4.5e-5 < densify_opacity_threshold_s2
    keep = g._xyz.detach()[1000:].clone()
    m_keep = g._moments(g._xyz)[0][1000:].clone()
    tr.train_step()  # step 2: Adam, then the prune
    n = g._xyz.shape[0]
    assert n <= 5000 and n >= 4000, n
    assert g.flat_params.data_ptr() <= g._xyz.data_ptr() < g.flat_params.data_ptr() + 4 * g.flat_params.numel()
    assert g._moments(g._xyz)[0].shape == (n, 3) and g.optimizer.step_count == 2
    assert torch.isfinite(m_keep).all() and keep.shape[0] == 5000
    tr.train_step()  # step 3 on the re-sized model
    torch.cuda.synchronize()
    assert tr._exec.N == n and torch.isfinite(tr.last_loss) and torch.isfinite(g.flat_params).all()
    assert tr.skipped_steps == 0


def test_sort_spatially_moves_parameters_moments_and_statistics_together():
    """The Morton-order layout step (densify.py: sort_spatially) is a pure permutation of the per-Gaussian rows: every
    parameter, both Adam moments and the densification statistics of a row travel with it, the other groups stay, and
    the densify that follows gives the same SET of Gaussians as without the sort."""
    from dimo_amd.densify import morton_order
    gold = np.load(os.path.join(GOLD, "densify.npz"))
    g, cfg = _model(gold)
    g.training_setup(cfg)
    _two_steps(g, gold)
    before = {k: (p.detach().clone(), g._moments(p)[0].clone(), g._moments(p)[1].clone())
              for k, p in g.per_gaussian().items() if p.numel()}
    stats = (g.xyz_gradient_accum.clone(), g.denom.clone(), g.max_radii2D.clone())
    c_m = g._moments(g._c_xyz)[0].clone()
    perm = g.sort_spatially()
    assert sorted(perm.tolist()) == list(range(perm.shape[0])) and not torch.equal(perm, torch.arange(perm.shape[0]))
    for k, p in g.per_gaussian().items():
        if p.numel():
            assert torch.equal(p.detach(), before[k][0][perm]), k
            assert torch.equal(g._moments(p)[0], before[k][1][perm]) and torch.equal(g._moments(p)[1], before[k][2][perm]), k
    assert torch.equal(g.xyz_gradient_accum, stats[0][perm]) and torch.equal(g.denom, stats[1][perm])
    assert torch.equal(g.max_radii2D, stats[2][perm]) and torch.equal(g._moments(g._c_xyz)[0], c_m)
    # sorted: neighbours in memory are neighbours in space (mean hop much shorter than between random rows), idempotent
    x = g._xyz.detach()
    assert (x[1:] - x[:-1]).norm(dim=1).mean() < 0.6 * (before["xyz"][0][1:] - before["xyz"][0][:-1]).norm(dim=1).mean()
    assert torch.equal(morton_order(x), torch.arange(x.shape[0]))
    # a prune after the sort removes the same Gaussians as before it would have
    g2, _ = _model(gold)
    g2.training_setup(cfg)
    _two_steps(g2, gold)
    g.prune(0.3, 4.0)
    g2.prune(0.3, 4.0)
    key = lambda m: sorted(map(tuple, m._xyz.detach().numpy().round(6).tolist()))
    assert key(g) == key(g2) and 0 < g._xyz.shape[0] < perm.shape[0]
