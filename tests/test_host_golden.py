"""Host-side logic vs golden vectors captured from the reference's own Python
(tests/golden/make_golden.py): positional encoding, TimeNet forward/backward, quaternion helpers,
cameras, lr schedule, SH helpers, and the whole deform stage of Renderer.render (both latent flavours,
both stages) including gradients of every parameter.  CPU only."""
import os

import numpy as np
import pytest
import torch

from tests.scenes import GRAD_STRIDE, timenet_weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def T(a):
    return torch.from_numpy(np.asarray(a))


def load_weights(net, seed, head_std):
    net.load_state_dict({k: T(v) for k, v in timenet_weights(seed, head_std).items()})


def test_positional_encoding():
    from dimo_amd.deform import get_embedder
    z = gold("pos_enc.npz")
    e3, d3 = get_embedder(10, 3)
    e1, d1 = get_embedder(6, 1)
    assert [d3, d1] == z["dims"].tolist() == [60, 12]
    np.testing.assert_allclose(e3(T(z["x3"])).numpy(), z["emb3"], atol=1e-6)
    np.testing.assert_allclose(e1(T(z["t1"])).numpy(), z["emb1"], atol=1e-6)


@pytest.mark.parametrize("M", [8, 512])
def test_timenet_forward_backward(M):
    from dimo_amd.deform import TimeNet
    z = gold(f"timenet_M{M}.npz")
    net = TimeNet(latent_code_dim=32, device="cpu")
    assert sum(p.numel() for p in net.parameters()) == 647431 and net.input_ch == 104
    assert sorted(net.state_dict()) == sorted(timenet_weights(0))
    load_weights(net, int(z["weight_seed"]), 1e-2)
    pts, lat = T(z["pts"]).requires_grad_(True), T(z["latent"]).requires_grad_(True)
    dp, dq = net(pts, float(z["t"]), lat)
    np.testing.assert_allclose(dp.detach().numpy(), z["dxyz"], atol=2e-6)
    np.testing.assert_allclose(dq.detach().numpy(), z["dquat"], atol=2e-6)
    ((dp * T(z["wp"])).sum() + (dq * T(z["wq"])).sum()).backward()
    np.testing.assert_allclose(pts.grad.numpy(), z["g_pts"], atol=1e-5 * max(1, np.abs(z["g_pts"]).max()))
    np.testing.assert_allclose(lat.grad.numpy(), z["g_latent"], atol=1e-5 * max(1, np.abs(z["g_latent"]).max()))
    for k, p in net.named_parameters():
        ref = z[f"grad.{k}"]
        got = p.grad.numpy().reshape(-1)[::GRAD_STRIDE]
        np.testing.assert_allclose(got, ref, atol=1e-5 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_timenet_init_is_identity_motion():
    from dimo_amd.deform import TimeNet
    torch.manual_seed(0)
    net = TimeNet(device="cpu")
    dp, dq = net(torch.rand(5, 3), 0.3, torch.randn(32))
    assert torch.all(dp == 0) and torch.equal(dq, torch.tensor([1.0, 0, 0, 0]).expand(5, 4))
    mlp, rot = net.get_mlp_parameters()
    assert len(rot) == 4 and len(mlp) + len(rot) == len(list(net.parameters()))


def test_timenet_batched_times():
    from dimo_amd.deform import TimeNet
    z = gold("timenet_tapply.npz")
    net = TimeNet(device="cpu")
    load_weights(net, 100, 1e-2)
    dp, dq = net(T(z["pts"]), T(z["times"]), T(z["latent"]), t_apply=True)
    np.testing.assert_allclose(dp.detach().numpy(), z["dxyz"], atol=2e-6)
    np.testing.assert_allclose(dq.detach().numpy(), z["dquat"], atol=2e-6)


def test_quaternion_helpers_and_lr():
    from dimo_amd.deform import build_rotation, build_rotation_3d, quat_mul
    from dimo_amd.gaussian_model import RGB2SH, get_expon_lr_func
    z = gold("quat_helpers.npz")
    np.testing.assert_allclose(build_rotation_3d(T(z["q"])).numpy(), z["R3d"], atol=1e-6)
    np.testing.assert_allclose(build_rotation(T(z["q"])[:, 2]).numpy(), z["R"], atol=1e-6)
    np.testing.assert_allclose(quat_mul(T(z["q1"]), T(z["q2"])).numpy(), z["qmul"], atol=1e-6)
    z = gold("lr_func.npz")
    f = get_expon_lr_func(lr_init=0.01, lr_final=0.0002, lr_delay_mult=0.02, max_steps=1000)
    f2 = get_expon_lr_func(lr_init=0.005, lr_final=0.0002, lr_delay_steps=100, lr_delay_mult=0.02, max_steps=1000)
    np.testing.assert_allclose([f(s) for s in z["steps"]], z["lr"], rtol=1e-12)
    np.testing.assert_allclose([f2(s) for s in z["steps"]], z["lr_delay"], rtol=1e-12)
    np.testing.assert_allclose(RGB2SH(torch.tensor([0.0, 0.3, 1.0])).numpy(), gold("sh_eval.npz")["rgb2sh"], atol=1e-6)


def test_cameras():
    from dimo_amd.camera import CameraCache, MiniCam, OrbitCamera, orbit_camera
    z = gold("cameras.npz")
    oc = OrbitCamera(800, 800, r=2, fovy=33.9)
    assert abs(oc.fovy - float(z["fovy"])) < 1e-12 and abs(oc.fovx - float(z["fovx"])) < 1e-12
    for i, az in enumerate(z["azimuths"]):
        pose = orbit_camera(0, az, 2)
        np.testing.assert_array_equal(pose, z[f"pose_{i}"])
        mc = MiniCam(pose, 512, 512, oc.fovy, oc.fovx, oc.near, oc.far, device="cpu")
        np.testing.assert_allclose(mc.world_view_transform.numpy(), z[f"wv_{i}"], atol=1e-7)
        np.testing.assert_allclose(mc.full_proj_transform.numpy(), z[f"fp_{i}"], atol=1e-6)
        np.testing.assert_allclose(mc.camera_center.numpy(), z[f"cc_{i}"], atol=0)
    np.testing.assert_allclose(mc.projection_matrix.numpy(), z["proj_512"], atol=1e-7)
    mc = MiniCam(z["pose_e"], 64, 48, oc.fovy, 0.9, 0.05, 50, device="cpu")
    np.testing.assert_allclose(mc.world_view_transform.numpy(), z["wv_e"], atol=1e-7)
    np.testing.assert_allclose(mc.full_proj_transform.numpy(), z["fp_e"], atol=1e-6)
    np.testing.assert_allclose(mc.projection_matrix.numpy(), z["proj_e"], atol=1e-7)
    cache = CameraCache(device="cpu")
    a = cache.get(0, 40.0, 2, 256, 256)
    assert cache.get(0, 40.0, 2, 256, 256) is a and cache.get(0, 40.0, 2, 128, 128) is not a
    np.testing.assert_allclose(a.full_proj_transform.numpy(), z["fp_1"], atol=1e-6)


class _Capture:
    """Stand-in rasterizer recording what Renderer.render hands to the native boundary."""
    last = None

    def __init__(self, settings, with_normal):
        self.s, self.with_normal = settings, with_normal

    def __call__(self, **kw):
        _Capture.last = dict(kw, settings=self.s)
        H, W = self.s.image_height, self.s.image_width
        z = lambda c: torch.zeros(c, H, W)
        n = kw["means3D"].shape[0]
        if self.with_normal:
            return z(3), z(1), z(3), z(1), torch.zeros(n, dtype=torch.int32), None
        return z(3), torch.zeros(n, dtype=torch.int32), z(1), z(1)


@pytest.mark.parametrize("name,vae", [("deform_latent.npz", False), ("deform_vae.npz", True)])
@pytest.mark.parametrize("stage", ["s1", "s2"])
def test_render_deform_stage_matches_reference(name, vae, stage):
    from dimo_amd.camera import MiniCam, OrbitCamera
    from dimo_amd.renderer import Renderer
    z = gold(name)
    rd = Renderer(sh_degree=0, white_background=True, radius=2, num_latent_code=5, latent_code_dim=32,
                  add_normal=True, vae_latent=vae, device="cpu", rasterizer_factory=_Capture)
    g = rd.gaussians
    P = lambda k: torch.nn.Parameter(T(z[f"param.{k}"]).clone())
    g._xyz, g._features_dc, g._scaling = P("xyz"), P("f_dc"), P("scaling")
    g._rotation, g._opacity, g._c_xyz, g._c_radius = P("rotation"), P("opacity"), P("c_xyz"), P("c_radius")
    g._features_rest = torch.nn.Parameter(torch.zeros(300, 0, 3))
    params = dict(xyz=g._xyz, f_dc=g._features_dc, scaling=g._scaling, rotation=g._rotation, opacity=g._opacity,
                  c_xyz=g._c_xyz, c_radius=g._c_radius)
    if vae:
        g._mu, g._log_var = P("mu"), P("log_var")
        params.update(mu=g._mu, log_var=g._log_var)
    else:
        g._latent_codes = P("latent_codes")
        params.update(latent_codes=g._latent_codes)
    load_weights(g._timenet, int(z["weight_seed"]), float(z["head_std"]))
    g.neighbor_dists, g.neighbor_indices = T(z["knn_dist"]), T(z["knn_idx"])
    oc = OrbitCamera(800, 800, r=2, fovy=33.9)
    cam = MiniCam(z["cam.pose"], 64, 64, oc.fovy, oc.fovx, oc.near, oc.far, device="cpu")

    torch.manual_seed(int(z["vae_seed"]))
    out = rd.render(cam, time=float(z["time"]), stage=stage, latent_index=int(z["latent_index"]))
    cap = _Capture.last
    loss = 0
    for key in ("means3D", "opacities", "scales", "rotations", "shs"):
        ref = z[f"{stage}.in.{key}"]
        np.testing.assert_allclose(cap[key].detach().numpy(), ref, atol=2e-6 * max(1, np.abs(ref).max()), err_msg=key)
        loss = loss + (cap[key] * T(z[f"{stage}.w.{key}"])).sum()
    np.testing.assert_allclose(out["cpts_t"].detach().numpy(), z[f"{stage}.cpts_t"], atol=2e-6)
    np.testing.assert_allclose(out["pts_t"].detach().numpy(), z[f"{stage}.pts_t"], atol=2e-6)
    loss = loss + (out["cpts_t"] * T(z[f"{stage}.w.cpts_t"])).sum()
    s = cap["settings"]
    np.testing.assert_allclose([s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier, s.sh_degree],
                               z[f"{stage}.settings"], rtol=1e-12)
    np.testing.assert_array_equal(s.bg.numpy(), z[f"{stage}.bg"])
    assert cap["cov3Ds_precomp"] is None and cap["extra_attrs"] is None and cap["colors_precomp"] is None
    assert out["normal"] is not None and out["viewspace_points"].requires_grad

    loss.backward()
    for k, p in params.items():
        ref = z[f"{stage}.grad.{k}"]
        got = np.zeros_like(ref) if p.grad is None else p.grad.numpy()
        np.testing.assert_allclose(got, ref, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)
    for k, p in g._timenet.named_parameters():
        ref = z[f"{stage}.grad.timenet.{k}"]
        got = (torch.zeros_like(p) if p.grad is None else p.grad).numpy().reshape(-1)[::GRAD_STRIDE]
        np.testing.assert_allclose(got, ref, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_image_losses_match_reference():
    from dimo_amd import losses
    from oracle.losses_ref import ssim_ref
    z = gold("image_losses.npz")
    a, b = T(z["img1"]).requires_grad_(True), T(z["img2"])
    s = ssim_ref(a, b)
    s.backward()
    assert abs(s.item() - float(z["ssim"])) < 1e-6
    np.testing.assert_allclose(a.grad.numpy(), z["g_img1"], atol=1e-7 + 1e-4 * np.abs(z["g_img1"]).max())
    d, nrm, rgb = (T(z[k]).requires_grad_(True) for k in ("depth", "normal", "rgb"))
    ea = losses.compute_edge_aware_smoothness_loss(d, rgb)
    bl = losses.compute_bilateral_normal_smoothness_loss(nrm, rgb)
    (ea + bl).backward()
    assert abs(ea.item() - float(z["edge_aware"])) < 1e-6 and abs(bl.item() - float(z["bilateral"])) < 1e-6
    for t, k in ((d, "g_depth"), (nrm, "g_normal"), (rgb, "g_rgb")):
        np.testing.assert_allclose(t.grad.numpy(), z[k], atol=1e-8 + 1e-4 * np.abs(z[k]).max())
