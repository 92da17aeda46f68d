"""Generates tests/golden/*.npz by IMPORTING the reference's Python (this container only).

    python tests/golden/make_golden.py            # needs /root/reference

The reference hard-codes "cuda" and imports CUDA-only extensions, so in THIS
process only: (1) stub modules are registered for the absent native
extensions / third parties, (2) torch factory calls and
`.cuda()/.to("cuda")` are redirected to the CPU.  Nothing of the reference
(source, bytecode) is written anywhere: the fixtures are inputs + outputs only.

The stub rasterizer captures the tensors `Renderer.render` hands to
`diff_gauss` (renderer/latent_gs_renderer.py:1255-1266), which pins the whole
deform stage (latent lookup -> TimeNet -> LBS -> quat_mul -> normalize ->
activations) including its backward, without the absent CUDA rasterizer.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)

# ----------------------------------------------------------------------------- cuda -> cpu redirection
for name in ("zeros", "ones", "tensor", "randn", "rand", "empty", "zeros_like", "ones_like", "randn_like", "full",
             "eye", "arange", "linspace"):
    orig = getattr(torch, name)

    def mk(orig):
        def f(*a, **k):
            if "device" in k and "cuda" in str(k["device"]):
                k["device"] = "cpu"
            return orig(*a, **k)
        return f
    setattr(torch, name, mk(orig))
torch.Tensor.cuda = lambda self, *a, **k: self
_to = torch.Tensor.to


def _tensor_to(self, *a, **k):
    a = tuple("cpu" if (isinstance(x, (str, torch.device)) and "cuda" in str(x)) else x for x in a)
    if "device" in k and "cuda" in str(k["device"]):
        k["device"] = "cpu"
    return _to(self, *a, **k)


torch.Tensor.to = _tensor_to
_mto = torch.nn.Module.to
torch.nn.Module.to = lambda self, *a, **k: self if any("cuda" in str(x) for x in a) else _mto(self, *a, **k)

# ----------------------------------------------------------------------------- stubs for absent modules
CAPTURE = {}


class _Settings:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Rasterizer:
    def __init__(self, raster_settings):
        self.s = raster_settings

    def __call__(self, **kw):
        CAPTURE.clear()
        CAPTURE.update(kw)
        CAPTURE["settings"] = self.s
        H, W = self.s.image_height, self.s.image_width
        z = lambda c: torch.zeros(c, H, W)
        n = kw["means3D"].shape[0]
        if "extra_attrs" in kw:
            return z(3), z(1), z(3), z(1), torch.zeros(n, dtype=torch.int32), None
        return z(3), torch.zeros(n, dtype=torch.int32), z(1), z(1)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_stub("plyfile", PlyData=None, PlyElement=None)
_stub("diff_gaussian_rasterization", GaussianRasterizationSettings=_Settings, GaussianRasterizer=_Rasterizer)
_stub("diff_gauss", GaussianRasterizationSettings=_Settings, GaussianRasterizer=_Rasterizer)
_stub("simple_knn")
_stub("simple_knn._C", distCUDA2=None)
_stub("open3d")
p3 = _stub("pytorch3d")
_stub("pytorch3d.transforms", quaternion_to_matrix=None)
p3.ops = _stub("pytorch3d.ops", ball_query=None, knn_points=None, sample_farthest_points=None)
_stub("pytorch3d.io", load_ply=None)
_stub("pytorch3d.loss")
_stub("pytorch3d.loss.mesh_laplacian_smoothing", cot_laplacian=None, laplacian=None)
sys.path.insert(0, REF)

from renderer import latent_gs_renderer as R  # noqa: E402
from renderer import gaussian_gs_renderer as RV  # noqa: E402
from src import loss as ref_loss  # noqa: E402
from src.pos_enc import get_embedder  # noqa: E402
from utils import cam_utils, sh_utils  # noqa: E402

from oracle import raster_oracle as ro  # noqa: E402  (only for knn indices fed to the reference)
from tests.scenes import GRAD_STRIDE, timenet_weights  # noqa: E402


def load_weights(net, seed, head_std):
    net.load_state_dict({k: torch.from_numpy(v) for k, v in timenet_weights(seed, head_std).items()})


def sg(t):  # strided sample of a big gradient
    return n(t).reshape(-1)[::GRAD_STRIDE]


ONLY = set(sys.argv[1:])  # e.g. `python tests/golden/make_golden.py densify_r.npz`: rewrite only these files


def save(name, **arrs):
    if ONLY and name not in ONLY:
        return
    np.savez_compressed(os.path.join(OUT, name), **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", name, len(arrs), "arrays")


def n(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- 1. positional encoding
g = torch.Generator().manual_seed(0)
x3 = torch.rand(16, 3, generator=g) - 0.5
t1 = torch.rand(16, 1, generator=g)
e3, d3 = get_embedder(10, 3)
e1, d1 = get_embedder(6, 1)
save("pos_enc.npz", x3=n(x3), t1=n(t1), emb3=n(e3(x3)), emb1=n(e1(t1)), dims=np.array([d3, d1]))

# ----------------------------------------------------------------------------- 2. TimeNet fwd + grads
torch.manual_seed(0)
net = R.TimeNet(latent_code_dim=32, device="cpu")
# the reference zero-inits the last pts layer; seeded non-trivial weights instead (tests/scenes.py)
load_weights(net, 100, 1e-2)
for M in (8, 512):
    pts = (torch.rand(M, 3, generator=g) - 0.5).requires_grad_(True)
    lat = torch.randn(32, generator=g).requires_grad_(True)
    tt = 7 / 21
    dp, dq = net(pts, tt, lat)
    wp, wq = torch.randn(M, 3, generator=g), torch.randn(M, 4, generator=g)
    net.zero_grad()
    ((dp * wp).sum() + (dq * wq).sum()).backward()
    extra = {f"grad.{k}": sg(p.grad) for k, p in net.named_parameters()}
    save(f"timenet_M{M}.npz", pts=n(pts), latent=n(lat), t=np.array(tt), dxyz=n(dp), dquat=n(dq), wp=n(wp), wq=n(wq),
         g_pts=n(pts.grad), g_latent=n(lat.grad), weight_seed=np.array(100), **extra)
# batched-time path used by the cache / ARAP (t_apply=True)
pts = torch.rand(1, 8, 3, generator=g) - 0.5
times = torch.rand(3, generator=g)[:, None, None].repeat(1, 8, 1)
dp, dq = net(pts, times, torch.randn(32, generator=torch.Generator().manual_seed(3)), t_apply=True)
save("timenet_tapply.npz", pts=n(pts), times=n(times), latent=n(torch.randn(32, generator=torch.Generator().manual_seed(3))),
     dxyz=n(dp), dquat=n(dq))

# ----------------------------------------------------------------------------- 3. helpers
q = torch.randn(5, 4, 4, generator=g)
save("quat_helpers.npz", q=n(q), R3d=n(R.build_rotation_3d(q)), q1=n(q[:, 0]), q2=n(q[:, 1]),
     qmul=n(R.quat_mul(q[:, 0], q[:, 1])), R=n(R.build_rotation(q[:, 2])))
xyzs, covs = torch.randn(6, 3, generator=g) * 0.1, torch.rand(6, 6, generator=g)
covs[:, [0, 3, 5]] += 2.0
save("gaussian_3d_coeff.npz", xyzs=n(xyzs), covs=n(covs), out=n(R.gaussian_3d_coeff(xyzs, covs)))
f = R.get_expon_lr_func(lr_init=0.01, lr_final=0.0002, lr_delay_mult=0.02, max_steps=1000)
f2 = R.get_expon_lr_func(lr_init=0.005, lr_final=0.0002, lr_delay_steps=100, lr_delay_mult=0.02, max_steps=1000)
steps = np.array([0, 1, 10, 100, 500, 999, 1000, 5000])
save("lr_func.npz", steps=steps, lr=np.array([f(s) for s in steps]), lr_delay=np.array([f2(s) for s in steps]))

# ----------------------------------------------------------------------------- 4. cameras
azis = [360 / 9 * i for i in range(9)]
oc = cam_utils.OrbitCamera(800, 800, r=2, fovy=33.9)
cams = {}
for res in (128, 256, 512):
    for i, az in enumerate(azis):
        pose = cam_utils.orbit_camera(0, az, 2)
        mc = R.MiniCam(pose, res, res, oc.fovy, oc.fovx, oc.near, oc.far)
        cams[f"pose_{i}"] = pose
        cams[f"wv_{i}"] = n(mc.world_view_transform)
        cams[f"fp_{i}"] = n(mc.full_proj_transform)
        cams[f"cc_{i}"] = n(mc.camera_center)
    cams[f"proj_{res}"] = n(mc.projection_matrix)
pose_e = cam_utils.orbit_camera(-20, 33, 1.5)
mc = R.MiniCam(pose_e, 64, 48, oc.fovy, 0.9, 0.05, 50)
save("cameras.npz", azimuths=np.array(azis), fovy=np.array(oc.fovy), fovx=np.array(oc.fovx), pose_e=pose_e,
     wv_e=n(mc.world_view_transform), fp_e=n(mc.full_proj_transform), cc_e=n(mc.camera_center),
     proj_e=n(mc.projection_matrix), **cams)

# ----------------------------------------------------------------------------- 5. image losses
img1 = torch.rand(2, 3, 40, 36, generator=g).requires_grad_(True)
img2 = torch.rand(2, 3, 40, 36, generator=g)
s = ref_loss.ssim(img1, img2)
s.backward()
depth = torch.rand(2, 40, 36, 1, generator=g).requires_grad_(True)
normal = torch.rand(2, 40, 36, 3, generator=g).requires_grad_(True)
rgb = torch.rand(2, 40, 36, 3, generator=g).requires_grad_(True)
ea = ref_loss.compute_edge_aware_smoothness_loss(depth, rgb)
bl = ref_loss.compute_bilateral_normal_smoothness_loss(normal, rgb)
(ea + bl).backward()
save("image_losses.npz", img1=n(img1), img2=n(img2), ssim=n(s), g_img1=n(img1.grad),
     ssim_per=n(ref_loss.ssim(img1, img2, size_average=False)), depth=n(depth), normal=n(normal), rgb=n(rgb),
     edge_aware=n(ea), bilateral=n(bl), g_depth=n(depth.grad), g_normal=n(normal.grad), g_rgb=n(rgb.grad))

# ----------------------------------------------------------------------------- 6. SH
xyz = torch.randn(64, 3, generator=g) * 0.3
campos = torch.tensor([0.1, -0.2, -2.0])
shs = torch.randn(64, 16, 3, generator=g) * 0.4
dirs = (xyz - campos) / (xyz - campos).norm(dim=1, keepdim=True)
out = {f"rgb_deg{d}": n(torch.clamp_min(sh_utils.eval_sh(d, shs.transpose(1, 2), dirs) + 0.5, 0.0)) for d in range(4)}
save("sh_eval.npz", xyz=n(xyz), campos=n(campos), shs=n(shs), rgb2sh=n(sh_utils.RGB2SH(torch.tensor([0.0, 0.3, 1.0]))),
     **out)

# ----------------------------------------------------------------------------- 7. Renderer.render deform stage
def deform_fixture(mod, name, vae):
    torch.manual_seed(1)
    np.random.seed(1)
    Ng, Mc, L = 300, 24, 5
    rd = mod.Renderer(sh_degree=0, white_background=True, radius=2, num_latent_code=L, latent_code_dim=32,
                      add_normal=True)
    gm = rd.gaussians
    P = lambda t: torch.nn.Parameter(t.clone().requires_grad_(True))
    gm._xyz = P((torch.rand(Ng, 3) - 0.5) * 0.8)
    gm._features_dc = P(torch.randn(Ng, 1, 3) * 0.3)
    gm._features_rest = P(torch.zeros(Ng, 0, 3))
    gm._scaling = P(torch.log(torch.rand(Ng, 3) * 0.03 + 0.005))
    gm._rotation = P(torch.randn(Ng, 4))
    gm._opacity = P(torch.randn(Ng, 1))
    gm._c_xyz = P((torch.rand(Mc, 3) - 0.5) * 0.8)
    gm._c_radius = P(torch.log(torch.rand(Mc, 1) * 0.1 + 0.05))
    gm._r = torch.empty(0)
    if vae:
        gm._mu = P(torch.randn(L, 32) * 0.5)
        gm._log_var = P(torch.randn(L, 32) * 0.1 - 2.0)
    else:
        gm._latent_codes = P(torch.randn(L, 32))
    load_weights(gm._timenet, 200, 5e-2)
    d, i = ro.knn(n(gm._c_xyz), n(gm._xyz), 4)
    gm.neighbor_dists, gm.neighbor_indices = torch.from_numpy(d), torch.from_numpy(i)
    pose = cam_utils.orbit_camera(0, 80.0, 2)
    cam = mod.MiniCam(pose, 64, 64, oc.fovy, oc.fovx, oc.near, oc.far)
    res = {}
    params = dict(xyz=gm._xyz, f_dc=gm._features_dc, scaling=gm._scaling, rotation=gm._rotation, opacity=gm._opacity,
                  c_xyz=gm._c_xyz, c_radius=gm._c_radius)
    if vae:
        params.update(mu=gm._mu, log_var=gm._log_var)
    else:
        params.update(latent_codes=gm._latent_codes)
    for k, p in params.items():
        res[f"param.{k}"] = n(p)
    res["weight_seed"], res["head_std"] = np.array(200), np.array(5e-2)
    res["knn_dist"], res["knn_idx"] = d, i
    for stage in ("s1", "s2"):
        torch.manual_seed(11)  # fixes the VAE eps draw: eps = randn_like(std) right after this seed
        out = rd.render(cam, time=5 / 21, stage=stage, latent_index=3)
        cap = dict(CAPTURE)
        wg = torch.Generator().manual_seed(7)
        L_ = 0
        for key in ("means3D", "opacities", "scales", "rotations", "shs"):
            w = torch.randn(cap[key].shape, generator=wg)
            res[f"{stage}.w.{key}"] = n(w)
            res[f"{stage}.in.{key}"] = n(cap[key])
            L_ = L_ + (cap[key] * w).sum()
        wc = torch.randn(out["cpts_t"].shape, generator=wg)
        res[f"{stage}.w.cpts_t"] = n(wc)
        res[f"{stage}.cpts_t"] = n(out["cpts_t"])
        res[f"{stage}.pts_t"] = n(out["pts_t"])
        L_ = L_ + (out["cpts_t"] * wc).sum()
        for p in list(params.values()) + list(gm._timenet.parameters()):
            p.grad = None
        L_.backward()
        for k, p in params.items():
            res[f"{stage}.grad.{k}"] = n(p.grad) if p.grad is not None else np.zeros(p.shape, np.float32)
        for k, p in gm._timenet.named_parameters():
            res[f"{stage}.grad.timenet.{k}"] = sg(p.grad) if p.grad is not None else sg(torch.zeros_like(p))
        s_ = cap["settings"]
        res[f"{stage}.settings"] = np.array([s_.image_height, s_.image_width, s_.tanfovx, s_.tanfovy, s_.scale_modifier,
                                             s_.sh_degree], np.float64)
        res[f"{stage}.bg"] = n(s_.bg)
    res["cam.pose"] = pose
    res["time"], res["latent_index"], res["vae_seed"] = np.array(5 / 21), np.array(3), np.array(11)
    save(name, **res)


deform_fixture(R, "deform_latent.npz", vae=False)
deform_fixture(RV, "deform_vae.npz", vae=True)
print("done")

# ----------------------------------------------------------------------------- 8. densify / prune / reset_opacity
# GaussianModel.densify_and_prune, prune and reset_opacity (renderer/latent_gs_renderer.py:571-574,652-924) on a small
# seeded model with non-trivial Adam moments.  The draws of densify_and_split come from the global torch generator
# (seeded right before the call): the product must consume it identically.
def densify_fixture():
    from dimo_amd.trainer import TrainConfig  # same field names as the reference's OmegaConf options
    opt = TrainConfig()
    torch.manual_seed(5)
    Ng, Mc, L = 400, 16, 3
    rd = R.Renderer(sh_degree=0, white_background=True, radius=2, num_latent_code=L, latent_code_dim=32, add_normal=True)
    gm = rd.gaussians
    P = lambda t: torch.nn.Parameter(t.clone().requires_grad_(True))
    gm._xyz = P((torch.rand(Ng, 3) - 0.5) * 0.8)
    gm._features_dc = P(torch.randn(Ng, 1, 3) * 0.3)
    gm._features_rest = P(torch.zeros(Ng, 0, 3))
    gm._scaling = P(torch.log(torch.rand(Ng, 3) * 0.12 + 0.004))   # some above, some below percent_dense * extent
    gm._rotation = P(torch.randn(Ng, 4))
    gm._opacity = P(torch.randn(Ng, 1) * 2.5)                       # some below the opacity threshold
    gm._c_xyz = P((torch.rand(Mc, 3) - 0.5) * 0.8)
    gm._c_radius = P(torch.log(torch.rand(Mc, 1) * 0.1 + 0.05))
    gm._r = torch.empty(0)
    gm._latent_codes = P(torch.randn(L, 32))
    gm.spatial_lr_scale = 1.0
    gm.max_radii2D = torch.zeros(Ng)
    gm.training_setup(opt)
    res = {}
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    per_gauss = dict(xyz=gm._xyz, f_dc=gm._features_dc, f_rest=gm._features_rest, opacity=gm._opacity,
                     scaling=gm._scaling, rotation=gm._rotation)
    for k, p in per_gauss.items():
        res[f"init.{k}"] = n(p).copy()  # .numpy() shares memory with the parameter the optimizer is about to move
    gg = torch.Generator().manual_seed(9)
    for it in range(2):  # two Adam steps so that every moment is non-trivial
        for k, p in per_gauss.items():
            p.grad = torch.randn(p.shape, generator=gg) * 0.01
            res[f"grad{it}.{k}"] = n(p.grad).copy()
        for grp in gm.optimizer.param_groups:  # the other groups get no gradient in this fixture
            if grp["name"] not in names:
                for q in grp["params"]:
                    q.grad = None
        gm.optimizer.step()
    accum, denom = torch.rand(Ng, 1, generator=gg) * 0.05, torch.randint(0, 3, (Ng, 1), generator=gg).float()
    gm.xyz_gradient_accum, gm.denom = accum.clone(), denom.clone()
    gm.max_radii2D = torch.rand(Ng, generator=gg) * 3
    res["accum"], res["denom"], res["max_radii2D"] = n(accum), n(denom), n(gm.max_radii2D)

    def snap(tag):
        for grp in gm.optimizer.param_groups:
            if grp["name"] in names:
                q = grp["params"][0]
                st = gm.optimizer.state.get(q, None)
                res[f"{tag}.{grp['name']}"] = n(q).copy()
                res[f"{tag}.exp_avg.{grp['name']}"] = n(st["exp_avg"]).copy()
                res[f"{tag}.exp_avg_sq.{grp['name']}"] = n(st["exp_avg_sq"]).copy()
                res[f"{tag}.step.{grp['name']}"] = np.array(float(st["step"]))
        res[f"{tag}.accum"], res[f"{tag}.denom"] = n(gm.xyz_gradient_accum).copy(), n(gm.denom).copy()
        res[f"{tag}.max_radii2D"] = n(gm.max_radii2D).copy()

    snap("before")
    torch.manual_seed(77)
    gm.densify_and_prune(opt.densify_grad_threshold, min_opacity=opt.densify_opacity_threshold_s1, extent=4,
                         max_screen_size=1)
    snap("densified")
    gm.max_radii2D = torch.rand(gm._xyz.shape[0], generator=gg) * 3
    res["prune.max_radii2D"] = n(gm.max_radii2D)
    gm.prune(min_opacity=0.3, extent=4, max_screen_size=1)
    snap("pruned")
    gm.reset_opacity()
    snap("reset")
    res["seed_split"] = np.array(77)
    res["thresholds"] = np.array([opt.densify_grad_threshold, opt.densify_opacity_threshold_s1, 4.0, 1.0, 0.3,
                                  opt.percent_dense])
    save("densify.npz", **res)


densify_fixture()
print("densify done")


# ----------------------------------------------------------------------------- 8b. stage-s1 densification
# The same entry points in the configuration stage s1 actually runs them in: `_r` is the shared (1, 1) log-radius that
# create_from_pcd makes, so get_scaling = exp(_r) for every Gaussian (renderer/latent_gs_renderer.py:341-351) and the
# clone / split selection, the split's sample stds, new_scaling and the world-size prune all use it instead of
# `_scaling`.  Two radii: above percent_dense * extent (every selected point is split) and below it (cloned).  Plus
# GUI.FPS's call `prune_points(idxs)` with an INDEX tensor (main_train_dimo.py:511-515), whose `~mask` is a bitwise not.
def densify_r_fixture():
    from dimo_amd.trainer import TrainConfig
    opt = TrainConfig()
    res = {}
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    for case, radius in (("split", 0.05), ("clone", 0.03)):
        torch.manual_seed(11)
        Ng, Mc, L = 300, 300, 3
        rd = R.Renderer(sh_degree=0, white_background=True, radius=2, num_latent_code=L, latent_code_dim=32,
                        add_normal=True)
        gm = rd.gaussians
        P = lambda t: torch.nn.Parameter(t.clone().requires_grad_(True))
        gm._xyz = P((torch.rand(Ng, 3) - 0.5) * 0.8)
        gm._features_dc = P(torch.randn(Ng, 1, 3) * 0.3)
        gm._features_rest = P(torch.zeros(Ng, 0, 3))
        gm._scaling = P(torch.log(torch.rand(Ng, 3) * 0.12 + 0.004))
        gm._rotation = P(torch.randn(Ng, 4))
        gm._opacity = P(torch.randn(Ng, 1) * 2.5)
        gm._c_xyz = P((torch.rand(Mc, 3) - 0.5) * 0.8)
        gm._c_radius = P(torch.log(torch.rand(Mc, 1) * 0.1 + 0.05))
        gm._r = P(torch.log(torch.tensor([[radius]])))
        gm._latent_codes = P(torch.randn(L, 32))
        gm.spatial_lr_scale = 1.0
        gm.max_radii2D = torch.zeros(Ng)
        gm.training_setup(opt)
        per_gauss = dict(xyz=gm._xyz, f_dc=gm._features_dc, f_rest=gm._features_rest, opacity=gm._opacity,
                         scaling=gm._scaling, rotation=gm._rotation)
        if case == "split":
            for k, p in per_gauss.items():
                res[f"init.{k}"] = n(p).copy()
        gg = torch.Generator().manual_seed(9)
        for it in range(2):
            for k, p in per_gauss.items():
                p.grad = torch.randn(p.shape, generator=gg) * 0.01
                if case == "split":
                    res[f"grad{it}.{k}"] = n(p.grad).copy()
            gm._r.grad = torch.full((1, 1), 0.02 * (it + 1))
            for grp in gm.optimizer.param_groups:
                if grp["name"] not in names + ("r",):
                    for q in grp["params"]:
                        q.grad = None
            gm.optimizer.step()
        accum, denom = torch.rand(Ng, 1, generator=gg) * 0.05, torch.randint(0, 3, (Ng, 1), generator=gg).float()
        gm.xyz_gradient_accum, gm.denom = accum.clone(), denom.clone()
        gm.max_radii2D = torch.rand(Ng, generator=gg) * 3
        if case == "split":
            res["accum"], res["denom"], res["max_radii2D"] = n(accum), n(denom), n(gm.max_radii2D)
        res[f"{case}.r_before"] = n(gm._r).copy()

        def snap(tag):
            for grp in gm.optimizer.param_groups:
                if grp["name"] in names + ("r",):
                    q = grp["params"][0]
                    st = gm.optimizer.state.get(q, None)
                    res[f"{tag}.{grp['name']}"] = n(q).copy()
                    res[f"{tag}.exp_avg.{grp['name']}"] = n(st["exp_avg"]).copy()
                    res[f"{tag}.exp_avg_sq.{grp['name']}"] = n(st["exp_avg_sq"]).copy()
                    res[f"{tag}.step.{grp['name']}"] = np.array(float(st["step"]))
            res[f"{tag}.accum"], res[f"{tag}.denom"] = n(gm.xyz_gradient_accum).copy(), n(gm.denom).copy()
            res[f"{tag}.max_radii2D"] = n(gm.max_radii2D).copy()

        torch.manual_seed(78)
        gm.densify_and_prune(opt.densify_grad_threshold, min_opacity=opt.densify_opacity_threshold_s1, extent=4,
                             max_screen_size=1)
        snap(f"{case}.densified")
        if case == "split":  # GUI.FPS hands prune_points an index tensor
            idxs = torch.tensor([5, 0, 17, 3, 40, 41, 2], dtype=torch.int64)
            gm.xyz_gradient_accum = torch.arange(gm._xyz.shape[0], dtype=torch.float32)[:, None].clone()
            gm.prune_points(idxs)
            res["fps.idxs"] = n(idxs)
            snap("fps")
    res["radii"] = np.array([0.05, 0.03])
    res["seed_split"] = np.array(78)
    res["thresholds"] = np.array([opt.densify_grad_threshold, opt.densify_opacity_threshold_s1, 4.0, 1.0,
                                  opt.percent_dense])
    save("densify_r.npz", **res)


densify_r_fixture()
print("densify_r done")



# ----------------------------------------------------------------------------- 9. ARAP regulariser
# utils/deform_utils.py: cal_connectivity_from_points_v2 (:115-141), estimate_rotation (:161-197), cal_arap_error
# (:208-236) on a small node sequence.  pytorch3d is absent: its ball_query is replaced by a restatement of its
# documented behaviour (the first K points of p2, in index order, with squared distance < radius^2; idx padded with
# -1, dists with 0) -- the only un-pinned piece of this fixture.
def arap_fixture():
    from utils import deform_utils as DU

    def ball_query(p1, p2, K, radius):
        T_, N1 = p1.shape[0], p1.shape[1]
        d2 = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
        idx = torch.full((T_, N1, K), -1, dtype=torch.long)
        dist = torch.zeros(T_, N1, K)
        for t in range(T_):
            for i in range(N1):
                hit = torch.nonzero(d2[t, i] < radius * radius).flatten()[:K]
                idx[t, i, :len(hit)] = hit
                dist[t, i, :len(hit)] = d2[t, i, hit]
        return dist, idx, None

    class _NN:  # ops.ball_query(...) returns a 3-tuple whose third item is sliced: hand back a sliceable dummy
        def __getitem__(self, k):
            return self

    DU.ops.ball_query = lambda p1, p2, K, radius: ball_query(p1, p2, K, radius)[:2] + (_NN(),)
    gg = torch.Generator().manual_seed(21)
    T_, Nv = 4, 60
    base = (torch.rand(Nv, 3, generator=gg) - 0.5) * 0.3
    pts = base[None] + 0.01 * torch.randn(T_, Nv, 3, generator=gg)
    # a rotation-like deformation on top, different per time
    for t in range(T_):
        a = 0.15 * t
        rot = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
        pts[t] = pts[t] @ rot.T
    pts = pts.clone().requires_grad_(True)
    ii, jj, nn_, _ = DU.cal_connectivity_from_points_v2(pts.detach(), K=10)
    err = DU.cal_arap_error(pts, ii, jj, nn_)
    err.backward()
    with torch.no_grad():
        rot1 = DU.estimate_rotation(pts[0].detach(), pts[1].detach(), ii, jj, nn_, K=10,
                                    weight=torch.zeros(Nv, 10).index_put_((ii, nn_), torch.tensor(1.0)))
    save("arap.npz", pts=n(pts).copy(), ii=n(ii).copy(), jj=n(jj).copy(), nn=n(nn_).copy(), error=n(err).copy(),
         g_pts=n(pts.grad).copy(), rot1=n(rot1).copy(), radius=np.array(0.1), K=np.array(10))


arap_fixture()
print("arap done")
