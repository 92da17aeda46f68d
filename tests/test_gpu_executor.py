"""GPU parity of the kernels the benchmark actually times: the BATCHED step executor (dimo_executor_forward_range /
dimo_executor_backward_launch_in_order: preprocess_fwd_batched, the batched scan / depth sort / placement,
blend_fwd_batched, blend_bwd_batched (one wave per item), preprocess_bwd_batched) at BASELINE.json's sizes, per
render against the C oracle (oracle/raster_ref.c):

  radii, tile rects, tiles_touched, offsets, (tile | depth) keys, sorted order, tile ranges      bit-exact
  colour / depth / normal / alpha images, final transmittance                                    <= 1e-4 L1
  per-Gaussian rasterizer gradients (before the skinning backward)                                <= 1e-4 rel-L1

Contract: renderer/latent_gs_renderer.py:1255-1266 (the diff_gauss call of Renderer.render).  The oracle is fed the
skinned Gaussians the executor's own skinning kernel produced (that kernel has its own oracle test,
tests/test_gpu_deform.py), so the integer stages can be compared bit for bit.
"""
import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro

pytestmark = pytest.mark.gpu
L1_TOL = 1e-4


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _rel_l1(a, b):
    return np.abs(a - b).sum() / (np.abs(b).sum() + 1e-12)


def _run_batched(N, res, renders_per_motion, n_motions, seed=0, streams="-2", monkeypatch=None, joint=False,
                 regime="trained"):
    """Drives the executor the way Trainer._forward_backward_direct does (one range per motion, each range in
    order on its private stream) with random TimeNet outputs and random gradient images."""
    from dimo_amd import _lib  # noqa: F401
    from dimo_amd.rasterizer import CapacityPolicy, inspect_state
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    if monkeypatch is not None:
        monkeypatch.setenv("DIMO_EXEC_STREAMS", streams)
    n = renders_per_motion * n_motions
    cfg = TrainConfig(num_pts=N, num_cpts=512, num_motions=max(4, n_motions), resolution=res,
                      motions_per_step=n_motions, views_per_step=2, frames_per_step=max(1, renders_per_motion // 2),
                      progressive_resolution=False)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                  capacity=CapacityPolicy(initial=max(1 << 20, 40 * N)))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=seed, num_latent=cfg.num_motions, regime=regime)
    tr = Trainer(cfg, rd)
    g = rd.gaussians
    tr.find_knn(4)
    ex = tr._executor(n)
    ex.set_common(g, rd.bg_color, True)
    H = W = res
    M = cfg.num_cpts
    gen = torch.Generator().manual_seed(seed + 1)
    # one (motion, frame) pair per two views, like the trainer's deformation groups
    P = (n + 1) // 2
    dxyz = (0.02 * torch.randn(P, M, 3, generator=gen)).cuda().contiguous()
    dquat = (torch.tensor([1.0, 0, 0, 0]) + 0.1 * torch.randn(P, M, 4, generator=gen)).cuda().contiguous()
    g_dxyz, g_dquat = torch.zeros_like(dxyz), torch.zeros_like(dquat)
    f32 = dict(dtype=torch.float32, device="cuda")
    img, depth = torch.empty(n, 3, H, W, **f32), torch.empty(n, 1, H, W, **f32)
    normal, alpha = torch.empty(n, 3, H, W, **f32), torch.empty(n, 1, H, W, **f32)
    grads = [(torch.randn(n, c, H, W, generator=gen)).cuda() for c in (3, 1, 3, 1)]
    cams, HW4 = [], H * W * 4
    for i in range(n):
        d = ex.descs[i]
        cam = tr.cams.get(0.0, tr.azimuths[(3 * i + (i // 2)) % len(tr.azimuths)], cfg.radius, W, H)
        cams.append(cam)
        d.view, d.proj, d.campos = (cam.world_view_transform.data_ptr(), cam.full_proj_transform.data_ptr(),
                                    cam.camera_center.data_ptr())
        d.tanfovx, d.tanfovy = cam.tanfovx, cam.tanfovy
        p = i // 2
        d.d_xyz, d.d_rot = dxyz.data_ptr() + p * M * 12, dquat.data_ptr() + p * M * 16
        d.g_d_xyz, d.g_d_rot = g_dxyz.data_ptr() + p * M * 12, g_dquat.data_ptr() + p * M * 16
        d.out_color, d.out_depth = img.data_ptr() + i * 3 * HW4, depth.data_ptr() + i * HW4
        d.out_normal, d.out_alpha = normal.data_ptr() + i * 3 * HW4, alpha.data_ptr() + i * HW4
        d.g_color, d.g_depth = grads[0].data_ptr() + i * 3 * HW4, grads[1].data_ptr() + i * HW4
        d.g_normal, d.g_alpha = grads[2].data_ptr() + i * 3 * HW4, grads[3].data_ptr() + i * HW4
    torch.cuda.synchronize()
    firsts = list(range(0, n, renders_per_motion))
    for first in firsts:
        ex.forward_range(first, renders_per_motion) if ex.ranged else None
    if not ex.ranged:
        ex.forward(n)
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().cpu().numpy()
    out = []
    for i in range(n):
        lead = ex.slots[i - (i % 2)]  # the deformation group's leader holds the skinned Gaussians
        s = ex.slots[i]
        out.append(dict(
            pts=cpu(lead["pts"]), rot=cpu(lead["rot"]), scales=cpu(lead["scales"]), opac=cpu(lead["opac"]),
            radii=cpu(s["radii"]), cam=cams[i], color=cpu(img[i]), depth=cpu(depth[i]), normal=cpu(normal[i]),
            alpha=cpu(alpha[i]),
            st={k: cpu(v) for k, v in inspect_state((s["geom"], s["bin"], s["img"]), N, H, W, ex.r_cap).items()}))
    # the optional per-pixel S plane (sum over the channels of gradient x rendered value, what dimo_image_loss emits):
    # handed to the second half of the renders, the first half lets the kernel form S from the final accumulators
    dot = ((grads[0] * img).sum(1, keepdim=True) + grads[1] * depth + (grads[2] * normal).sum(1, keepdim=True)
           + grads[3] * alpha).contiguous()
    for i in range(n // 2, n):
        ex.descs[i].g_dot = dot.data_ptr() + i * HW4
    torch.cuda.synchronize()
    if joint:  # the default schedule's backward: launches over up to 8 renders spanning the motions' ranges
        assert ex.ranged
        ex.backward_launch_joint(0, n)
    else:
        for first in firsts:
            if ex.ranged:
                ex.backward_launch_in_order(first, renders_per_motion)
            else:
                ex.backward_launch(first, renders_per_motion)
    torch.cuda.synchronize()
    for i in range(n):
        s = ex.slots[i]
        out[i]["g"] = {k: cpu(s["g_" + k]).copy() for k in ("means3D", "means2D", "shs", "opac", "scales", "rot")}
        out[i]["gw"] = [cpu(x[i]) for x in grads]
    bg = cpu(rd.bg_color)
    f_dc = cpu(g._features_dc)
    return out, f_dc, bg, ex


def _check_render(o_hip, f_dc, bg, H, W):
    cam = o_hip["cam"]
    n = lambda t: t.detach().cpu().numpy()
    o = ro.forward(o_hip["pts"], f_dc, None, o_hip["opac"], o_hip["scales"], o_hip["rot"], None, 1.0,
                   n(cam.world_view_transform), n(cam.full_proj_transform), n(cam.camera_center), bg, cam.tanfovx,
                   cam.tanfovy, H, W, 0, f64=False)
    st = o_hip["st"]
    R = int(st["total"][0])
    assert int(st["total"][1]) == 0, "instance capacity overflow in the test"
    assert R == o["R"]
    assert np.array_equal(o_hip["radii"], o["radii"])
    assert np.array_equal(st["tiles_touched"].view(np.uint32), o["tiles_touched"])
    assert np.array_equal(st["offsets"].view(np.uint32), o["offsets"])
    assert np.array_equal(st["rect"].astype(np.int32), o["rect"])
    vis = o["radii"] > 0
    sp = st["splat"]
    assert np.array_equal(_bits(sp[vis, 0:2]), _bits(o["xy"][vis])), "pixel means differ"
    assert np.array_equal(_bits(sp[vis, 2:5]), _bits(o["conic_op"][vis, :3])), "conics differ"
    assert np.array_equal(_bits(sp[vis, 9]), _bits(o["feat"][vis, 3])), "depths differ"
    assert np.array_equal(st["keys_sorted"][:R].view(np.uint64), o["keys_sorted"]), "sort keys differ"
    assert np.array_equal(st["vals_sorted"][:R].view(np.uint32), o["vals_sorted"]), "sorted order differs"
    assert np.array_equal(st["ranges"].view(np.uint32), o["ranges"])
    assert (st["n_contrib"].view(np.uint32) != o["n_contrib"]).mean() <= 2e-3
    for k, ok in (("color", "out_color"), ("depth", "out_depth"), ("alpha", "out_alpha"), ("normal", "out_normal")):
        err = np.abs(o_hip[k] - o[ok]).mean()
        assert err <= L1_TOL, (k, err)
    assert np.abs(st["final_T"] - o["final_T"]).mean() <= L1_TOL
    go = ro.backward(o, *o_hip["gw"])
    g = o_hip["g"]
    pairs = (("means3D", go["dL_dmeans3D"]), ("shs", go["dL_dshs"]), ("opac", go["dL_dopacity"]),
             ("scales", go["dL_dscales"]), ("rot", go["dL_drot"]))
    for k, want in pairs:
        err = _rel_l1(g[k].reshape(-1), want.reshape(-1))
        assert err <= L1_TOL, (k, err)
        assert np.isfinite(g[k]).all()
    err = _rel_l1(g["means2D"][:, :2].reshape(-1), go["dL_dmean2D"].reshape(-1))
    assert err <= L1_TOL, ("means2D", err)
    return R


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("N,res,per_motion,motions", [
    (100_000, 512, 4, 2),    # C3: the benchmark workload -- 8 renders, blend_bwd_batched<true, 4> over 4 renders
    (50_000, 256, 4, 1),     # C2
    (200_000, 1024, 2, 1),   # C5's shape: 2 renders
])
def test_batched_executor_kernels_against_the_oracle(N, res, per_motion, motions, monkeypatch):
    outs, f_dc, bg, ex = _run_batched(N, res, per_motion, motions, seed=N % 97, monkeypatch=monkeypatch)
    assert ex.batched and ex.ranged
    Rs = [_check_render(o, f_dc, bg, res, res) for o in outs]
    assert min(Rs) > 5 * N  # a dense workload: every Gaussian lands in several tiles


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("N,res,per_motion,motions", [
    (100_000, 512, 4, 2),   # C3 as bench.py times it: ONE blend_bwd_batched launch over the step's 8 renders
    (30_000, 256, 6, 2),    # 12 renders: ranges of 6 do not merge into a launch of <= 8 -> two launches of 6
    (30_000, 256, 10, 1),   # a range longer than a launch: cut 8 + 2 exactly like its forward
])
def test_joint_backward_launch_against_the_oracle(N, res, per_motion, motions, monkeypatch):
    """dimo_executor_backward_launch_joint (main_train_dimo.py:415: the ONE backward over all the step's renders) --
    the launch bench.py's roofline is quoted on -- per render against the C oracle."""
    outs, f_dc, bg, ex = _run_batched(N, res, per_motion, motions, seed=N % 89, monkeypatch=monkeypatch, joint=True)
    assert ex.batched and ex.ranged
    for o in outs:
        _check_render(o, f_dc, bg, res, res)


@pytest.mark.timeout(3600)
@pytest.mark.parametrize("regime,joint", [("init", False), ("init", True), ("low", True)])
def test_init_regime_batched_kernels_at_c3_against_the_oracle(regime, joint, monkeypatch):
    """The benchmark's 8-render C3 batch in SURVEY 8d's "init" regime (every opacity 0.05, the state the reference's
    stage s2 starts from: renderer/latent_gs_renderer.py:431,1038-1058) and right behind it (U(0.01, 0.1)): every pixel
    looks through its tile's whole list, the forward checkpoints every bucket of every tile and the backward's deep
    queue holds ~13 of every 15 items.  Per-motion launches (the timed default) and the joint launch (the roofline's),
    per render against the C oracle, same bars as the trained regime."""
    outs, f_dc, bg, ex = _run_batched(100_000, 512, 4, 2, seed=11, monkeypatch=monkeypatch, joint=joint, regime=regime)
    assert ex.batched and ex.ranged
    deep = []
    for o in outs:
        _check_render(o, f_dc, bg, 512, 512)
        rg = o["st"]["ranges"].view(np.uint32).astype(np.int64)
        deep.append(o["st"]["n_contrib"].view(np.uint32).mean() / max((rg[:, 1] - rg[:, 0]).mean(), 1))
    assert min(deep) > 0.8, deep  # the deep regime: a pixel's last entry sits near the end of its list


def test_batched_executor_single_stream_mode_small(monkeypatch):
    """Fully batched mode on the caller's stream (DIMO_EXEC_STREAMS=0): all 8 renders in ONE launch per stage."""
    outs, f_dc, bg, ex = _run_batched(20_000, 128, 8, 1, seed=5, streams="0", monkeypatch=monkeypatch)
    assert ex.batched and not ex.ranged
    for o in outs:
        _check_render(o, f_dc, bg, 128, 128)
