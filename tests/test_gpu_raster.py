"""GPU parity: HIP rasterizer (through the C ABI) vs the CPU oracle on identical seeded inputs.

Integer stages (radii, tile rects, tiles_touched, offsets, sort keys, sorted order, tile ranges)
and the per-Gaussian fp32 projection outputs must be BIT-EXACT; images and gradients must agree
within 1e-4 L1 (north-star tolerance).  n_contrib may differ only where expf rounding flips a
threshold test (bounded fraction).
"""
import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from tests.scenes import camera_np, random_scene

pytestmark = pytest.mark.gpu
L1_TOL = 1e-4


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _settings(cam, bg, deg, scale_mod=1.0):
    from dimo_amd.rasterizer import GaussianRasterizationSettings
    d = _dev()
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=d)
    return GaussianRasterizationSettings(cam["H"], cam["W"], cam["tanfovx"], cam["tanfovy"], f(bg), scale_mod,
                                         f(cam["view"]), f(cam["proj"]), deg, f(cam["campos"]), False, False)


def _run_hip(sc, cam, bg, deg, with_normal=True, grads=None, scale_mod=1.0, capacity=None):
    """Returns (outputs dict, state dict, grads dict)."""
    from dimo_amd import rasterizer as rz
    d = _dev()
    t = {k: (torch.tensor(np.asarray(v), dtype=torch.float32, device=d).requires_grad_(True) if v is not None else None)
         for k, v in sc.items()}
    settings = _settings(cam, bg, deg, scale_mod)
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    captured = {}
    orig = rz._Rasterize.forward

    out = rz._Rasterize.apply(t["means3D"], means2D, t.get("shs"), t.get("colors"), t["opacities"], t.get("scales"),
                              t.get("rotations"), t.get("cov3D"), settings, with_normal, capacity)
    if with_normal:
        color, depth, normal, alpha, radii = out
    else:
        color, depth, alpha, radii = out
        normal = None
    fn = color.grad_fn
    saved = fn.saved_tensors
    geom, bin_ws, img_ws = saved[-3], saved[-2], saved[-1]
    N = t["means3D"].shape[0]
    st = rz.inspect_state((geom, bin_ws, img_ws), N, cam["H"], cam["W"], fn.r_cap)
    R = int(st["total"][0].item())
    res = dict(color=color, depth=depth, normal=normal, alpha=alpha, radii=radii)
    g = {}
    if grads is not None:
        loss = (color * grads[0]).sum() + (depth * grads[1]).sum() + (alpha * grads[3]).sum()
        if with_normal:
            loss = loss + (normal * grads[2]).sum()
        loss.backward()
        g = {k: v.grad for k, v in t.items() if v is not None}
        g["means2D"] = means2D.grad
    torch.cuda.synchronize()
    return res, st, R, g


def _oracle(sc, cam, bg, deg, scale_mod=1.0):
    f = lambda k: None if sc.get(k) is None else np.asarray(sc[k], np.float32)
    return ro.forward(f("means3D"), f("shs"), f("colors"), f("opacities"), f("scales"), f("rotations"), f("cov3D"),
                      scale_mod, cam["view"], cam["proj"], cam["campos"], np.asarray(bg, np.float32), cam["tanfovx"],
                      cam["tanfovy"], cam["H"], cam["W"], deg, f64=False)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _check_forward(sc, cam, bg, deg, with_normal=True, scale_mod=1.0):
    res, st, R, _ = _run_hip(sc, cam, bg, deg, with_normal, scale_mod=scale_mod)
    o = _oracle(sc, cam, bg, deg, scale_mod)
    n = lambda x: x.detach().cpu().numpy()
    # ---- integer / per-Gaussian stages: bit-exact
    assert R == o["R"]
    assert np.array_equal(n(res["radii"]), o["radii"])
    assert np.array_equal(n(st["tiles_touched"]).view(np.uint32), o["tiles_touched"])
    assert np.array_equal(n(st["offsets"]).view(np.uint32), o["offsets"])
    assert np.array_equal(n(st["rect"]).astype(np.int32), o["rect"])
    sp = n(st["splat"])
    vis = o["radii"] > 0
    assert np.array_equal(_bits(sp[vis, 0:2]), _bits(o["xy"][vis])), "pixel means differ"
    assert np.array_equal(_bits(sp[vis, 2:5]), _bits(o["conic_op"][vis, :3])), "conics differ"
    assert np.array_equal(_bits(sp[vis, 9]), _bits(o["feat"][vis, 3])), "depths differ"
    assert np.array_equal(_bits(sp[vis, 6:9]), _bits(o["feat"][vis, 0:3])), "colours differ"
    assert np.array_equal(_bits(sp[vis, 10:13]), _bits(o["feat"][vis, 4:7])), "normals differ"
    # (the library places every instance straight into its sorted slot: there is no unsorted emission to compare.
    # It stores the 32 DEPTH bits of an instance's key, not the 64-bit (tile | depth) word: `inspect_state` rebuilds the
    # published keys from (tile ranges, depth bits) -- rasterizer.py -- so the tile half of this comparison is implied
    # by the `ranges` comparison two lines down; the depth bits and the order are compared for real)
    assert np.array_equal(n(st["keys_sorted"])[:R].view(np.uint64), o["keys_sorted"]), "sort keys differ"
    assert np.array_equal(n(st["vals_sorted"])[:R].view(np.uint32), o["vals_sorted"]), "sorted order differs"
    assert np.array_equal(n(st["ranges"]).view(np.uint32), o["ranges"])
    # the (supertile, depth bin) entries the projection counted per Gaussian (what the level-1 capacity check rests on;
    # found unverified by tools/mutate_emulated.py) against the tile rectangles
    import ctypes as C
    from dimo_amd import _lib
    lay = (C.c_size_t * 10)()
    assert _lib.lib().dimo_debug_bin_geom_layout(len(o["radii"]), cam["H"], cam["W"], lay) == 0
    ss, rc = int(lay[9]), n(st["rect"]).astype(np.int64)
    on = o["tiles_touched"] > 0
    ent = ((((rc[:, 2] - 1) >> ss) - (rc[:, 0] >> ss) + 1) * (((rc[:, 3] - 1) >> ss) - (rc[:, 1] >> ss) + 1))[on].sum()
    assert int(n(st["total"])[2]) == int(ent), "level-1 entry count"
    # ---- images
    H, W = cam["H"], cam["W"]
    nc = n(st["n_contrib"]).view(np.uint32)
    assert (nc != o["n_contrib"]).mean() <= 2e-3, "n_contrib mismatch beyond expf threshold flips"
    for k, ok in (("color", "out_color"), ("depth", "out_depth"), ("alpha", "out_alpha"), ("normal", "out_normal")):
        if res[k] is None:
            continue
        err = np.abs(n(res[k]) - o[ok]).mean()
        assert err <= L1_TOL, (k, err)
    assert np.abs(n(st["final_T"]) - o["final_T"]).mean() <= L1_TOL
    return res, st, o


@pytest.mark.parametrize("N,H,W,deg,M", [(1000, 128, 128, 0, 1), (5000, 80, 96, 0, 1), (3000, 128, 160, 3, 16),
                                         (2000, 64, 64, 1, 4), (20000, 256, 256, 0, 1),
                                         (1500, 50, 70, 0, 1)])  # (the last: partial tiles, a width not a multiple of 4)
def test_forward_parity(N, H, W, deg, M):
    cam = camera_np(25.0, elevation=8, W=W, H=H)
    sc = random_scene(N, seed=N, sh_coeffs=M, scale=0.02)
    _check_forward(sc, cam, (1.0, 1.0, 1.0), deg)


def test_forward_parity_four_output_flavour_and_scale_modifier():
    cam = camera_np(-70.0, W=128, H=96)
    sc = random_scene(4000, seed=11, scale=0.03)
    res, st, o = _check_forward(sc, cam, (0.2, 0.4, 0.6), 0, with_normal=False, scale_mod=0.7)
    assert res["normal"] is None


def test_forward_parity_precomputed_colour_and_cov():
    cam = camera_np(120.0, elevation=-20, W=96, H=96)
    sc = random_scene(1500, seed=5, scale=0.04)
    rng = np.random.default_rng(2)
    A = rng.standard_normal((1500, 3, 3)) * 0.03
    S = A @ A.transpose(0, 2, 1) + 1e-5 * np.eye(3)
    sc2 = dict(means3D=sc["means3D"], opacities=sc["opacities"], colors=rng.random((1500, 3)),
               cov3D=np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1))
    _check_forward(sc2, cam, (0, 0, 0), 0)


def test_edge_cases_empty_culled_offscreen():
    cam = camera_np(0.0, W=64, H=64)
    empty = dict(means3D=np.zeros((0, 3)), shs=np.zeros((0, 1, 3)), opacities=np.zeros((0, 1)),
                 scales=np.zeros((0, 3)), rotations=np.zeros((0, 4)))
    res, st, R, _ = _run_hip(empty, cam, (1, 1, 1), 0)
    assert R == 0 and torch.all(res["color"] == 1) and torch.all(res["alpha"] == 0)
    # all Gaussians behind the camera / outside the frustum
    sc = random_scene(100, seed=1)
    sc["means3D"][:, 2] += 5.0
    _check_forward(sc, cam, (0, 0, 0), 0)
    # duplicated depths -> ties resolved by emission order; huge Gaussians covering every tile
    sc = random_scene(300, seed=3, scale=0.6)
    sc["means3D"][100:200] = sc["means3D"][0:100]
    _check_forward(sc, cam, (0.5, 0.5, 0.5), 0)


@pytest.mark.parametrize("N", [2500, 9000])
def test_long_tile_lists_with_depth_ties(N):
    """Every Gaussian covers every tile (per-tile lists of N instances, multi-block sort) and half of them
    share their depth with another one: the order must still be the published stable order."""
    cam = camera_np(15.0, W=48, H=32)
    sc = random_scene(N, seed=N, scale=0.5, opacity=(0.01, 0.05))
    sc["means3D"] *= 0.2
    sc["means3D"][N // 2:] = sc["means3D"][: N - N // 2]  # ties in depth -> order decided by Gaussian id
    res, st, o = _check_forward(sc, cam, (0.3, 0.3, 0.3), 0)
    assert (o["ranges"][:, 1] - o["ranges"][:, 0]).max() >= 0.9 * N


@pytest.mark.parametrize("case", ["one_depth_1500", "one_depth_5000", "slab_and_outliers", "two_depths"])
def test_depth_sort_bucket_paths(case):
    """Level 1 drops the entries into (supertile, depth bin) buckets and bucket_sort ranks each bucket on its own
    (binning.hip): a workgroup in LDS up to 2048 entries, slices of a larger bucket by several workgroups, eight byte
    passes of a single workgroup when hundreds of entries share nearly one depth; a bucket of more than 4096 entries
    spills into the overflow area of the unsorted array.  Scenes that drive those paths -- thousands of Gaussians at ONE
    depth (order = index), a thin slab next to far outliers, two depths only -- must give the oracle's keys and order
    bit for bit.  (The same scenes run on the CPU SIMT emulation in tests/test_binning_emulated.py.)"""
    cam = camera_np(0.0, W=96, H=64)
    if case.startswith("one_depth"):
        N = int(case.split("_")[-1])
        sc = random_scene(N, seed=N, scale=0.02)
        sc["means3D"][:] = sc["means3D"][0]          # one position: one depth, one bucket of N entries
    elif case == "slab_and_outliers":
        N = 7000
        sc = random_scene(N, seed=5, scale=0.02)
        view = np.asarray(cam["view"], np.float64)   # row-vector convention: p_view = [p 1] @ view
        axis = view[:3, 2] / np.linalg.norm(view[:3, 2])
        p = sc["means3D"].astype(np.float64)
        p -= np.outer(p @ axis, axis) * (1.0 - 1e-5)  # squeeze the cloud to a slab 1e-5 of its depth extent
        sc["means3D"][:] = p.astype(np.float32)
        sc["means3D"][:6] += (np.arange(6)[:, None] * 0.6 - 1.5) * axis.astype(np.float32)  # far outliers set the range
    else:
        N = 6000
        sc = random_scene(N, seed=9, scale=0.02)
        sc["means3D"][: N // 2] = sc["means3D"][0]
        sc["means3D"][N // 2:] = sc["means3D"][N // 2]
    _check_forward(sc, cam, (0.2, 0.2, 0.2), 0)


@pytest.mark.parametrize("far", [30.0, 1.0e5, -1.0])
def test_depth_sort_with_floaters(far):
    """A few floaters far behind (or in front of) the scene must not decide the depth bins: the binned range brackets
    the bulk (largest block minimum / smallest block maximum), keys outside it fall into the end bins, and a bin that
    still overflows is cut into slices sorted by extra workgroups (binning.hip).  Keys and order stay the oracle's."""
    cam = camera_np(0.0, W=128, H=96)
    sc = random_scene(60_000, seed=7, scale=0.006, opacity=(0.05, 0.4))
    view = np.asarray(cam["view"], np.float64)
    axis = (view[:3, 2] / np.linalg.norm(view[:3, 2])).astype(np.float32)
    sc["means3D"][:8] += np.float32(far) * axis
    _check_forward(sc, cam, (0.1, 0.1, 0.1), 0)


def test_depth_sort_with_256_bins_above_400k_gaussians():
    """Above 400 000 Gaussians the depth sort switches from 128 to 256 coarse bins (binning.hip: depth_bins_log2)."""
    cam = camera_np(10.0, W=128, H=96)
    sc = random_scene(420_000, seed=42, scale=0.004, opacity=(0.05, 0.3))
    _check_forward(sc, cam, (0.1, 0.1, 0.1), 0)


@pytest.mark.parametrize("H,W", [(1040, 2048), (1296, 1296), (2048, 2048)])  # (the last: the largest image allowed, 16384 tiles)
def test_images_with_more_than_4096_tiles(H, W):
    """8 x 8-tile supertiles (ss_shift 3: 64-bit tile masks, 16-byte sorted entries) and more than 4096 tiles: the
    fill keeps every tile's first slot in dynamically sized LDS (33 KB at 8320 tiles, 64 KB at 16384)."""
    cam = camera_np(15.0, elevation=-5, W=W, H=H)
    sc = random_scene(6000, seed=77, scale=0.02)
    _check_forward(sc, cam, (0.0, 0.0, 0.0), 0)


def test_more_than_two_million_gaussians_are_refused():
    """A level-1 workgroup walks at most 8 blocks of 256 Gaussians and a render has at most 1024 of them (binning.hip:
    MAX_L1_PER, MAX_SEG): beyond 2 097 152 Gaussians per render the binning returns DIMO_E_ARG, no launch."""
    cam = camera_np(0.0, W=64, H=64)
    sc = random_scene(2_097_153 + 255, seed=1, scale=0.001)
    with pytest.raises(RuntimeError):
        _run_hip(sc, cam, (0, 0, 0), 0)


def test_image_beyond_the_supertile_grid_is_refused():
    """More than 256 supertiles of 8 x 8 tiles (16384 tiles, e.g. above 2048^2 pixels): DIMO_E_ARG, no launch."""
    cam = camera_np(0.0, W=2064, H=2064)
    sc = random_scene(100, seed=1)
    with pytest.raises(RuntimeError):
        _run_hip(sc, cam, (0, 0, 0), 0)


def _rel_l1(a, b):
    return np.abs(a - b).sum() / (np.abs(b).sum() + 1e-12)


def _check_backward(sc, cam, bg, deg, names, with_normal=True, seed=0):
    H, W = cam["H"], cam["W"]
    rng = np.random.default_rng(seed)
    gw = [rng.standard_normal(s).astype(np.float32) for s in ((3, H, W), (1, H, W), (3, H, W), (1, H, W))]
    if not with_normal:
        gw[2] = np.zeros((3, H, W), np.float32)
    d = _dev()
    res, st, R, g = _run_hip(sc, cam, bg, deg, with_normal, grads=[torch.tensor(x, device=d) for x in gw])
    o = _oracle(sc, cam, bg, deg)
    go = ro.backward(o, *gw)
    n = lambda x: x.detach().cpu().numpy()
    for k, gk in names.items():
        a, b = n(g[k]).reshape(-1), go[gk].reshape(-1)
        if k == "means2D":
            a = n(g[k])[:, :2].reshape(-1)
        err = _rel_l1(a, b)
        assert err <= L1_TOL, (k, err)
        assert np.isfinite(a).all()


GRADS_SH = dict(means3D="dL_dmeans3D", means2D="dL_dmean2D", shs="dL_dshs", opacities="dL_dopacity",
                scales="dL_dscales", rotations="dL_drot")


@pytest.mark.parametrize("N,H,W,deg,M", [(1000, 128, 128, 0, 1), (4000, 80, 96, 3, 16), (20000, 256, 256, 0, 1),
                                         (1500, 50, 70, 1, 4)])
def test_backward_parity(N, H, W, deg, M):
    cam = camera_np(25.0, elevation=8, W=W, H=H)
    sc = random_scene(N, seed=N + 1, sh_coeffs=M, scale=0.02)
    _check_backward(sc, cam, (1.0, 1.0, 1.0), deg, GRADS_SH)


def test_backward_view_direction_gradient_of_the_sh_colours():
    """tests/test_raster_emulated.py's case on the GPU: large degree-2 / degree-3 SH bands and a loss on the colour image
    only, so that dL/dmeans3D is carried by the colour's VIEW-DIRECTION part (a 1 % error in one of its degree-3 terms
    stayed under the 1e-4 bar of the ordinary scenes: tools/mutate_emulated.py)."""
    cam = camera_np(25.0, elevation=8, W=64, H=64, radius=1.2)
    sc = random_scene(1200, seed=31, sh_coeffs=16, scale=0.03)
    sc["shs"][:, 4:] *= 12.0
    sc["shs"][:, 0] += 80.0
    H, W = cam["H"], cam["W"]
    rng = np.random.default_rng(3)
    gw = [rng.standard_normal((3, H, W)).astype(np.float32), np.zeros((1, H, W), np.float32),
          np.zeros((3, H, W), np.float32), np.zeros((1, H, W), np.float32)]
    d = _dev()
    _, _, _, g = _run_hip(sc, cam, (0.0, 0.0, 0.0), 3, True, grads=[torch.tensor(x, device=d) for x in gw])
    o = _oracle(sc, cam, (0.0, 0.0, 0.0), 3)
    go = ro.backward(o, *gw)
    for k, gk in (("means3D", "dL_dmeans3D"), ("shs", "dL_dshs")):
        err = _rel_l1(g[k].detach().cpu().numpy().reshape(-1), go[gk].reshape(-1))
        assert err <= L1_TOL, (k, err)


def test_backward_gaussians_over_more_than_64_tiles():
    """Two ways from the blend backward's gradient records to a Gaussian's sums: up to 64 tile instances by its 64-bit hit
    mask (one atomic OR per record), more by the record flags and a whole wave (preprocess.hip).  A scene that has both
    kinds -- 256 tiles, a third of the Gaussians over most of them -- must give the oracle's gradients."""
    cam = camera_np(20.0, W=256, H=256)
    sc = random_scene(900, seed=5, scale=0.02)
    sc["scales"][::3] *= 10.0
    o = _oracle(sc, cam, (0.2, 0.2, 0.2), 0)
    assert (o["tiles_touched"] > 64).sum() > 100 and ((o["tiles_touched"] > 0) & (o["tiles_touched"] <= 64)).sum() > 100
    _check_backward(sc, cam, (0.2, 0.2, 0.2), 0, GRADS_SH)


def test_backward_parity_four_output_flavour():
    cam = camera_np(200.0, W=96, H=128)
    sc = random_scene(3000, seed=21, scale=0.03)
    _check_backward(sc, cam, (0.1, 0.2, 0.3), 0, GRADS_SH, with_normal=False)


def test_backward_parity_precomp():
    cam = camera_np(120.0, elevation=-20, W=96, H=96)
    sc = random_scene(1500, seed=5, scale=0.04)
    rng = np.random.default_rng(2)
    A = rng.standard_normal((1500, 3, 3)) * 0.03
    S = A @ A.transpose(0, 2, 1) + 1e-5 * np.eye(3)
    sc2 = dict(means3D=sc["means3D"], opacities=sc["opacities"], colors=rng.random((1500, 3)),
               cov3D=np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1))
    _check_backward(sc2, cam, (0, 0, 0), 0, dict(means3D="dL_dmeans3D", colors="dL_dcolors",
                                                   opacities="dL_dopacity", cov3D="dL_dcov3D"))


@pytest.mark.parametrize("seed", range(16))
def test_raster_fuzz_small_and_ragged_shapes(seed):
    """The seeded shapes of tests/test_raster_emulated.py's fuzz (one to a few Gaussians, images of a single row /
    column / tile, splats from specks to larger than the image, cameras inside the cloud, opacities down to 0, both
    flavours, SH degrees 0-3), here on the GPU: forward and backward against the oracle."""
    from tests.test_raster_emulated import _fuzz_case
    sc, cam, bg, deg, with_normal = _fuzz_case(seed)
    _, _, o = _check_forward(sc, cam, bg, deg, with_normal)
    if o["R"] > 0:
        _check_backward(sc, cam, bg, deg, GRADS_SH, with_normal, seed=seed)


def test_capacity_policy_async_and_overflow():
    from dimo_amd.rasterizer import CapacityPolicy
    cam = camera_np(10.0, W=128, H=128)
    sc = random_scene(3000, seed=9, scale=0.03)
    exact, st_e, R, _ = _run_hip(sc, cam, (1, 1, 1), 0)
    pol = CapacityPolicy(initial=2 * R)
    res, st, R2, _ = _run_hip(sc, cam, (1, 1, 1), 0, capacity=pol)
    assert R2 == R and pol.check()
    assert torch.equal(res["color"], exact["color"]) and torch.equal(res["alpha"], exact["alpha"])
    small = CapacityPolicy(initial=R // 2)
    _run_hip(sc, cam, (1, 1, 1), 0, capacity=small)  # must not crash or write out of bounds
    assert small.check() is False and small.capacity >= R
    res3, _, _, _ = _run_hip(sc, cam, (1, 1, 1), 0, capacity=small)
    assert small.check() and torch.equal(res3["color"], exact["color"])


def test_debug_flag_is_a_checked_mode():
    """`raster_settings.debug` (latent_gs_renderer.py:1145): errors surface as RuntimeError at the stage that caused
    them -- a truncated render (capacity too small) and non-finite outputs -- and never as an abort."""
    from dimo_amd import rasterizer as rz
    from dimo_amd.rasterizer import CapacityPolicy
    d = _dev()
    cam = camera_np(10.0, W=64, H=64)
    sc = random_scene(500, seed=4, scale=0.05)
    t = {k: torch.tensor(np.asarray(v), dtype=torch.float32, device=d) for k, v in sc.items()}
    st = _settings(cam, (0, 0, 0), 0)._replace(debug=True)
    args = lambda tt: (tt["means3D"], torch.zeros_like(tt["means3D"]), tt["shs"], None, tt["opacities"], tt["scales"],
                       tt["rotations"], None, st, True)
    rz._Rasterize.apply(*args(t), None)  # clean inputs pass
    with pytest.raises(RuntimeError, match="capacity"):
        rz._Rasterize.apply(*args(t), CapacityPolicy(initial=16))
    colors = torch.rand(500, 3, device=d)
    colors[:50] = float("nan")  # (NaN SH coefficients would be clamped away by max(colour + 0.5, 0), as upstream)
    with pytest.raises(RuntimeError, match="non-finite"):
        rz._Rasterize.apply(t["means3D"], torch.zeros_like(t["means3D"]), None, colors, t["opacities"], t["scales"],
                            t["rotations"], None, st, True, None)


@pytest.mark.parametrize("N,R,scale", [(50_000, 256, 0.02), (100_000, 512, 0.012), (200_000, 1024, 0.008)])
def test_baseline_config_sizes_against_the_oracle(N, R, scale):
    """BASELINE.json's C2 / C3 / C5 shapes (50k @ 256^2, 100k @ 512^2, 200k @ 1024^2): the C oracle finishes these in
    seconds, so the full comparison -- bit-exact integer stages, 1e-4 L1 images and gradients -- runs at full size."""
    cam = camera_np(40.0, elevation=5, W=R, H=R)
    sc = random_scene(N, seed=N + 7, scale=scale, anisotropy=0.3)
    _check_forward(sc, cam, (0.0, 0.0, 0.0), 0)
    _check_backward(sc, cam, (0.0, 0.0, 0.0), 0, GRADS_SH)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("opacity", [(0.05, 0.05), (0.01, 0.1)], ids=["every_opacity_0.05", "opacity_0.01_to_0.1"])
def test_init_regime_at_c3_size_against_the_oracle(opacity):
    """SURVEY 8d's FIRST synthetic regime at C3 size: the reference creates every Gaussian at opacity 0.05
    (renderer/latent_gs_renderer.py:431) and stage s2 starts from 102 400 of them (:1038-1058).  Nothing saturates: a
    pixel's last contributing entry sits near the end of its tile's list (~900 entries deep against ~85 in the trained
    regime), so the forward writes a checkpoint at every bucket boundary of every tile and the backward runs ~15 buckets
    per tile from them -- the deep-chain path the trained scenes barely touch.  Same comparison as above: integer stages
    bit for bit, images and every gradient within 1e-4."""
    cam = camera_np(40.0, elevation=5, W=512, H=512)
    sc = random_scene(100_000, seed=100_007, scale=0.012, anisotropy=0.3, opacity=opacity)
    res, st, o = _check_forward(sc, cam, (0.0, 0.0, 0.0), 0)
    ranges = o["ranges"].astype(np.int64)
    mean_list = (ranges[:, 1] - ranges[:, 0]).mean()
    assert o["n_contrib"].mean() > 0.8 * mean_list > 500, "not the deep regime this test is for"
    _check_backward(sc, cam, (0.0, 0.0, 0.0), 0, GRADS_SH)


def test_full_size_properties_100k_512():
    """BASELINE config size (100k Gaussians, 512^2): size-independent properties instead of the oracle."""
    cam = camera_np(40.0, W=512, H=512)
    sc = random_scene(100_000, seed=100, scale=0.012, anisotropy=0.2)
    res, st, R, _ = _run_hip(sc, cam, (0, 0, 0), 0)
    n = lambda x: x.detach().cpu().numpy()
    keys = n(st["keys_sorted"])[:R].view(np.uint64)
    assert np.all(keys[1:] >= keys[:-1]), "keys not sorted"
    # the sorted multiset equals the multiset the per-Gaussian state implies (checksum of keys and of values)
    rect = n(st["rect"]).astype(np.int64)  # x0 y0 x1 y1 in tiles
    tiles = n(st["tiles_touched"]).view(np.uint32).astype(np.uint64)
    depth_bits = n(st["splat"])[:, 9].view(np.uint32).astype(np.uint64)
    w, h = rect[:, 2] - rect[:, 0], rect[:, 3] - rect[:, 1]
    tiles_x = 512 // 16
    # sum over the rect of (y * tiles_x + x) = tiles_x * w * sum(y) + h * sum(x)
    sum_y = h * rect[:, 1] + h * (h - 1) // 2
    sum_x = w * rect[:, 0] + w * (w - 1) // 2
    tile_id_sum = (tiles_x * w * sum_y + h * sum_x).astype(np.uint64)
    expect = ((tile_id_sum << np.uint64(32)) + tiles * depth_bits).sum(dtype=np.uint64)
    assert int(expect) == int(keys.sum(dtype=np.uint64))
    ids = np.arange(len(tiles), dtype=np.uint64)
    assert int((ids * tiles).sum(dtype=np.uint64)) == int(n(st["vals_sorted"])[:R].astype(np.uint64).sum())
    ranges = n(st["ranges"]).view(np.uint32).astype(np.int64)
    assert (ranges[:, 1] - ranges[:, 0]).sum() == R
    a = n(res["alpha"])
    T = n(st["final_T"])
    assert np.abs(a[0] - (1 - T)).max() < 1e-4  # alpha = 1 - final transmittance
    assert a.min() >= 0 and a.max() <= 1 + 1e-5 and T.min() >= 1e-4 - 1e-7
    # linearity in colour: doubling the SH DC term (with zero background) doubles C - 0.5*alpha ... use colours
    col = np.random.default_rng(0).random((100_000, 3))
    sc_c = dict(means3D=sc["means3D"], scales=sc["scales"], rotations=sc["rotations"], opacities=sc["opacities"],
                colors=col)
    r1, _, _, _ = _run_hip(sc_c, cam, (0, 0, 0), 0)
    sc_c["colors"] = 2 * col
    r2, _, _, _ = _run_hip(sc_c, cam, (0, 0, 0), 0)
    assert torch.allclose(2 * r1["color"], r2["color"], atol=1e-5)
    assert torch.equal(r1["alpha"], r2["alpha"])
