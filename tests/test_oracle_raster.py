"""Pins the CPU oracle of the rasterizer (oracle/raster_ref.c).

The reference holds no tests or golden vectors for this boundary,
so the oracle is pinned by closed-form known-answer cases and by float64 finite
differences of its own backward.
"""
import math

import numpy as np
import pytest

from oracle import raster_oracle as ro
from tests.scenes import camera_np, random_scene


def _fwd(sc, cam, bg=(0, 0, 0), deg=0, f64=False, **kw):
    return ro.forward(sc["means3D"], sc.get("shs"), sc.get("colors"), sc["opacities"], sc.get("scales"),
                      sc.get("rotations"), sc.get("cov3D"), kw.get("scale_mod", 1.0), cam["view"], cam["proj"],
                      cam["campos"], np.asarray(bg, float), cam["tanfovx"], cam["tanfovy"], cam["H"], cam["W"], deg,
                      f64=f64)


def _one(xyz, s=0.02, o=0.8, col=(1.0, 0.5, 0.25)):
    xyz = np.atleast_2d(np.asarray(xyz, float))
    n = xyz.shape[0]
    return dict(means3D=xyz, scales=np.full((n, 3), s), rotations=np.tile([1.0, 0, 0, 0], (n, 1)),
                opacities=np.full((n, 1), o), colors=np.tile(np.asarray(col, float), (n, 1)))


def test_single_gaussian_closed_form():
    W = H = 128
    cam = camera_np(0.0, W=W, H=H)
    s, o = 0.02, 0.8
    st = _fwd(_one([0, 0, 0], s, o), cam, f64=True)
    fx = W / (2 * cam["tanfovx"])
    var = (fx * s / 2.0) ** 2 + 0.3  # depth = orbit radius 2
    assert st["radii"][0] == math.ceil(3 * math.sqrt(var))
    np.testing.assert_allclose(st["xy"][0], [(W - 1) / 2, (H - 1) / 2], atol=1e-9)
    np.testing.assert_allclose(st["feat"][0, 3], 2.0, atol=1e-12)
    a = o * math.exp(-0.5 * (0.25 + 0.25) / var)
    assert abs(st["out_alpha"][0, 63, 63] - a) < 1e-12
    np.testing.assert_allclose(st["out_color"][:, 64, 64], np.array([1.0, 0.5, 0.25]) * a, atol=1e-12)
    assert abs(st["out_depth"][0, 63, 64] - 2.0 * a) < 1e-12
    # isotropic + identity quaternion: axis 0 (ties -> lowest), flipped towards the camera, in view space
    n = st["out_normal"][:, 63, 63] / a
    assert abs(np.linalg.norm(n) - 1) < 1e-9
    # 16x16 footprint check: radius r around centre 63.5 -> tiles
    r = st["radii"][0]
    x0, x1 = int((63.5 - r) / 16), int((63.5 + r + 15) / 16)
    assert list(st["rect"][0]) == [x0, x0, x1, x1]
    assert st["tiles_touched"][0] == (x1 - x0) ** 2 == st["R"]


def test_two_gaussians_order_dependence():
    cam = camera_np(0.0, W=64, H=64)
    bg = np.array([0.1, 0.2, 0.3])
    # camera sits at +z (campos sign quirk aside); nearer Gaussian = larger world z
    sc = _one([[0, 0, 0.2], [0, 0, -0.2]], s=0.05, o=0.6)
    sc["colors"] = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    st = _fwd(sc, cam, bg=bg, f64=True)
    d = st["feat"][:, 3]
    near, far = (0, 1) if d[0] < d[1] else (1, 0)
    assert list(st["vals_sorted"][:2]) in ([near, far],) or st["vals_sorted"][0] == near
    px = (31, 31)
    al = []
    for g in (near, far):
        dx, dy = st["xy"][g] - np.array([px[1], px[0]])
        A, B, C, o = st["conic_op"][g]
        al.append(min(0.99, o * math.exp(-0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy)))
    exp = sc["colors"][near] * al[0] + sc["colors"][far] * al[1] * (1 - al[0]) + bg * (1 - al[0]) * (1 - al[1])
    np.testing.assert_allclose(st["out_color"][:, px[0], px[1]], exp, atol=1e-12)
    np.testing.assert_allclose(st["out_alpha"][0, px[0], px[1]], 1 - (1 - al[0]) * (1 - al[1]), atol=1e-12)
    assert st["n_contrib"][px] == 2


def test_near_plane_cull_and_offscreen():
    cam = camera_np(0.0, W=64, H=64)
    # camera at distance 2 on +z looking at origin: world z=1.9 -> view depth 0.1 (< 0.2) ; z=3 is behind
    st = _fwd(_one([[0, 0, 1.9], [0, 0, 3.0], [5.0, 0, 0], [0, 0, 0]]), cam)
    assert list(st["radii"][:3]) == [0, 0, 0] and st["radii"][3] > 0
    assert list(st["tiles_touched"][:3]) == [0, 0, 0]
    assert st["R"] == st["tiles_touched"][3]


def test_tile_corner_touches_four():
    W = H = 64
    cam = camera_np(0.0, W=W, H=H)
    # centre of the image (31.5, 31.5) is next to the corner of 4 tiles at (32,32); tiny Gaussian:
    # var = 0.3 (low-pass only), lambda_max = 0.3 + sqrt(max(0.1, 0)) -> radius = ceil(3 sqrt(0.616)) = 3
    st = _fwd(_one([0, 0, 0], s=0.0005), cam)
    assert st["radii"][0] == 3
    assert st["tiles_touched"][0] == 4
    assert sorted((st["keys_sorted"] >> np.uint64(32)).tolist()) == [5, 6, 9, 10]
    # ranges are contiguous and cover R
    assert st["ranges"][5].tolist() == [0, 1] and st["ranges"][10].tolist() == [3, 4]


def test_saturating_stack_early_termination():
    cam = camera_np(0.0, W=32, H=32)
    n = 40
    z = np.linspace(0.4, -0.4, n)
    sc = _one(np.stack([np.zeros(n), np.zeros(n), z], 1), s=0.2, o=0.95)
    st = _fwd(sc, cam, f64=True)
    nc = st["n_contrib"][15, 15]
    # alpha ~0.95 each: T after k = 0.05^k ; stops before T < 1e-4 -> 3 contributors
    assert nc == 3
    assert st["final_T"][15, 15] >= 1e-4
    assert st["final_T"][15, 15] * 0.05 < 1.1e-4
    # sorted by depth: nearest (largest world z) first
    tile0 = st["vals_sorted"][st["ranges"][0, 0]:st["ranges"][0, 1]]
    assert np.all(np.diff(st["feat"][tile0, 3]) >= 0)


def test_keys_sorted_stable_and_ranges():
    cam = camera_np(40.0, W=96, H=80)
    sc = random_scene(500, seed=3)
    # duplicate depths to exercise tie-breaking by emission order
    sc["means3D"][100:200] = sc["means3D"][0:100]
    st = _fwd(sc, cam)
    ks, vs = st["keys_sorted"], st["vals_sorted"]
    assert np.all(ks[1:] >= ks[:-1])
    same = ks[1:] == ks[:-1]
    assert same.any()
    assert np.all(vs[1:][same] > vs[:-1][same])  # ties keep Gaussian-id order
    assert st["offsets"][-1] == st["R"] == st["tiles_touched"].sum()
    gx = (96 + 15) // 16
    tiles = (ks >> np.uint64(32)).astype(np.int64)
    for t in range(st["ranges"].shape[0]):
        lo, hi = st["ranges"][t]
        assert np.all(tiles[lo:hi] == t)
    assert (st["ranges"][:, 1] - st["ranges"][:, 0]).sum() == st["R"]
    # ragged image: last tile row/column partially outside
    assert st["ranges"].shape[0] == gx * ((80 + 15) // 16)


def test_empty_scene():
    cam = camera_np(0.0, W=32, H=32)
    sc = _one(np.zeros((0, 3)))
    st = _fwd(sc, cam, bg=(1, 1, 1))
    assert st["R"] == 0
    assert np.all(st["out_color"] == 1) and np.all(st["out_alpha"] == 0) and np.all(st["n_contrib"] == 0)
    g = ro.backward(st, np.ones((3, 32, 32)), np.ones((1, 32, 32)), np.ones((3, 32, 32)), np.ones((1, 32, 32)))
    assert g["dL_dmeans3D"].shape == (0, 3)


def _fd_check(sc, cam, deg, names, n_samples=24, eps=1e-6, seed=5):
    H, W = cam["H"], cam["W"]
    rng = np.random.default_rng(seed)
    ws = [rng.standard_normal(s) for s in ((3, H, W), (1, H, W), (3, H, W), (1, H, W))]
    bg = np.array([0.3, 0.6, 0.9])

    def loss(s):
        st = _fwd(s, cam, bg=bg, deg=deg, f64=True)
        return sum((st[k] * w).sum() for k, w in zip(("out_color", "out_depth", "out_normal", "out_alpha"), ws)), st

    _, st = loss(sc)
    g = ro.backward(st, *ws)
    for k, gk in names.items():
        G = g[gk].reshape(sc[k].shape)
        errs = []
        for _ in range(n_samples):
            ix = tuple(rng.integers(0, s) for s in sc[k].shape)
            s2 = {a: (b.copy() if b is not None else None) for a, b in sc.items()}
            s2[k][ix] += eps
            lp, _ = loss(s2)
            s2[k][ix] -= 2 * eps
            lm, _ = loss(s2)
            fd = (lp - lm) / (2 * eps)
            errs.append(abs(fd - G[ix]) / (abs(fd) + abs(G[ix]) + 1e-6))
        errs = np.array(errs)
        # the rasterizer is piecewise smooth (3-sigma rect, alpha<1/255, T<1e-4 cut-offs): allow rare jumps
        assert np.median(errs) < 1e-6, (k, errs)
        assert (errs < 1e-4).mean() >= 0.85, (k, errs)


@pytest.mark.parametrize("deg,M", [(0, 1), (1, 4), (3, 16)])
def test_backward_finite_differences_sh(deg, M):
    cam = camera_np(30.0, elevation=10, W=64, H=64)
    sc = random_scene(60, seed=2, sh_coeffs=M, scale=0.05)
    _fd_check(sc, cam, deg, dict(means3D="dL_dmeans3D", shs="dL_dshs", opacities="dL_dopacity",
                                 scales="dL_dscales", rotations="dL_drot"))


def test_backward_finite_differences_precomp():
    cam = camera_np(-50.0, elevation=-15, W=48, H=64)
    sc = random_scene(40, seed=7, scale=0.06)
    rng = np.random.default_rng(1)
    A = rng.standard_normal((40, 3, 3)) * 0.05
    S = A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3)
    sc2 = dict(means3D=sc["means3D"], opacities=sc["opacities"], colors=rng.random((40, 3)),
               cov3D=np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1))
    _fd_check(sc2, cam, 0, dict(means3D="dL_dmeans3D", colors="dL_dcolors", opacities="dL_dopacity",
                                cov3D="dL_dcov3D"))


def test_sh_clamp_blocks_gradient():
    cam = camera_np(0.0, W=32, H=32)
    sc = random_scene(20, seed=4, scale=0.08)
    sc["shs"][:, 0, 0] = -5.0  # red clamps at 0 everywhere
    st = _fwd(sc, cam, f64=True)
    assert st["clamped"][st["radii"] > 0, 0].all() and not st["clamped"][:, 1].any()
    g = ro.backward(st, np.ones((3, 32, 32)), np.zeros((1, 32, 32)), np.zeros((3, 32, 32)), np.zeros((1, 32, 32)))
    assert np.all(g["dL_dshs"][:, 0, 0] == 0) and np.any(g["dL_dshs"][:, 0, 1] != 0)


def test_eval_sh_matches_reference_formula():
    """Degree-3 SH colour vs utils/sh_utils.py:57-112 (golden fixture generated from the reference)."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "sh_eval.npz")
    if not os.path.exists(path):
        pytest.skip("golden fixture missing")
    z = np.load(path)
    cam = camera_np(0.0, W=32, H=32)
    for deg in (0, 1, 2, 3):
        sc = dict(means3D=z["xyz"], shs=z["shs"], opacities=np.full((len(z["xyz"]), 1), 0.5),
                  scales=np.full((len(z["xyz"]), 3), 0.01), rotations=np.tile([1.0, 0, 0, 0], (len(z["xyz"]), 1)))
        cam2 = dict(cam, campos=z["campos"])
        st = _fwd(sc, cam2, deg=deg, f64=True)
        vis = st["radii"] > 0
        assert vis.sum() > 10
        np.testing.assert_allclose(st["feat"][vis, :3], z[f"rgb_deg{deg}"][vis], atol=2e-6)  # fixture is fp32


def test_knn_and_dist2_against_numpy():
    rng = np.random.default_rng(0)
    ref, q = rng.standard_normal((64, 3)), rng.standard_normal((300, 3))
    d, i = ro.knn(ref, q, 4, f64=True)
    D = np.linalg.norm(q[:, None] - ref[None], axis=-1)
    order = np.argsort(D, axis=1, kind="stable")[:, :4]
    assert np.array_equal(i, order)
    np.testing.assert_allclose(d, np.take_along_axis(D, order, 1), atol=1e-12)
    pts = rng.standard_normal((200, 3))
    D2 = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    np.fill_diagonal(D2, np.inf)
    np.testing.assert_allclose(ro.dist2(pts, f64=True), np.sort(D2, 1)[:, :3].mean(1), rtol=1e-12)
