"""The rasterizer's whole C ABI -- dimo_raster_preprocess_forward, dimo_raster_render_forward (tile binning + blend
forward), dimo_raster_backward (blend backward + projection backward): preprocess.hip, binning.hip, blend.hip and
wave_ops.hpp as hipcc compiles them -- run on the CPU SIMT emulation (tests/simt/) against the oracle, with the checks
and tolerances tests/test_gpu_raster.py holds the GPU to: integer stages and per-Gaussian outputs bit for bit, images and
gradients within 1e-4 L1.  The blend backward's 13-value wave reduction runs as the shim's instruction-for-instruction
spelling of its asm block (DPP adds with row / bank masks, permlane swaps).  No GPU; the GPU tests stay the parity tests
proper."""
import ctypes as C

import numpy as np
import pytest

from oracle import raster_oracle as ro
from tests.scenes import camera_np, random_scene
from tests.simt import build as simt_build
from tests.simt import harness as hz

L1_TOL = 1e-4
_R = None


def R_():
    global _R
    if _R is None:
        lib = C.CDLL(simt_build.build(target="raster"))
        p, f, i, z, q = C.c_void_p, C.c_float, C.c_int, C.c_size_t, C.c_int64
        lib.dimo_raster_preprocess_forward.argtypes = [i, i, i, i, i, p, p, p, p, p, p, p, f, p, p, p, f, f, p, p, z, p, p]
        lib.dimo_raster_render_forward.argtypes = [i, i, i, q, p, p, p, z, p, z, p, p, p, p, p]
        lib.dimo_raster_backward.argtypes = [i, i, i, i, i, q, p, p, p, p, p, p, p, f, p, p, p, p, f, f] + [p] * 17 + [z, p]
        lib.simt_raster_layout.argtypes = [i, i, i, q, C.POINTER(z)]
        _R = lib
    return _R


def _ptr(a):
    return None if a is None else a.ctypes.data


def _layout(N, H, W, r_cap):
    lay = (C.c_size_t * 16)()
    R_().simt_raster_layout(N, H, W, r_cap, lay)
    names = ("geom_bytes", "bin_bytes", "img_bytes", "scratch_bytes", "splat", "rect", "tiles", "offsets", "total",
             "vals", "ranges", "final_T", "n_contrib", "final_acc", "cap", "splat_size")
    return dict(zip(names, (int(x) for x in lay)))


class Run:
    """One render through the emulated C ABI, the way dimo_amd/rasterizer.py drives the product's."""

    def __init__(self, sc, cam, bg, deg, with_normal=True, scale_mod=1.0):
        f = lambda k: None if sc.get(k) is None else np.ascontiguousarray(sc[k], np.float32)
        self.a = {k: f(k) for k in ("means3D", "shs", "colors", "opacities", "scales", "rotations", "cov3D")}
        a = self.a
        self.N = N = len(a["means3D"])
        self.H, self.W, self.deg, self.with_normal, self.scale_mod = cam["H"], cam["W"], deg, with_normal, scale_mod
        self.M = 0 if a["shs"] is None else a["shs"].reshape(N, -1, 3).shape[1]
        self.cam = cam
        self.view, self.proj, self.campos = (np.ascontiguousarray(cam[k], np.float32).reshape(-1) for k in ("view", "proj", "campos"))
        self.bg = np.asarray(bg, np.float32)
        H, W = self.H, self.W
        L0 = _layout(N, H, W, 1)
        self.geom = hz.workspace(L0["geom_bytes"], 0x5A)
        self.radii = np.full(N, -1, np.int32)
        r = np.zeros(1, np.int64)
        rc = R_().dimo_raster_preprocess_forward(N, deg, self.M, H, W, _ptr(a["means3D"]), _ptr(a["shs"]), _ptr(a["colors"]),
                                                 _ptr(a["opacities"]), _ptr(a["scales"]), _ptr(a["rotations"]), _ptr(a["cov3D"]),
                                                 scale_mod, _ptr(self.view), _ptr(self.proj), _ptr(self.campos), cam["tanfovx"],
                                                 cam["tanfovy"], _ptr(self.radii), _ptr(self.geom), self.geom.nbytes, _ptr(r), None)
        assert rc == 0
        self.R = int(r[0])
        self.r_cap = max(self.R, 1)
        self.L = L = _layout(N, H, W, self.r_cap)
        self.bin = hz.workspace(L["bin_bytes"], 0x5A)
        self.img = hz.workspace(L["img_bytes"], 0x5A)
        self.color, self.depth = np.full((3, H, W), np.nan, np.float32), np.full((1, H, W), np.nan, np.float32)
        self.normal = np.full((3, H, W), np.nan, np.float32) if with_normal else None
        self.alpha = np.full((1, H, W), np.nan, np.float32)
        rc = R_().dimo_raster_render_forward(N, H, W, self.r_cap, _ptr(self.bg), _ptr(self.geom), _ptr(self.bin), self.bin.nbytes,
                                             _ptr(self.img), self.img.nbytes, _ptr(self.color), _ptr(self.depth),
                                             _ptr(self.normal), _ptr(self.alpha), None)
        assert rc == 0

    def view_of(self, buf, off, dtype, count):
        return buf[off:off + count * np.dtype(dtype).itemsize].view(dtype)

    def backward(self, gw):
        a, N = self.a, self.N
        g = dict(means3D=np.full((N, 3), np.nan, np.float32), means2D=np.full((N, 3), np.nan, np.float32),
                 shs=None if a["shs"] is None else np.full((N, max(self.M, 1), 3), np.nan, np.float32),
                 colors=None if a["colors"] is None else np.full((N, 3), np.nan, np.float32),
                 opacities=np.full((N, 1), np.nan, np.float32),
                 scales=None if a["cov3D"] is not None else np.full((N, 3), np.nan, np.float32),
                 rotations=None if a["cov3D"] is not None else np.full((N, 4), np.nan, np.float32),
                 cov3D=None if a["cov3D"] is None else np.full((N, 6), np.nan, np.float32))
        scratch = hz.workspace(self.L["scratch_bytes"], 0x5A)
        gw = [np.ascontiguousarray(x, np.float32) for x in gw]
        rc = R_().dimo_raster_backward(
            N, self.deg, self.M, self.H, self.W, self.r_cap, _ptr(a["means3D"]), _ptr(a["shs"]), _ptr(a["colors"]),
            _ptr(a["opacities"]), _ptr(a["scales"]), _ptr(a["rotations"]), _ptr(a["cov3D"]), self.scale_mod, _ptr(self.view),
            _ptr(self.proj), _ptr(self.campos), _ptr(self.bg), self.cam["tanfovx"], self.cam["tanfovy"], _ptr(self.radii),
            _ptr(self.geom), _ptr(self.bin), _ptr(self.img), _ptr(gw[0]), _ptr(gw[1]), _ptr(gw[2]) if self.with_normal else None,
            _ptr(gw[3]), _ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["shs"]), _ptr(g["colors"]), _ptr(g["opacities"]),
            _ptr(g["scales"]), _ptr(g["rotations"]), _ptr(g["cov3D"]), _ptr(scratch), scratch.nbytes, None)
        assert rc == 0
        return g


def _oracle(sc, cam, bg, deg, scale_mod=1.0):
    f = lambda k: None if sc.get(k) is None else np.asarray(sc[k], np.float32)
    return ro.forward(f("means3D"), f("shs"), f("colors"), f("opacities"), f("scales"), f("rotations"), f("cov3D"),
                      scale_mod, cam["view"], cam["proj"], cam["campos"], np.asarray(bg, np.float32), cam["tanfovx"],
                      cam["tanfovy"], cam["H"], cam["W"], deg, f64=False)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _check_forward(sc, cam, bg, deg, with_normal=True, scale_mod=1.0):
    r = Run(sc, cam, bg, deg, with_normal, scale_mod)
    o = _oracle(sc, cam, bg, deg, scale_mod)
    N, H, W, L = r.N, r.H, r.W, r.L
    T = ((H + 15) // 16) * ((W + 15) // 16)
    assert r.R == o["R"]
    assert np.array_equal(r.radii, o["radii"])
    assert np.array_equal(r.view_of(r.geom, L["tiles"], np.uint32, N), o["tiles_touched"])
    assert np.array_equal(r.view_of(r.geom, L["offsets"], np.uint32, N), o["offsets"])
    sp = r.view_of(r.geom, L["splat"], np.float32, N * L["splat_size"] // 4).reshape(N, -1)
    vis = o["radii"] > 0
    assert np.array_equal(_bits(sp[vis, 0:2]), _bits(o["xy"][vis])), "pixel means differ"
    assert np.array_equal(_bits(sp[vis, 2:5]), _bits(o["conic_op"][vis, :3])), "conics differ"
    assert np.array_equal(r.view_of(r.bin, L["vals"], np.uint32, r.R), o["vals_sorted"]), "sorted order differs"
    assert np.array_equal(r.view_of(r.bin, L["ranges"], np.uint32, 2 * T).reshape(T, 2), o["ranges"].reshape(T, 2))
    nc = r.view_of(r.img, L["n_contrib"], np.uint32, H * W).reshape(H, W)
    assert (nc != o["n_contrib"]).mean() <= 2e-3, "n_contrib mismatch beyond expf threshold flips"
    for got, ok in ((r.color, "out_color"), (r.depth, "out_depth"), (r.alpha, "out_alpha"), (r.normal, "out_normal")):
        if got is None:
            continue
        assert np.isfinite(got).all()
        err = np.abs(got - o[ok]).mean()
        assert err <= L1_TOL, (ok, err)
    fT = r.view_of(r.img, L["final_T"], np.float32, H * W).reshape(H, W)
    assert np.abs(fT - o["final_T"]).mean() <= L1_TOL
    return r, o


@pytest.mark.parametrize("N,H,W,deg,M", [(1000, 128, 128, 0, 1), (2500, 80, 96, 0, 1), (1500, 64, 80, 3, 16),
                                         (1200, 50, 70, 1, 4)])
def test_emulated_forward_parity(N, H, W, deg, M):
    cam = camera_np(25.0, elevation=8, W=W, H=H)
    sc = random_scene(N, seed=N, sh_coeffs=M, scale=0.02)
    _check_forward(sc, cam, (0.2, 0.4, 0.6), deg)


def test_emulated_forward_four_output_flavour_and_scale_modifier():
    cam = camera_np(200.0, W=96, H=64)
    sc = random_scene(1500, seed=11, scale=0.03)
    _check_forward(sc, cam, (1.0, 1.0, 1.0), 0, with_normal=False, scale_mod=0.7)


def test_emulated_forward_long_lists():
    """Lists of many hundred entries per tile: several 256-record batches, early termination, checkpoints at every
    bucket of 64 behind the head."""
    cam = camera_np(40.0, W=48, H=48)
    sc = random_scene(4000, seed=3, scale=0.05)
    r, o = _check_forward(sc, cam, (0.0, 0.0, 0.0), 0)
    assert (o["ranges"].reshape(-1, 2)[:, 1] - o["ranges"].reshape(-1, 2)[:, 0]).max() > 600


def _rel_l1(a, b):
    return np.abs(a - b).sum() / (np.abs(b).sum() + 1e-12)


GRADS_SH = dict(means3D="dL_dmeans3D", means2D="dL_dmean2D", shs="dL_dshs", opacities="dL_dopacity",
                scales="dL_dscales", rotations="dL_drot")


def _check_backward(sc, cam, bg, deg, names, with_normal=True, seed=0):
    H, W = cam["H"], cam["W"]
    rng = np.random.default_rng(seed)
    gw = [rng.standard_normal(s).astype(np.float32) for s in ((3, H, W), (1, H, W), (3, H, W), (1, H, W))]
    if not with_normal:
        gw[2] = np.zeros((3, H, W), np.float32)
    r = Run(sc, cam, bg, deg, with_normal)
    g = r.backward(gw)
    o = _oracle(sc, cam, bg, deg)
    go = ro.backward(o, *gw)
    for k, gk in names.items():
        a, b = g[k].reshape(-1), go[gk].reshape(-1)
        if k == "means2D":
            a = g[k][:, :2].reshape(-1)
        assert np.isfinite(a).all(), k
        err = _rel_l1(a, b)
        assert err <= L1_TOL, (k, err)


@pytest.mark.parametrize("N,H,W,deg,M", [(1000, 128, 128, 0, 1), (2000, 80, 96, 3, 16), (1500, 50, 70, 1, 4)])
def test_emulated_backward_parity(N, H, W, deg, M):
    cam = camera_np(25.0, elevation=8, W=W, H=H)
    sc = random_scene(N, seed=N + 1, sh_coeffs=M, scale=0.02)
    _check_backward(sc, cam, (1.0, 1.0, 1.0), deg, GRADS_SH)


def test_emulated_backward_four_output_flavour():
    cam = camera_np(200.0, W=96, H=64)
    sc = random_scene(1500, seed=21, scale=0.03)
    _check_backward(sc, cam, (0.1, 0.2, 0.3), 0, GRADS_SH, with_normal=False)


def test_emulated_backward_long_lists_read_checkpoints():
    cam = camera_np(40.0, W=48, H=48)
    sc = random_scene(4000, seed=3, scale=0.05)
    _check_backward(sc, cam, (0.0, 0.0, 0.0), 0, GRADS_SH)


@pytest.mark.parametrize("opacity", [(0.05, 0.05), (0.01, 0.1)], ids=["every_opacity_0.05", "opacity_0.01_to_0.1"])
def test_emulated_init_regime_whole_lists_deep(opacity):
    """SURVEY 8d's "init" regime (every opacity 0.05, renderer/latent_gs_renderer.py:431) at small size: nothing
    saturates, every pixel's last entry sits near the end of its tile's list of several hundred, so the forward
    checkpoints every bucket and the backward's deep queue carries most of the items (tests/test_gpu_raster.py and
    tests/test_gpu_executor.py hold the same regime at C3 size on the GPU)."""
    cam = camera_np(40.0, W=48, H=48)
    sc = random_scene(3000, seed=13, scale=0.05, opacity=opacity)
    r, o = _check_forward(sc, cam, (0.0, 0.0, 0.0), 0)
    rg = o["ranges"].reshape(-1, 2).astype(np.int64)
    assert o["n_contrib"].mean() > 0.8 * (rg[:, 1] - rg[:, 0]).mean() > 300
    _check_backward(sc, cam, (0.0, 0.0, 0.0), 0, GRADS_SH)


def test_emulated_backward_gaussians_over_more_than_64_tiles():
    """Two ways from the blend backward's gradient records to a Gaussian's sums: up to 64 tile instances by its 64-bit hit
    mask (one atomic OR per record), more by the record flags and a whole wave (preprocess.hip).  A scene that has both
    kinds -- 100 tiles, a third of the Gaussians over most of them -- must give the oracle's gradients."""
    cam = camera_np(20.0, W=160, H=160)
    sc = random_scene(240, seed=5, scale=0.02)
    sc["scales"][::3] *= 10.0
    _check_backward(sc, cam, (0.2, 0.2, 0.2), 0, GRADS_SH)


def test_emulated_backward_precomputed_colour_and_cov():
    cam = camera_np(120.0, elevation=-20, W=96, H=96)
    sc = random_scene(1500, seed=5, scale=0.04)
    rng = np.random.default_rng(2)
    A = rng.standard_normal((1500, 3, 3)) * 0.03
    S = A @ A.transpose(0, 2, 1) + 1e-5 * np.eye(3)
    sc2 = dict(means3D=sc["means3D"], opacities=sc["opacities"], colors=rng.random((1500, 3)),
               cov3D=np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1))
    _check_backward(sc2, cam, (0, 0, 0), 0, dict(means3D="dL_dmeans3D", colors="dL_dcolors", opacities="dL_dopacity",
                                                  cov3D="dL_dcov3D"))


@pytest.mark.parametrize("order", ["reverse", "random:3", "random:11"])
def test_emulated_forward_and_backward_under_other_fiber_schedules(order, monkeypatch):
    """A lockstep GPU wave forgives a missing barrier; a workgroup whose fibers take turns in reverse or shuffled order
    does not (tests/simt/runtime.cpp).  Forward and backward of a scene with long lists, every schedule."""
    monkeypatch.setenv("SIMT_ORDER", order)
    cam = camera_np(40.0, W=48, H=64)
    sc = random_scene(3000, seed=9, scale=0.05)
    _check_forward(sc, cam, (0.3, 0.3, 0.3), 0)
    _check_backward(sc, cam, (0.3, 0.3, 0.3), 0, GRADS_SH)


def _fuzz_case(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([1, 3, 63, 64, 65, 200, 700, 1500]))
    H, W = int(rng.integers(1, 90)), int(rng.integers(1, 90))
    deg = int(rng.integers(0, 4))
    M = (deg + 1) ** 2
    scale = float(rng.choice([0.005, 0.02, 0.08, 0.4]))  # specks ... splats larger than the image
    cam = camera_np(float(rng.uniform(0, 360)), elevation=float(rng.uniform(-60, 60)), radius=float(rng.uniform(0.6, 3.0)),
                    W=W, H=H)
    sc = random_scene(N, seed=seed, sh_coeffs=M, scale=scale, opacity=(0.0 if seed % 3 == 0 else 0.2, 1.0))
    bg = tuple(float(x) for x in rng.random(3))
    return sc, cam, bg, deg, bool(rng.integers(0, 2))


@pytest.mark.parametrize("seed", range(16))
def test_emulated_raster_fuzz(seed):
    """Seeded random shapes the fixed cases do not hold: one to a few Gaussians, images of a single row / column / tile,
    sizes that are no multiple of 16, splats from specks to larger than the image, cameras inside the cloud (near-plane
    culling), opacities down to 0 (below the 1/255 threshold), both rasterizer flavours, SH degrees 0-3 -- forward and
    backward against the oracle."""
    sc, cam, bg, deg, with_normal = _fuzz_case(seed)
    r, o = _check_forward(sc, cam, bg, deg, with_normal)
    if o["R"] > 0:
        _check_backward(sc, cam, bg, deg, GRADS_SH, with_normal, seed=seed)
    else:
        g = r.backward([np.ones(s, np.float32) for s in ((3, cam["H"], cam["W"]), (1, cam["H"], cam["W"]),
                                                          (3, cam["H"], cam["W"]), (1, cam["H"], cam["W"]))])
        assert all(np.abs(v).max() == 0 for v in g.values() if v is not None)


@pytest.mark.parametrize("fraction", [0.9, 0.5, 0.1, 0.0])
def test_emulated_instance_capacity_overflow_stays_in_bounds(fraction):
    """A render with MORE tile instances than its workspaces were sized for (the capacity policy sizes them from a
    running bound without waiting for the device: dimo_amd/rasterizer.py CapacityPolicy) must flag the overflow, write
    nothing out of bounds and leave finite images; forward and backward.  Here every workspace sits between two guard
    regions in host memory, so a stray write is seen (or is a segmentation fault) rather than forgiven."""
    cam = camera_np(25.0, elevation=8, W=96, H=80)
    sc = random_scene(1500, seed=12, scale=0.04)
    full = Run(sc, cam, (0.1, 0.1, 0.1), 0)
    r_cap = max(1, int(full.R * fraction))
    G = 4096  # guard bytes on either side
    L = _layout(full.N, full.H, full.W, r_cap)

    def guarded(nbytes):
        buf = np.full(nbytes + 2 * G, 0xA7, np.uint8)
        return buf, buf[G:G + nbytes]
    bin_all, bin_ws = guarded(L["bin_bytes"])
    img_all, img_ws = guarded(L["img_bytes"])
    scr_all, scratch = guarded(L["scratch_bytes"])
    H, W, N = full.H, full.W, full.N
    color, depth, normal, alpha = (np.full(s, np.nan, np.float32) for s in ((3, H, W), (1, H, W), (3, H, W), (1, H, W)))
    rc = R_().dimo_raster_render_forward(N, H, W, r_cap, _ptr(full.bg), _ptr(full.geom), _ptr(bin_ws), bin_ws.nbytes,
                                         _ptr(img_ws), img_ws.nbytes, _ptr(color), _ptr(depth), _ptr(normal), _ptr(alpha), None)
    assert rc == 0
    total = full.geom[L["total"]:L["total"] + 16].view(np.uint32)
    assert int(total[0]) == full.R and int(total[1]) == 1, "the overflow is flagged next to the instance count"
    for a in (color, depth, normal, alpha):
        assert np.isfinite(a).all()
    a = full.a
    g = [np.full(s, np.nan, np.float32) for s in ((N, 3), (N, 3), (N, 1, 3), (N, 1), (N, 3), (N, 4))]
    gw = [np.ones(s, np.float32) for s in ((3, H, W), (1, H, W), (3, H, W), (1, H, W))]
    rc = R_().dimo_raster_backward(
        N, 0, 1, H, W, r_cap, _ptr(a["means3D"]), _ptr(a["shs"]), None, _ptr(a["opacities"]), _ptr(a["scales"]),
        _ptr(a["rotations"]), None, 1.0, _ptr(full.view), _ptr(full.proj), _ptr(full.campos), _ptr(full.bg),
        cam["tanfovx"], cam["tanfovy"], _ptr(full.radii), _ptr(full.geom), _ptr(bin_ws), _ptr(img_ws), _ptr(gw[0]),
        _ptr(gw[1]), _ptr(gw[2]), _ptr(gw[3]), _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), None, _ptr(g[3]), _ptr(g[4]), _ptr(g[5]),
        None, _ptr(scratch), scratch.nbytes, None)
    assert rc == 0
    for x in g:
        assert np.isfinite(x).all()
    for whole in (bin_all, img_all, scr_all):
        assert (whole[:G] == 0xA7).all() and (whole[-G:] == 0xA7).all(), "a write outside a workspace"


def test_emulated_backward_view_direction_gradient_of_the_sh_colours():
    """dL/dmeans3D has a part that goes through the VIEW DIRECTION of the SH colour (preprocess.hip: dRGB/d(direction));
    next to the geometric part it is small, and a 1 % error in one of its degree-3 terms stayed under the 1e-4 bar of the
    ordinary scenes (tools/mutate_emulated.py).  Here the degree-2 / degree-3 bands are large and the loss sees the colour
    image only, so that part carries the gradient."""
    cam = camera_np(25.0, elevation=8, W=64, H=64, radius=1.2)  # (a close camera: directions differ across the cloud)
    sc = random_scene(1200, seed=31, sh_coeffs=16, scale=0.03)
    sc["shs"][:, 4:] *= 12.0
    sc["shs"][:, 0] += 80.0  # (keeps colour + 0.5 positive: the clamp passes the gradient)
    H, W = cam["H"], cam["W"]
    rng = np.random.default_rng(3)
    gw = [rng.standard_normal((3, H, W)).astype(np.float32), np.zeros((1, H, W), np.float32),
          np.zeros((3, H, W), np.float32), np.zeros((1, H, W), np.float32)]
    r = Run(sc, cam, (0.0, 0.0, 0.0), 3)
    g = r.backward(gw)
    o = _oracle(sc, cam, (0.0, 0.0, 0.0), 3)
    go = ro.backward(o, *gw)
    assert (o["clamped"] == 0).mean() > 0.95
    for k, gk in (("means3D", "dL_dmeans3D"), ("shs", "dL_dshs")):
        err = _rel_l1(g[k].reshape(-1), go[gk].reshape(-1))
        assert err <= L1_TOL, (k, err)
