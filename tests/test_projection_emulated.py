"""The projection kernel (dimo_amd/csrc/preprocess.hip, forward) run on the CPU SIMT emulation (tests/simt/) against
the oracle -- radii, tile rectangles, tiles touched and the per-Gaussian fp32 outputs (pixel mean, conic, depth, colour,
normal) BIT FOR BIT, as tests/test_gpu_raster.py demands of the GPU (the file is built with -ffp-contract=off there and
here) -- and then chained into the emulated binning: the geometry workspace the emulated projection leaves goes
straight into the emulated level1 / bucket_sort / level2_fill, whose lists must be the oracle's.  The same source text
as the product, no GPU; the GPU tests stay the parity tests proper."""
import ctypes as C

import numpy as np
import pytest

from oracle import raster_oracle as ro
from tests.scenes import camera_np, random_scene
from tests.simt import build as simt_build
from tests.simt import harness as hz

_P = None


def P():
    global _P
    if _P is None:
        lib = C.CDLL(simt_build.build(target="project"))
        p, f, i = C.c_void_p, C.c_float, C.c_int
        lib.dimo_raster_preprocess_forward.argtypes = [i, i, i, i, i, p, p, p, p, p, p, p, f, p, p, p, f, f, p, p, C.c_size_t, p, p]
        lib.simt_project_layout.argtypes = [i, C.POINTER(C.c_size_t)]
        _P = lib
    return _P


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _project(sc, cam, deg):
    f = lambda k: None if sc.get(k) is None else np.ascontiguousarray(sc[k], np.float32)
    N = len(sc["means3D"])
    lay = (C.c_size_t * 8)()
    P().simt_project_layout(N, lay)
    G = dict(zip(("splat", "rect", "tiles", "flags", "total", "key32", "block_sums", "bytes"), [int(x) for x in lay]))
    geom = hz.workspace(G["bytes"], 0x5A)
    radii = np.full(N, -1, np.int32)
    R = np.zeros(1, np.int64)
    arrs = {k: f(k) for k in ("means3D", "shs", "colors", "opacities", "scales", "rotations", "cov3D")}
    ptr = lambda a: None if a is None else a.ctypes.data
    M = 0 if arrs["shs"] is None else arrs["shs"].reshape(N, -1, 3).shape[1]
    view, proj, campos = (np.ascontiguousarray(cam[k], np.float32).reshape(-1) for k in ("view", "proj", "campos"))
    rc = P().dimo_raster_preprocess_forward(N, deg, M, cam["H"], cam["W"], ptr(arrs["means3D"]), ptr(arrs["shs"]),
                                            ptr(arrs["colors"]), ptr(arrs["opacities"]), ptr(arrs["scales"]),
                                            ptr(arrs["rotations"]), ptr(arrs["cov3D"]), 1.0, ptr(view), ptr(proj),
                                            ptr(campos), cam["tanfovx"], cam["tanfovy"], ptr(radii), geom.ctypes.data,
                                            geom.nbytes, R.ctypes.data, None)
    assert rc == 0
    o = ro.preprocess_forward(arrs["means3D"], arrs["shs"], arrs["colors"], arrs["opacities"], arrs["scales"],
                              arrs["rotations"], arrs["cov3D"], 1.0, cam["view"], cam["proj"], cam["campos"],
                              cam["tanfovx"], cam["tanfovy"], cam["H"], cam["W"], deg)
    return geom, G, radii, int(R[0]), o


@pytest.mark.parametrize("N,H,W,deg,M", [(1000, 128, 128, 0, 1), (3000, 128, 160, 3, 16), (2000, 64, 64, 1, 4),
                                         (1500, 50, 70, 0, 1), (4000, 512, 512, 2, 9)])
def test_emulated_projection_bit_exact(N, H, W, deg, M):
    sc = random_scene(N, seed=N, sh_coeffs=M)
    cam = camera_np(10.0, W=W, H=H)
    geom, G, radii, R, o = _project(sc, cam, deg)
    assert R == o["R"]
    assert np.array_equal(radii, o["radii"])
    rect = geom[G["rect"]:G["rect"] + 8 * N].view(np.uint16).reshape(N, 4).astype(np.int32)
    tiles = geom[G["tiles"]:G["tiles"] + 4 * N].view(np.uint32)
    vis = o["radii"] > 0
    assert np.array_equal(tiles, o["tiles_touched"])
    assert np.array_equal(rect[vis], o["rect"][vis])
    sp = geom[G["splat"]:G["splat"] + 64 * N].view(np.float32).reshape(N, 16)
    assert np.array_equal(_bits(sp[vis, 0:2]), _bits(o["xy"][vis])), "pixel means differ"
    assert np.array_equal(_bits(sp[vis, 2:5]), _bits(o["conic_op"][vis, :3])), "conics differ"
    assert np.array_equal(_bits(sp[vis, 9]), _bits(o["feat"][vis, 3])), "depths differ"
    assert np.array_equal(_bits(sp[vis, 6:9]), _bits(o["feat"][vis, 0:3])), "colours differ"
    assert np.array_equal(_bits(sp[vis, 10:13]), _bits(o["feat"][vis, 4:7])), "normals differ"
    key = geom[G["key32"]:G["key32"] + 4 * N].view(np.uint32)
    assert np.array_equal(key[vis], _bits(o["feat"][vis, 3])) and np.all(key[~vis] == 0xFFFFFFFF)


def test_emulated_projection_into_emulated_binning():
    """The emulated projection's geometry workspace, as it is, through the emulated binning chain."""
    cam = camera_np(25.0, W=256, H=192)
    sc = random_scene(6000, seed=17, scale=0.03)
    geom, G, radii, R, o = _project(sc, cam, 0)
    ro.bin_tiles(o)
    L = hz.lib()
    N, H, W = 6000, 192, 256
    Gb, Bb = hz._layouts(N, H, W, max(R, 1))
    assert Gb["bytes"] == G["bytes"]
    # (the workspace was poisoned: the bucket totals and counters the binning starts from are what the projection's first
    # block cleared)
    bin_ws = hz.workspace(Bb["bytes"], 0xA5)
    assert L.simt_bin_instances(N, H, W, max(R, 1), geom.ctypes.data, bin_ws.ctypes.data) == 0
    vals = bin_ws[Bb["vals"]:Bb["vals"] + 4 * R].view(np.uint32)
    ranges = bin_ws[Bb["ranges"]:Bb["ranges"] + 8 * Bb["T"]].view(np.uint32).reshape(-1, 2)
    assert np.array_equal(vals, o["vals_sorted"]) and np.array_equal(ranges, o["ranges"])
    # the (supertile, depth bin) ENTRIES the projection counted per Gaussian -- what the level-1 capacity check rests on
    # (a mutant that dropped a "+ 1" from that count went unnoticed: tools/mutate_emulated.py) -- against the rectangles
    ss = L.simt_supertile_shift(H, W)
    rect = geom[G["rect"]:G["rect"] + 8 * N].view(np.uint16).reshape(N, 4).astype(np.int64)
    on = o["tiles_touched"] > 0
    ent = ((((rect[:, 2] - 1) >> ss) - (rect[:, 0] >> ss) + 1) * (((rect[:, 3] - 1) >> ss) - (rect[:, 1] >> ss) + 1))[on].sum()
    total = geom[G["total"]:G["total"] + 16].view(np.uint32)
    assert int(total[0]) == R and int(total[1]) == 0 and int(total[2]) == int(ent)
