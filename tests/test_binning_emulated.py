"""The binning chain of dimo_amd/csrc/binning.hip (the SAME source text hipcc compiles for gfx950) run on a CPU SIMT
emulation -- 256 fibers per workgroup, wave operations and barriers as rendezvous: tests/simt/ -- against the CPU
oracle's stable (tile | depth bits) sort (oracle/raster_ref.c: ref_bin) and a numpy restatement of it.

What this covers without a GPU: the integer logic of every stage (bucket counts, shares, bucket sort with its sub-bin,
slice and byte-pass paths, window tables, ordered filters, tile ranges, the overflow policy), for the shapes the GPU
parity tests drive (tests/test_gpu_raster.py) -- bit for bit.  What it does not: timing, the memory model, anything
the gfx950 backend does to the code.  The GPU tests stay the parity tests proper; the product never loads this.
"""
import numpy as np
import pytest

from oracle import raster_oracle as ro
from tests.scenes import camera_np, random_scene
from tests.simt import harness as hz


def _project(sc, cam):
    """Per-Gaussian words of the projection stage from the oracle: tile rectangle, tiles touched, depth bits."""
    f = lambda k: None if sc.get(k) is None else np.asarray(sc[k], np.float32)
    st = ro.preprocess_forward(f("means3D"), f("shs"), None, f("opacities"), f("scales"), f("rotations"), None, 1.0,
                               cam["view"], cam["proj"], cam["campos"], cam["tanfovx"], cam["tanfovy"], cam["H"],
                               cam["W"], 0)
    ro.bin_tiles(st)
    key = np.ascontiguousarray(st["feat"][:, 3], np.float32).view(np.uint32)
    return st, st["rect"], st["tiles_touched"], key


def _check(r, st):
    assert r["R"] == st["R"] and r["overflow"] == 0
    assert np.array_equal(r["offsets"], st["offsets"])
    assert np.array_equal(r["ranges"], st["ranges"])
    assert np.array_equal(r["dkeys"], (st["keys_sorted"] & np.uint64(0xFFFFFFFF)).astype(np.uint32)), "depth bits differ"
    assert np.array_equal(r["vals"], st["vals_sorted"]), "sorted order differs"
    T = len(st["ranges"])
    assert sorted(r["order"].tolist()) == list(range(T)), "dispatch order is not a permutation of the tiles"
    ln = (st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0]) >> 4
    assert np.all(np.diff(np.minimum(ln[r["order"]], 255)) <= 0), "tiles not by descending list length class"
    assert not r["bk_tot"].any(), "bucket totals not cleared for the next run"


def _scene_check(sc, cam, **kw):
    st, rect, tiles, key = _project(sc, cam)
    r = hz.run_binning(rect, tiles, key, cam["H"], cam["W"], **kw)
    _check(r, st)
    return r, st


@pytest.mark.parametrize("N,H,W", [(1000, 128, 128), (5000, 80, 96), (1500, 50, 70), (4000, 256, 256), (3000, 512, 512),
                                   (2000, 1040, 1296)])
def test_emulated_binning_matches_the_oracle(N, H, W):
    # supertile edges 1, 1, 1, 2, 4 and 8 tiles
    _scene_check(random_scene(N, seed=N, scale=0.03 if H < 1000 else 0.015), camera_np(10.0, W=W, H=H))


def test_emulated_binning_numpy_restatement_agrees_with_the_oracle():
    sc, cam = random_scene(800, seed=3), camera_np(5.0, W=96, H=64)
    st, rect, tiles, key = _project(sc, cam)
    e = hz.expected(rect, tiles, key, 64, 96)
    assert e["R"] == st["R"] and np.array_equal(e["vals"], st["vals_sorted"]) and np.array_equal(e["ranges"], st["ranges"])


def test_emulated_edge_cases_empty_and_offscreen():
    cam = camera_np(0.0, W=64, H=64)
    sc = random_scene(300, seed=1)
    sc["means3D"][:, 2] += 100.0  # everything behind / off screen
    r, st = _scene_check(sc, cam)
    assert r["R"] == 0
    sc = random_scene(1, seed=2)
    _scene_check(sc, cam)


@pytest.mark.parametrize("N", [2500])
def test_emulated_long_tile_lists_with_depth_ties(N):
    cam = camera_np(15.0, W=48, H=32)
    sc = random_scene(N, seed=N, scale=0.5, opacity=(0.01, 0.05))
    sc["means3D"] *= 0.2
    sc["means3D"][N // 2:] = sc["means3D"][: N - N // 2]
    r, st = _scene_check(sc, cam)
    assert (st["ranges"][:, 1] - st["ranges"][:, 0]).max() >= 0.9 * N


@pytest.mark.parametrize("case", ["one_depth_1500", "one_depth_5000", "slab_and_outliers", "two_depths"])
def test_emulated_depth_sort_bucket_paths(case):
    """The scenes of tests/test_gpu_raster.py::test_depth_sort_bucket_paths: LDS sort, slices, byte passes."""
    cam = camera_np(0.0, W=96, H=64)
    if case.startswith("one_depth"):
        N = int(case.split("_")[-1])
        sc = random_scene(N, seed=N, scale=0.02)
        sc["means3D"][:] = sc["means3D"][0]
    elif case == "slab_and_outliers":
        N = 7000
        sc = random_scene(N, seed=5, scale=0.02)
        view = np.asarray(cam["view"], np.float64)
        axis = view[:3, 2] / np.linalg.norm(view[:3, 2])
        p = sc["means3D"].astype(np.float64)
        p -= np.outer(p @ axis, axis) * (1.0 - 1e-5)
        sc["means3D"][:] = p.astype(np.float32)
        sc["means3D"][:6] += (np.arange(6)[:, None] * 0.6 - 1.5) * axis.astype(np.float32)
    else:
        N = 6000
        sc = random_scene(N, seed=9, scale=0.02)
        sc["means3D"][: N // 2] = sc["means3D"][0]
        sc["means3D"][N // 2:] = sc["means3D"][N // 2]
    _scene_check(sc, cam)


@pytest.mark.parametrize("far", [30.0, -1.0])
def test_emulated_depth_sort_with_floaters(far):
    cam = camera_np(0.0, W=128, H=96)
    sc = random_scene(30_000, seed=7, scale=0.006, opacity=(0.05, 0.4))
    view = np.asarray(cam["view"], np.float64)
    axis = (view[:3, 2] / np.linalg.norm(view[:3, 2])).astype(np.float32)
    sc["means3D"][:8] += np.float32(far) * axis
    _scene_check(sc, cam)


def test_emulated_stage_s1_shape_big_gaussians():
    """A few hundred Gaussians, each over most of the image (the BigList walk of level 1)."""
    cam = camera_np(0.0, W=256, H=256)
    sc = random_scene(512, seed=11, scale=0.4, opacity=(0.01, 0.05))
    _scene_check(sc, cam)


def test_emulated_batched_kernels_and_record_flags():
    """The batched entry point (blockIdx.y = render): two renders in one launch chain; the fill pass clears the
    backward's record flags over [0, R) and reports (R, overflow) to the step-level array."""
    cam = camera_np(20.0, W=128, H=128)
    st, rect, tiles, key = _project(random_scene(3000, seed=21), cam)
    for r in hz.run_binning(rect, tiles, key, 128, 128, n_batched=2):
        _check(r, st)
        assert r["totals_out"].tolist() == [st["R"], 0]
        assert not r["flags"].any()


def test_emulated_capacity_overflow_is_flagged_and_clamped():
    cam = camera_np(20.0, W=128, H=128)
    st, rect, tiles, key = _project(random_scene(3000, seed=22), cam)
    cap = st["R"] // 2
    r = hz.run_binning(rect, tiles, key, 128, 128, R_cap=cap)
    assert r["R"] == st["R"] and r["overflow"] == 1
    assert r["ranges"].max() <= cap
    keep = st["ranges"][:, 1] <= cap  # lists that end inside the capacity are complete and in order
    for t in np.nonzero(keep)[0][:50]:
        a, b = st["ranges"][t]
        assert np.array_equal(r["vals"][a:b], st["vals_sorted"][a:b])


@pytest.mark.parametrize("order", ["reverse", "random:1", "random:2"])
def test_emulated_binning_under_other_fiber_schedules(order, monkeypatch):
    """Between two rendezvous a fiber runs undisturbed, so the order the fibers take turns in decides whose plain LDS /
    global accesses come first.  The default (thread 0 first) is the order a missing barrier is most likely to survive;
    the reverse order and seeded shuffles are not.  Every path once more: slices, byte passes, overflow records, big
    Gaussians, the batched entry point."""
    monkeypatch.setenv("SIMT_ORDER", order)
    cam = camera_np(0.0, W=96, H=64)
    sc = random_scene(5000, seed=5000, scale=0.02)
    sc["means3D"][:] = sc["means3D"][0]  # one depth, one bucket of 5000 entries per tile: overflow records, byte passes
    _scene_check(sc, cam)
    cam = camera_np(0.0, W=128, H=96)
    sc = random_scene(30_000, seed=7, scale=0.006, opacity=(0.05, 0.4))  # slices
    _scene_check(sc, cam)
    cam = camera_np(20.0, W=256, H=256)
    st, rect, tiles, key = _project(random_scene(3000, seed=21, scale=0.08), cam)  # big Gaussians, 2 x 2-tile supertiles
    for r in hz.run_binning(rect, tiles, key, 256, 256, n_batched=2):
        _check(r, st)


def test_emulated_unsorted_model_many_buckets_per_workgroup():
    """Gaussians in no spatial order: every level-1 workgroup touches most buckets (the `direct` path of the group walk)."""
    cam = camera_np(0.0, W=512, H=512)
    sc = random_scene(20_000, seed=5, scale=0.01)
    _scene_check(sc, cam)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_emulated_binning_fuzz(seed):
    """Seeded random inputs straight at the chain (no projection): image sizes from 16^2 to 1300 x 2048 (every supertile
    edge, partial supertiles, more and fewer than 32 buckets), 1 to 20 000 Gaussians, rectangles of a tile / a few tiles /
    the whole image, visible fractions down to 5 %, depths uniform / one value / three values / a thin slab with
    floaters, the single and the batched entry points, clean and poisoned workspaces -- against the numpy restatement of
    the published order.  (275 such configurations were run when the chain was rewritten, under the forward, reverse
    and shuffled fiber schedules.)"""
    rng = np.random.default_rng(seed)
    for done in range(8):
        H, W, rect, tiles, key, what = hz.fuzz_config(rng)
        nb = int(rng.choice([0, 0, 2, 3]))
        poison = bool(rng.integers(0, 2))
        r = hz.run_binning(rect, tiles, key, H, W, n_batched=nb, poison=poison)
        e = hz.expected(rect, tiles, key, H, W)
        for rr in (r if nb else [r]):
            assert rr["R"] == e["R"] and rr["overflow"] == 0, (seed, done, what, nb)
            assert np.array_equal(rr["ranges"], e["ranges"]), (seed, done, what, nb)
            assert np.array_equal(rr["vals"], e["vals"]), (seed, done, what, nb)
            assert np.array_equal(rr["dkeys"], e["dkeys"]), (seed, done, what, nb)
