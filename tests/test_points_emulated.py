"""KNN, distCUDA2 (brute force and the O(N) grid form) and farthest point sampling: dimo_amd/csrc/knn.hip and fps.hip --
the SAME source text hipcc compiles for gfx950 -- run on the CPU SIMT emulation (tests/simt/) against the oracle, bit
for bit (both files are built with -ffp-contract=off for the GPU and here, as the oracle is).  The cases of
tests/test_gpu_ops.py at sizes the emulation finishes in seconds; the GPU tests stay the parity tests proper."""
import ctypes as C

import numpy as np
import pytest

from oracle import raster_oracle as ro
from tests.simt import build as simt_build

_L = None


def L():
    global _L
    if _L is None:
        lib = C.CDLL(simt_build.build(target="points"))
        p = C.c_void_p
        lib.dimo_knn.argtypes = [C.c_int, C.c_int, C.c_int, p, p, p, p, p]
        lib.dimo_knn_seeded.argtypes = [C.c_int, C.c_int, C.c_int, p, p, p, p, p, p]
        lib.dimo_dist2.argtypes = [C.c_int, p, p, p]
        lib.dimo_dist2_workspace_bytes.argtypes = [C.c_int]
        lib.dimo_dist2_workspace_bytes.restype = C.c_size_t
        lib.dimo_dist2_grid.argtypes = [C.c_int, p, p, p, C.c_size_t, p]
        lib.dimo_farthest_point_sample.argtypes = [C.c_int, C.c_int, p, p, p, p]
        _L = lib
    return _L


def _ptr(a):
    return a.ctypes.data if a is not None else None


def _knn(ref, q, k, seed=None):
    M, N = len(ref), len(q)
    d, i = np.full((N, k), np.nan, np.float32), np.full((N, k), -7, np.int64)
    if seed is None:
        assert L().dimo_knn(M, N, k, _ptr(ref), _ptr(q), _ptr(d), _ptr(i), None) == 0
    else:
        seed = np.ascontiguousarray(seed, np.int64)
        assert L().dimo_knn_seeded(M, N, k, _ptr(ref), _ptr(q), _ptr(d), _ptr(i), _ptr(seed), None) == 0
    return d, i


def _clouds(M, N):
    rng = np.random.default_rng(M + N)
    ref = rng.standard_normal((M, 3)).astype(np.float32)
    q = rng.standard_normal((N, 3)).astype(np.float32)
    q[: min(N, M) // 2] = ref[: min(N, M) // 2]  # exact hits (zero distance) and ties
    ref[1] = ref[0]
    return ref, q


@pytest.mark.parametrize("M,N,k", [(512, 3000, 4), (24, 300, 4), (3, 50, 4), (700, 1000, 8), (300, 500, 1), (40, 200, 16)])
def test_emulated_knn_bit_exact(M, N, k):
    ref, q = _clouds(M, N)
    d, i = _knn(ref, q, k)
    do, io = ro.knn(ref, q, k)
    assert np.array_equal(i, io)
    assert np.array_equal(d.view(np.uint32), do.view(np.uint32))


@pytest.mark.parametrize("M,N", [(512, 4000), (5, 300)])
def test_emulated_seeded_knn_gives_the_unseeded_result_for_any_seeds(M, N):
    ref, q = _clouds(M, N)
    rng = np.random.default_rng(M * 7 + N)
    order = np.argsort((q[:, 0] > 0) * 2 + (q[:, 1] > 0))  # some spatial coherence inside the waves
    q = np.ascontiguousarray(q[order])
    do, io = ro.knn(ref, q, 4)
    _, prev = _knn(ref, (q + 0.01 * rng.standard_normal(q.shape)).astype(np.float32), 4)
    bad = rng.integers(0, M, (N, 4))
    bad[::3, 1] = bad[::3, 0]   # repeated index
    bad[1::3, 2] = -1           # negative
    bad[2::3, 3] = M + 5        # out of range
    for seed in (prev, rng.integers(0, M, (N, 4)), bad, io):
        d, i = _knn(ref, q, 4, seed=seed)
        assert np.array_equal(i, io)
        assert np.array_equal(d.view(np.uint32), do.view(np.uint32))


@pytest.mark.parametrize("N", [5, 1000, 6000])
def test_emulated_dist2_bit_exact(N):
    rng = np.random.default_rng(N)
    pts = rng.standard_normal((N, 3)).astype(np.float32)
    if N > 10:
        pts[7] = pts[3]  # duplicate point: distance 0 to its twin, still excluded only by index
    out = np.full(N, np.nan, np.float32)
    assert L().dimo_dist2(N, _ptr(pts), _ptr(out), None) == 0
    assert np.array_equal(out.view(np.uint32), ro.dist2(pts).view(np.uint32))


@pytest.mark.parametrize("case", ["ball", "clusters", "plane", "line", "duplicates", "outlier", "very_far_from_origin"])
def test_emulated_dist2_grid_equals_brute_force_bit_for_bit(case):
    rng = np.random.default_rng(11)
    N = 6000
    if case == "ball":
        pts = rng.standard_normal((N, 3)) * 0.3
    elif case == "clusters":
        pts = rng.standard_normal((N, 3)) * 0.01 + rng.integers(0, 5, (N, 3)) * 1.0
    elif case == "plane":
        pts = rng.random((N, 3))
        pts[:, 2] = 0.25
    elif case == "line":
        pts = np.zeros((N, 3))
        pts[:, 0] = rng.random(N)
    elif case == "duplicates":
        pts = np.repeat(rng.random((N // 6, 3)), 6, axis=0)
    elif case == "very_far_from_origin":
        pts = rng.random((N, 3)) + np.array([1.0e4, -0.7e4, 0.3e4])
    else:
        pts = rng.random((N, 3))
        pts[17] = (1.0e4, -3.0e3, 50.0)
    pts = np.ascontiguousarray(pts, np.float32)
    n = len(pts)
    brute, grid = np.full(n, np.nan, np.float32), np.full(n, np.nan, np.float32)
    assert L().dimo_dist2(n, _ptr(pts), _ptr(brute), None) == 0
    ws = np.full(int(L().dimo_dist2_workspace_bytes(n)), 0xA5, np.uint8)
    assert L().dimo_dist2_grid(n, _ptr(pts), _ptr(grid), _ptr(ws), ws.nbytes, None) == 0
    assert np.array_equal(brute.view(np.uint32), ro.dist2(pts).view(np.uint32))
    assert np.array_equal(grid.view(np.uint32), brute.view(np.uint32))


@pytest.mark.parametrize("N,K", [(1, 1), (50, 50), (1000, 64), (3000, 512), (1500, 7)])
def test_emulated_farthest_point_sampling(N, K):
    from oracle.regularizers_ref import farthest_point_sample_ref
    rng = np.random.default_rng(N + K)
    xyz = rng.standard_normal((N, 3)).astype(np.float32)
    if N > 20:
        xyz[11] = xyz[5]  # equal distances: the lowest index wins
    scratch = np.zeros(N, np.float32)
    out = np.full(K, -1, np.int64)
    assert L().dimo_farthest_point_sample(N, K, _ptr(xyz), _ptr(scratch), _ptr(out), None) == 0
    assert np.array_equal(out, farthest_point_sample_ref(xyz, K))


@pytest.mark.parametrize("N,K", [(700, 40), (300, 300)])
def test_emulated_farthest_point_sampling_with_ties_across_waves(N, K):
    """Points on a small integer lattice, many of them several times: the farthest distance is shared by points of
    different waves at almost every step, and the LOWEST index has to win (the cross-wave tie-break went untested with
    random points: tools/mutate_emulated.py)."""
    from oracle.regularizers_ref import farthest_point_sample_ref
    rng = np.random.default_rng(N)
    xyz = rng.integers(0, 4, (N, 3)).astype(np.float32)
    scratch = np.zeros(N, np.float32)
    out = np.full(K, -1, np.int64)
    assert L().dimo_farthest_point_sample(N, K, _ptr(xyz), _ptr(scratch), _ptr(out), None) == 0
    assert np.array_equal(out, farthest_point_sample_ref(xyz, K))


def test_emulated_farthest_point_sampling_tie_inside_one_thread_and_output_bound():
    """Two identical far points whose indices are 1024 apart -- the same thread scans both -- among points at the origin:
    the lower index has to win inside the thread too; and the kernel writes K indices, not K + 1 (a sentinel behind the
    output).  Both went unnoticed by a mutant (tools/mutate_emulated.py)."""
    from oracle.regularizers_ref import farthest_point_sample_ref
    N, K = 1100, 5
    xyz = np.zeros((N, 3), np.float32)
    xyz[5] = xyz[1029] = (3.0, 4.0, 0.0)
    xyz[300] = (0.0, 0.0, 1.0)
    scratch = np.zeros(N, np.float32)
    out = np.full(K + 1, -7, np.int64)
    assert L().dimo_farthest_point_sample(N, K, _ptr(xyz), _ptr(scratch), _ptr(out), None) == 0
    want = farthest_point_sample_ref(xyz, K)
    assert want[1] == 5
    assert np.array_equal(out[:K], want) and out[K] == -7
