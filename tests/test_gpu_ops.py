"""GPU parity of the neighbour-search and SSIM kernels (through the C ABI) vs oracle / golden fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("M,N,k", [(512, 10000, 4), (24, 300, 4), (3, 50, 4), (5000, 2000, 8), (700, 1000, 1)])
def test_knn_bit_exact(M, N, k):
    from dimo_amd.knn_cuda import KNN
    rng = np.random.default_rng(M + N)
    ref = rng.standard_normal((M, 3)).astype(np.float32)
    q = rng.standard_normal((N, 3)).astype(np.float32)
    q[: min(N, M) // 2] = ref[: min(N, M) // 2]  # exact hits (zero distance) and ties
    ref[1] = ref[0]
    d, i = KNN(k=k, transpose_mode=True)(torch.tensor(ref).cuda()[None], torch.tensor(q).cuda()[None])
    assert d.shape == (1, N, k) and i.dtype == torch.int64
    do, io = ro.knn(ref, q, k)
    assert np.array_equal(i[0].cpu().numpy(), io)
    assert np.array_equal(d[0].cpu().numpy().view(np.uint32), do.view(np.uint32))


@pytest.mark.parametrize("M,N", [(512, 20000), (5, 300), (1500, 4097)])
def test_seeded_knn_gives_the_unseeded_result_for_any_seeds(M, N):
    """dimo_knn_seeded: the seeds only prune -- good ones (the neighbours before a small move), random ones, and
    broken ones (repeated, negative, out of range) all give the oracle's bits; ties and exact hits included."""
    from dimo_amd.knn_cuda import knn_points
    rng = np.random.default_rng(M * 7 + N)
    ref = rng.standard_normal((M, 3)).astype(np.float32)
    q = rng.standard_normal((N, 3)).astype(np.float32)
    q[: min(N, M) // 2] = ref[: min(N, M) // 2]
    ref[1] = ref[0]
    order = np.argsort((q[:, 0] > 0) * 2 + (q[:, 1] > 0))  # some spatial coherence inside the waves
    q = np.ascontiguousarray(q[order])
    do, io = ro.knn(ref, q, 4)
    R, Q = torch.tensor(ref).cuda(), torch.tensor(q).cuda()
    _, prev = knn_points(R, (Q + 0.01 * torch.randn_like(Q)).contiguous(), 4)
    bad = torch.randint(0, M, (N, 4), device="cuda")
    bad[::3, 1] = bad[::3, 0]          # repeated index
    bad[1::3, 2] = -1                  # negative
    bad[2::3, 3] = M + 5               # out of range
    for seed in (prev, torch.randint(0, M, (N, 4), device="cuda"), bad, torch.tensor(io).cuda()):
        d, i = knn_points(R, Q, 4, seed=seed.contiguous())
        assert np.array_equal(i.cpu().numpy(), io)
        assert np.array_equal(d.cpu().numpy().view(np.uint32), do.view(np.uint32))


def test_knn_matches_deform_fixture():
    from dimo_amd.knn_cuda import knn_points
    z = np.load(os.path.join(GOLD, "deform_latent.npz"))
    d, i = knn_points(torch.tensor(z["param.c_xyz"]).cuda(), torch.tensor(z["param.xyz"]).cuda(), 4)
    assert np.array_equal(i.cpu().numpy(), z["knn_idx"])
    np.testing.assert_array_equal(d.cpu().numpy(), z["knn_dist"])


@pytest.mark.parametrize("N", [5, 1000, 30000])
def test_dist2_bit_exact(N):
    from dimo_amd.simple_knn._C import distCUDA2
    rng = np.random.default_rng(N)
    pts = rng.standard_normal((N, 3)).astype(np.float32)
    if N > 10:
        pts[7] = pts[3]  # duplicate point: distance 0 to its twin, still excluded only by index
    out = distCUDA2(torch.tensor(pts).cuda()).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ro.dist2(pts).view(np.uint32))


@pytest.mark.parametrize("case", ["ball", "clusters", "plane", "line", "duplicates", "outlier", "far_from_origin",
                                  "very_far_from_origin"])
def test_dist2_grid_equals_brute_force_bit_for_bit(case):
    """dimo_dist2_grid (uniform grid, O(N): renderer/latent_gs_renderer.py:426 at the 1e5-1e6 points of a real
    initialisation) against the brute-force dimo_dist2 (itself bit-exact against the oracle above) on layouts that
    stress the grid: empty cells, a degenerate bounding box, many points per cell, one far point stretching the box."""
    import ctypes as C
    from dimo_amd import _lib
    rng = np.random.default_rng(11)
    N = 60000
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
    elif case in ("far_from_origin", "very_far_from_origin"):
        # |origin| / extent of 1e3 and 1e4: the own-cell face distance must not be computed as q - (o + c * cs), whose
        # rounding at ulp(|o|) exceeds the ring search's safety margin there (advisor, round 3)
        off = 1.0e3 if case == "far_from_origin" else 1.0e4
        pts = rng.random((N, 3)) + np.array([off, -0.7 * off, 0.3 * off])
    else:
        pts = rng.random((N, 3))
        pts[17] = (1.0e4, -3.0e3, 50.0)
    pts = torch.tensor(np.ascontiguousarray(pts, np.float32)).cuda()
    n = pts.shape[0]
    L = _lib.lib()
    brute, grid = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    _lib.check(L.dimo_dist2(n, _lib.ptr(pts), _lib.ptr(brute), _lib.current_stream()), "dimo_dist2")
    ws = torch.empty(L.dimo_dist2_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    _lib.check(L.dimo_dist2_grid(n, _lib.ptr(pts), _lib.ptr(grid), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               "dimo_dist2_grid")
    assert torch.equal(brute.view(torch.int32), grid.view(torch.int32))
    assert L.dimo_dist2_grid(n, _lib.ptr(pts), _lib.ptr(grid), _lib.ptr(ws), 16, _lib.current_stream()) == -3  # workspace


def test_ssim_matches_reference_fixture():
    from dimo_amd.fused_ssim import fused_ssim
    z = np.load(os.path.join(GOLD, "image_losses.npz"))
    a = torch.tensor(z["img1"]).cuda().requires_grad_(True)
    b = torch.tensor(z["img2"]).cuda()
    s = fused_ssim(a, b)
    s.backward()
    assert abs(s.item() - float(z["ssim"])) < 1e-5
    assert np.abs(a.grad.cpu().numpy() - z["g_img1"]).max() < 1e-6 + 1e-4 * np.abs(z["g_img1"]).max()


@pytest.mark.parametrize("shape", [(1, 3, 128, 128), (4, 3, 67, 45), (2, 1, 16, 16), (8, 3, 512, 512)])
def test_ssim_against_torch_conv(shape):
    """Same formula as src/loss.py:144-175 evaluated with torch conv2d on the GPU (fp32 reference)."""
    import torch.nn.functional as F
    from dimo_amd.fused_ssim import fused_ssim
    g = torch.Generator().manual_seed(shape[2])
    a = torch.rand(shape, generator=g).cuda().requires_grad_(True)
    b = (torch.rand(shape, generator=g).cuda() * 0.5 + 0.25 * a.detach())
    w1 = torch.tensor([np.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    w1 = (w1 / w1.sum()).unsqueeze(1)
    win = w1.mm(w1.t())[None, None].expand(shape[1], 1, 11, 11).contiguous().cuda()

    def ref(x, y):
        conv = lambda t: F.conv2d(t, win, padding=5, groups=shape[1])
        mu1, mu2 = conv(x), conv(y)
        s1, s2, s12 = conv(x * x) - mu1 ** 2, conv(y * y) - mu2 ** 2, conv(x * y) - mu1 * mu2
        return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 ** 2 + mu2 ** 2 + 1e-4) * (s1 + s2 + 9e-4))).mean()

    r = ref(a, b)
    (gr,) = torch.autograd.grad(r, a)
    s = fused_ssim(a, b)
    (gs,) = torch.autograd.grad(s, a)
    assert abs(s.item() - r.item()) < 2e-5
    assert (gs - gr).abs().sum() / gr.abs().sum() < 1e-3


@pytest.mark.parametrize("shape,clamp", [((4, 3, 96, 130), 1), ((2, 3, 512, 512), 0), ((1, 1, 7, 9), 1)])
def test_ssim_one_launch_equals_forward_then_backward(shape, clamp):
    """dimo_ssim_forward_backward (derivative planes kept on chip) against dimo_ssim_forward + dimo_ssim_backward."""
    from dimo_amd import _lib
    L, st = _lib.lib(), _lib.current_stream()
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H)
    a = (torch.rand(shape, generator=g) * 1.4 - 0.2).cuda()  # some values outside [0, 1]: the clamp path matters
    b = torch.rand(shape, generator=g).cuda()
    coef = torch.tensor([-0.37], device="cuda")
    s2, part, g2 = torch.empty(1, device="cuda"), torch.empty(3, *shape, device="cuda"), torch.empty(shape, device="cuda")
    _lib.check(L.dimo_ssim_forward(B, C, H, W, clamp, _lib.ptr(a), _lib.ptr(b), _lib.ptr(s2), _lib.ptr(part), st), "f")
    _lib.check(L.dimo_ssim_backward(B, C, H, W, clamp, _lib.ptr(a), _lib.ptr(b), _lib.ptr(part), _lib.ptr(coef),
                                    _lib.ptr(g2), st), "b")
    s1, g1 = torch.empty(1, device="cuda"), torch.empty(shape, device="cuda")
    _lib.check(L.dimo_ssim_forward_backward(B, C, H, W, clamp, _lib.ptr(a), _lib.ptr(b), _lib.ptr(coef), _lib.ptr(s1),
                                            _lib.ptr(g1), st), "fb")
    assert abs(s1.item() - s2.item()) <= 1e-5 * abs(s2.item())
    assert (g1 - g2).abs().max() <= 1e-6 * g2.abs().max() + 1e-12


def test_flat_adam_matches_torch_adam_and_skip_flag():
    """FlatAdam (one HIP launch over the flat bucket) vs torch.optim.Adam with per-group learning rates."""
    from dimo_amd.flat_adam import FlatAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (1000, 1, 3), (1000, 1), (37,), (256, 104), (5, 32)]
    lrs = [0.01, 0.0025, 0.05, 0.005, 0.0002, 0.0]
    total = sum(int(np.prod(s)) for s in shapes)
    flat = torch.randn(total, generator=g).cuda()
    grads = torch.zeros(total + 4, device="cuda")
    views, ref_params, o = [], [], 0
    for s in shapes:
        n = int(np.prod(s))
        p = torch.nn.Parameter(flat[o:o + n].view(s))
        p.grad = grads[o:o + n].view(s)
        views.append(p)
        ref_params.append(torch.nn.Parameter(flat[o:o + n].view(s).clone()))
        o += n
    groups = [{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(views, lrs))]
    opt = FlatAdam(groups, flat, grads[:total], eps=1e-15)
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref_params, lrs)], lr=0.0, eps=1e-15)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for step in range(5):
        gr = torch.randn(total, generator=g).cuda() * (10.0 ** (step - 2))
        grads[:total].copy_(gr)
        o = 0
        for p in ref_params:
            p.grad = gr[o:o + p.numel()].view(p.shape).clone()
            o += p.numel()
        if step == 3:  # change a learning rate like update_learning_rate does
            groups[0]["lr"] = 0.002
            ref.param_groups[0]["lr"] = 0.002
        opt.step(skip_flags=flag, zero_grad=True)
        ref.step()
        assert torch.count_nonzero(grads[:total]) == 0
    got = flat.cpu()
    want = torch.cat([p.detach().reshape(-1) for p in ref_params]).cpu()
    assert (got - want).abs().max() <= 2e-6 * max(1.0, want.abs().max().item())
    # skip flag: parameters and moments untouched, gradients still cleared
    before = (flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone())
    grads[:total].normal_()
    flag.fill_(1)
    opt.step(skip_flags=flag, zero_grad=True)
    assert torch.equal(flat, before[0]) and torch.equal(opt.exp_avg, before[1]) and torch.equal(opt.exp_avg_sq, before[2])
    assert torch.count_nonzero(grads[:total]) == 0


def test_wave_reduce16_of_the_blend_backward():
    """The 64-lane x 16-value halving reduction (DPP row_ror / row_shl / row_shr with bank masks, v_permlane32/16_swap,
    quad_perm: csrc/wave_ops.hpp) against a plain float64 column sum, on values of mixed sign and magnitude."""
    from dimo_amd import _lib
    g = torch.Generator().manual_seed(7)
    for trial in range(4):
        x = torch.randn(64, 16, generator=g) * (10.0 ** torch.randint(-3, 3, (64, 16), generator=g).float())
        if trial == 0:
            x = torch.arange(64 * 16, dtype=torch.float32).reshape(64, 16)  # every (lane, value) distinguishable
        xd = x.cuda().contiguous()
        out = torch.full((48,), float("nan"), device="cuda")
        _lib.check(_lib.lib().dimo_selftest_wave_reduce16(_lib.ptr(xd), _lib.ptr(out), _lib.current_stream()), "selftest")
        want = x.double().sum(0)
        got = out.cpu().double()
        # all 16 values; the 13-value and 10-value forms of the blend backward (diff_gauss / diff_gaussian_rasterization
        # flavours: their first halving step skips the fetch of values that are not there)
        for lo, used in ((0, 16), (16, 13), (32, 10)):
            assert torch.allclose(got[lo:lo + used], want[:used], rtol=1e-5,
                                  atol=1e-5 * float(x.abs().sum(0).max())), (trial, used, got, want)


def test_flat_adam_reports_to_pinned_host_memory_and_clears_the_next_steps_accumulators():
    """dimo_flat_adam_step's optional extras (round 4): `report` -- device words copied to a PINNED host slot behind a
    sequence number by the optimizer's own launch (CapacityPolicy.collect_report / poll: the instance counts reach the
    host without a copy engine or an event) -- and `zero_extra`.  The update itself must not change."""
    from dimo_amd.flat_adam import FlatAdam
    from dimo_amd.rasterizer import CapacityPolicy
    gen = torch.Generator().manual_seed(0)
    n = 4099
    res = []
    for extras in (False, True):
        p = torch.randn(n + (4 - n % 4) % 4, generator=gen.manual_seed(1)).cuda()
        gr = torch.randn(p.shape, generator=gen.manual_seed(2)).cuda()
        prm = torch.nn.Parameter(p[:n])
        opt = FlatAdam([dict(params=[prm], lr=0.01, name="x")], p, gr)
        kw = {}
        if extras:
            words = torch.tensor([[123456, 0], [654321, 0], [7, 0]], dtype=torch.int32, device="cuda")
            slot = torch.zeros(16, dtype=torch.int32, pin_memory=True)
            junk = torch.full((1000,), 3.0, device="cuda")
            kw = dict(report=(words, slot, 41), zero_extra=junk)
        opt.step(zero_grad=True, **kw)
        torch.cuda.synchronize()
        res.append(p.clone())
        assert float(gr.abs().max()) == 0.0
        if extras:
            assert int(slot[0]) == 41 and slot[1:7].tolist() == [123456, 0, 654321, 0, 7, 0]
            assert float(junk.abs().max()) == 0.0
    assert torch.equal(res[0], res[1])
    # the policy's side: collect_report -> (words, slot, seq); poll() evaluates the slot once its number is there
    pol = CapacityPolicy(initial=1000, margin=1.5)
    words = torch.tensor([[900, 0], [400, 0]], dtype=torch.int32, device="cuda")
    pol.track(words)
    tot, slot, seq = pol.collect_report()
    p = torch.zeros(8, device="cuda")
    opt = FlatAdam([dict(params=[torch.nn.Parameter(p)], lr=0.01, name="x")], p, torch.zeros(8, device="cuda"))
    opt.step(report=(tot, slot, seq))
    assert pol.poll(lag=0) == 0 and pol.last_r_max == 900 and pol.capacity >= 1350
    pol.track(torch.tensor([[10, 1]], dtype=torch.int32, device="cuda"))
    rep = pol.collect_report()
    opt.step(report=rep)
    assert pol.poll(lag=1) == 0 and pol.poll(lag=0) == 1  # (the newest report is left alone with lag = 1)
