"""The skinning kernels (dimo_amd/csrc/deform.hip + deform_body.hpp + wave_ops.hpp) and the flat Adam step (adam.hip)
run on the CPU SIMT emulation (tests/simt/) against the oracle / torch, with the tolerances tests/test_gpu_deform.py and
tests/test_gpu_ops.py hold the GPU to.  The DPP builtins are emulated lane for lane; wave_reduce16's asm block runs as
the shim's instruction-for-instruction spelling of it (v_add_f32_dpp with row / bank masks, v_permlane32/16_swap), which
pins the claim the kernels rest on: lane l ends with the wave total of value reduce16_slot(l).  Same source text as the
product, no GPU; the GPU tests stay the parity tests proper."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.deform_ref import skinning_ref
from tests.simt import build as simt_build
from tests.simt import harness as hz

_D = None
NAMES = ("xyz", "rotation", "scaling", "opacity", "c_xyz", "c_log_radius", "d_xyz", "d_rot")
TOL = 1e-4


def D():
    global _D
    if _D is None:
        lib = C.CDLL(simt_build.build(target="deform"))
        p, i, f, z, q = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64
        lib.dimo_deform_forward.argtypes = [i, i, i] + [p] * 15
        lib.dimo_deform_backward.argtypes = [i, i, i, i] + [p] * 23 + [z, p]
        lib.dimo_deform_backward_scratch_bytes.argtypes = [i, i]
        lib.dimo_deform_backward_scratch_bytes.restype = z
        lib.dimo_flat_adam_step.argtypes = [q, p, p, p, p, i, p, p, f, f, f, q, p, i, i, i, p, p, i, p, C.c_uint32, p, q, q, q, i, p]
        lib.simt_wave_reduce16.argtypes = [p, p]
        lib.simt_wave_scatter.argtypes = [p, i, p, p, p, i]
        _D = lib
    return _D


def _rand_inputs(N, M, seed, clustered=False):
    g = torch.Generator().manual_seed(seed)
    c_xyz = torch.rand(M, 3, generator=g) - 0.5
    xyz = torch.rand(N, 3, generator=g) - 0.5
    if clustered:  # neighbouring Gaussians share their control points (the Morton-ordered case the matching is for)
        xyz = xyz[torch.argsort((xyz[:, 0] * 4).floor() * 16 + (xyz[:, 1] * 4).floor() * 4 + (xyz[:, 2] * 4).floor())]
    nn_dist, nn_idx = torch.topk(torch.cdist(xyz, c_xyz), min(4, M), dim=1, largest=False)
    if M < 4:
        nn_dist = torch.cat([nn_dist] + [nn_dist[:, -1:]] * (4 - M), 1)
        nn_idx = torch.cat([nn_idx] + [nn_idx[:, -1:]] * (4 - M), 1)
    return dict(xyz=xyz.contiguous(), rotation=torch.randn(N, 4, generator=g), scaling=torch.randn(N, 3, generator=g) - 3,
                opacity=torch.randn(N, 1, generator=g), c_xyz=c_xyz,
                c_log_radius=torch.log(torch.rand(M, 1, generator=g) * 0.2 + 0.05),
                d_xyz=torch.randn(M, 3, generator=g) * 0.05,
                d_rot=torch.tensor([1.0, 0, 0, 0]) + 0.3 * torch.randn(M, 4, generator=g),
                nn_dist=nn_dist.contiguous(), nn_idx=nn_idx.contiguous())


def _np(t, dt=np.float32):
    return np.ascontiguousarray(t.detach().numpy(), dtype=dt)


def _emulated(a, local_frame, w):
    """forward outputs and the gradients of sum(out * w) from the emulated kernels"""
    N, M = a["xyz"].shape[0], a["c_xyz"].shape[0]
    ins = [_np(a[k]) for k in NAMES] + [_np(a["nn_dist"]), _np(a["nn_idx"], np.int64)]
    outs = [np.full(s, np.nan, np.float32) for s in ((N, 3), (N, 4), (N, 3), (N, 1))]
    ptr = lambda x: x.ctypes.data
    rc = D().dimo_deform_forward(N, M, int(local_frame), *[ptr(x) for x in ins], *[ptr(x) for x in outs], None)
    assert rc == 0
    grads = [np.full(_np(a[k]).shape, np.nan, np.float32) for k in NAMES]
    nb = D().dimo_deform_backward_scratch_bytes(N, M)
    scratch = hz.workspace(max(nb, 16), 0x5A)
    rc = D().dimo_deform_backward(N, M, int(local_frame), 0, *[ptr(x) for x in ins], *[ptr(_np(x)) for x in w],
                                  *[ptr(x) for x in grads], ptr(scratch), nb, None)
    assert rc == 0
    return outs, dict(zip(NAMES, grads))


def _compare(a, local_frame=True, seed=0):
    cpu = {k: (v.clone().double().requires_grad_(True) if k in NAMES else v) for k, v in a.items()}
    cpu["nn_dist"] = a["nn_dist"].double()
    ref = skinning_ref(**cpu, local_frame=local_frame)
    g = torch.Generator().manual_seed(seed + 1)
    w = [torch.randn(r.shape, generator=g) for r in ref]
    sum((r * x.double()).sum() for r, x in zip(ref, w)).backward()
    outs, grads = _emulated(a, local_frame, w)
    for r, o in zip(ref, outs):
        err = np.abs(o.astype(np.float64) - r.detach().numpy()).mean()
        assert err <= TOL * max(1.0, r.detach().abs().mean().item()), err
    for k in NAMES:
        r, o = cpu[k].grad, grads[k].astype(np.float64)
        if r is None:  # input unused by this variant (c_xyz when local_frame=False): the kernel writes zeros
            assert np.abs(o).max() == 0, k
            continue
        rel = np.abs(o - r.numpy().reshape(o.shape)).sum() / (r.abs().sum().item() + 1e-12)
        assert rel <= TOL, (k, rel)


@pytest.mark.parametrize("N,M,clustered", [(1000, 32, False), (777, 5, False), (3000, 200, True), (700, 1500, False),
                                           (64, 3, False)])
def test_emulated_skinning_vs_oracle(N, M, clustered):
    _compare(_rand_inputs(N, M, seed=N + M, clustered=clustered))


def test_emulated_skinning_global_frame_variant():
    _compare(_rand_inputs(1500, 64, seed=3), local_frame=False)


def test_emulated_skinning_empty_model_and_bad_arguments():
    a = _rand_inputs(8, 4, seed=1)
    ins = [_np(a[k]) for k in NAMES] + [_np(a["nn_dist"]), _np(a["nn_idx"], np.int64)]
    ptr = lambda x: x.ctypes.data
    outs = [np.zeros(s, np.float32) for s in ((8, 3), (8, 4), (8, 3), (8, 1))]
    assert D().dimo_deform_forward(0, 4, 1, *[ptr(x) for x in ins], *[ptr(x) for x in outs], None) == 0
    assert D().dimo_deform_forward(8, 0, 1, *[ptr(x) for x in ins], *[ptr(x) for x in outs], None) != 0
    assert D().dimo_deform_forward(8, 1 << 20, 1, *[ptr(x) for x in ins], *[ptr(x) for x in outs], None) != 0
    assert D().dimo_deform_forward(8, 4, 1, *([None] + [ptr(x) for x in ins[1:]]), *[ptr(x) for x in outs], None) != 0


def test_emulated_wave_reduce16_slots():
    """Lane l ends with the wave total of value reduce16_slot(l) -- all 16 values, and the 13- / 10-value forms."""
    g = np.random.default_rng(7)
    for trial in range(3):
        x = (g.standard_normal((64, 16)) * 10.0 ** g.integers(-3, 3, (64, 16))).astype(np.float32)
        if trial == 0:
            x = np.arange(64 * 16, dtype=np.float32).reshape(64, 16)  # every (lane, value) distinguishable, sums exact
        out = np.full(48, np.nan, np.float32)
        D().simt_wave_reduce16(x.ctypes.data, out.ctypes.data)
        want = x.astype(np.float64).sum(0)
        for lo, used in ((0, 16), (16, 13), (32, 10)):
            np.testing.assert_allclose(out[lo:lo + used], want[:used], rtol=1e-5, atol=1e-5 * np.abs(x).sum(0).max())
        if trial == 0:
            assert np.array_equal(out[:16].astype(np.float64), want)


@pytest.mark.parametrize("matched", [1, 0])
@pytest.mark.parametrize("pattern", ["two", "runs", "many", "all_same", "some_invalid", "none_valid"])
def test_emulated_wave_scatter_add(pattern, matched):
    """The two scatter-adds of the skinning backward on one wave, for the index patterns that take their different
    paths: few distinct indices (matched pairs), more than 2 x MATCH_ROUNDS (the run-combined fallback), runs inside
    DPP rows, invalid lanes."""
    g = np.random.default_rng(hash(pattern) % 1000)
    rows, stride = 40, 11
    idx = {"two": g.integers(0, 2, 64) * 7, "runs": np.repeat(g.integers(0, rows, 16), 4),
           "many": g.integers(0, rows, 64), "all_same": np.full(64, 5),
           "some_invalid": g.integers(0, 6, 64), "none_valid": g.integers(0, rows, 64)}[pattern].astype(np.int32)
    valid = np.ones(64, np.int32)
    if pattern == "some_invalid":
        valid = (g.random(64) < 0.6).astype(np.int32)
    if pattern == "none_valid":
        valid[:] = 0
    vals = g.standard_normal((64, 8)).astype(np.float32)
    table = g.standard_normal((rows, stride)).astype(np.float32)
    want = table.astype(np.float64)
    for l in range(64):
        if valid[l]:
            want[idx[l], :8] += vals[l]
    D().simt_wave_scatter(table.ctypes.data, stride, idx.ctypes.data, vals.ctypes.data, valid.ctypes.data, matched)
    np.testing.assert_allclose(table, want, rtol=1e-5, atol=2e-5)


def test_emulated_flat_adam_matches_torch_adam_and_skip_flag():
    """dimo_flat_adam_step on the emulation vs torch.optim.Adam with per-segment learning rates (the GPU test's
    tolerance), a learning-rate change, the skip flag, the gradient clear and the two-launch split of a step."""
    g = torch.Generator().manual_seed(0)
    sizes, lrs = [3000, 3000, 1000, 39, 2600, 160], [0.01, 0.0025, 0.05, 0.005, 0.0002, 0.0]
    total = sum(sizes)
    flat = _np(torch.randn(total + 3, generator=g))[:total]  # (16-byte alignment: numpy's allocations are)
    # (every bucket is followed by eight sentinel floats: total = 9799 = 3 mod 4, the kernel's float4 path must
    # stop in front of the ragged tail and the tail loop at n -- a mutant of either bound wrote one element too far and
    # went unnoticed: tools/mutate_emulated.py)
    SENT = np.float32(-12345.0)
    bufs = [np.full(total + 8, SENT, np.float32) for _ in range(4)]
    p, m, v, grads = (b[:total] for b in bufs)
    p[:], m[:], v[:], grads[:] = flat, 0.0, 0.0, 0.0
    ref_params = [torch.nn.Parameter(torch.from_numpy(flat[o - n:o].copy())) for n, o in zip(sizes, np.cumsum(sizes))]
    ref = torch.optim.Adam([{"params": [q], "lr": lr} for q, lr in zip(ref_params, lrs)], lr=0.0, eps=1e-15)
    ends = np.cumsum(sizes).astype(np.int64)
    lr = np.array(lrs, np.float32)
    flag = np.zeros(1, np.int32)
    skipped = np.zeros(2, np.int32)  # (two words, written alternately)
    ptr = lambda x: x.ctypes.data

    def step(k, lo=0, hi=0, final=1):
        return D().dimo_flat_adam_step(total, ptr(p), ptr(grads), ptr(m), ptr(v), len(sizes), ptr(ends), ptr(lr), 0.9,
                                       0.999, 1e-15, k, ptr(flag), 1, 0, 1, ptr(skipped), None, 0, None, 0, None, 0,
                                       lo, hi, final, None)
    for k in range(5):
        gr = torch.randn(total, generator=g) * (10.0 ** (k - 2))
        grads[:] = gr.numpy()
        o = 0
        for q in ref_params:
            q.grad = gr[o:o + q.numel()].clone()
            o += q.numel()
        if k == 3:
            lr[0] = 0.002
            ref.param_groups[0]["lr"] = 0.002
        if k == 2:  # one step as two launches (the early part runs next to the TimeNet backward in the product)
            assert step(k + 1, 0, 4000, 0) == 0 and step(k + 1, 4000, total, 1) == 0
        else:
            assert step(k + 1) == 0
        ref.step()
        assert np.count_nonzero(grads) == 0
    assert all((b[total:] == SENT).all() for b in bufs), "a write past the end of a bucket"
    want = torch.cat([q.detach() for q in ref_params]).numpy()
    assert np.abs(p - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    before = (p.copy(), m.copy(), v.copy())
    grads[:] = 1.0
    flag[0] = 1
    assert step(6) == 0
    assert np.array_equal(p, before[0]) and np.array_equal(m, before[1]) and np.array_equal(v, before[2])
    assert np.count_nonzero(grads) == 0 and skipped.max() == 1


def test_emulated_flat_adam_report_and_next_steps_accumulators():
    """dimo_flat_adam_step's optional extras (tests/test_gpu_ops.py holds the GPU to the same): `report` -- device words
    copied behind a sequence number into a host-visible slot by the FINAL part of a step only --, and a scratch region
    zeroed for the next step."""
    n = 4096
    p, g, m, v = (np.zeros(n, np.float32) for _ in range(4))
    g[:] = 1.0
    ends, lr = np.array([n], np.int64), np.array([0.1], np.float32)
    src = np.array([1234, 1], np.uint32)
    dst = np.zeros(3, np.uint32)
    extra = np.ones(1000, np.float32)
    ptr = lambda x: x.ctypes.data

    def step(lo, hi, final, seq):
        return D().dimo_flat_adam_step(n, ptr(p), ptr(g), ptr(m), ptr(v), 1, ptr(ends), ptr(lr), 0.9, 0.999, 1e-15, 1,
                                       None, 0, 0, 1, None, ptr(src), 2, ptr(dst), seq, ptr(extra), extra.size, lo, hi,
                                       final, None)
    assert step(0, 2048, 0, 7) == 0  # the early part of a two-launch step: no report, nothing zeroed
    assert dst[0] == 0 and (extra == 1.0).all() and (p[:2048] != 0).all() and (p[2048:] == 0).all()
    assert step(2048, n, 1, 7) == 0
    assert list(dst) == [7, 1234, 1] and (extra == 0.0).all() and (p != 0).all() and (g == 0).all()
