"""The image-loss kernels (dimo_amd/csrc/ssim.hip, image_loss.hip, loss_terms.hpp as hipcc compiles them) run on the
CPU SIMT emulation (tests/simt/) against the float64 restatement of the reference's loss assembly
(oracle/losses_ref.py; main_train_dimo.py:331-372, src/loss.py:64-106,132-175) -- the inputs, ragged shapes and
tolerances of tests/test_gpu_losses.py: the two-kernel path (SSIM forward + backward, then dimo_image_loss), the
one-pass kernel (dimo_ssim_image_loss, the default of the step) with per-image target / mask pointer lists, and the fused
SSIM forward+backward.  No GPU; the GPU tests stay the parity tests proper."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.losses_ref import motion_loss_ref
from tests.simt import build as simt_build

_L = None
SHAPES = [(4, 64, 48, 1.0, (True, True)), (1, 33, 70, 0.5, (True, False)), (3, 96, 80, 0.75, (False, True)),
          (2, 40, 40, 1.0, (False, False)),
          # a wave owns 62 columns x 8 rows: one-column last strip, one-row last block
          (2, 17, 125, 1.0, (True, True)), (1, 9, 63, 1.0, (True, True)), (1, 2, 300, 1.0, (True, True)),
          (1, 70, 2, 1.0, (True, True))]


def Lz():
    global _L
    if _L is None:
        lib = C.CDLL(simt_build.build(target="losses"))
        p, f, i = C.c_void_p, C.c_float, C.c_int
        lib.dimo_ssim_forward.argtypes = [i, i, i, i, i, p, p, p, p, p]
        lib.dimo_ssim_backward.argtypes = [i, i, i, i, i, p, p, p, p, p, p]
        lib.dimo_ssim_forward_backward.argtypes = [i, i, i, i, i, p, p, p, p, p, p]
        lib.dimo_image_loss.argtypes = [i, i, i, p, p, p, p, p, p, i, p, f, f, f, f, f, p, p, p, p, p, p, p, p, p, p]
        lib.dimo_ssim_image_loss.argtypes = [i, i, i, p, p, p, p, p, p, i, p, f, f, f, f, f, p, p, p, p, p, p, p, p, p, p, p]
        _L = lib
    return _L


def _cfg(dn):
    from dimo_amd.trainer import TrainConfig
    return TrainConfig(add_depth=dn[0], add_normal=dn[1])


def _lam(cfg):
    return dict(mse=cfg.lambda_mse, ssim=cfg.lambda_ssim, mask=cfg.lambda_mask, smooth=cfg.lambda_smooth,
                bilateral=cfg.lambda_bilateral)


def _inputs(B, H, W, share, dn):
    cfg = _cfg(dn)
    g = torch.Generator().manual_seed(B * H + W)
    image = torch.rand(B, 3, H, W, generator=g) * 1.4 - 0.2  # values outside [0,1] exercise the clamp mask
    image[:, :, :4] = 1.0  # exact boundary values (white background) must pass gradient like torch.clamp
    image[:, :, 4:6] = 0.0  # ... and the lower boundary (a mutant that closed it went unnoticed: tools/mutate_emulated.py)
    depth = torch.rand(B, 1, H, W, generator=g) * 2
    normal = torch.randn(B, 3, H, W, generator=g)
    alpha = torch.rand(B, 1, H, W, generator=g)
    gt = torch.rand(B, 3, H, W, generator=g)
    mask = (torch.rand(1, H, W, generator=g) > 0.5).float()
    wts = [1.0 if b % 2 == 0 else 0.5 for b in range(B)]
    n_img = round(B / share)
    leaves = [t.clone().double().requires_grad_(True) for t in (image, depth, normal, alpha)]
    ref = motion_loss_ref(leaves[0], leaves[1] if dn[0] else None, leaves[2] if dn[1] else None, leaves[3],
                          gt.double(), mask.double(), wts, _lam(cfg), share=B / n_img)
    ref.backward()
    n = lambda t: np.ascontiguousarray(t.numpy(), np.float32)
    return cfg, n_img, wts, [n(x) for x in (image, depth, normal, alpha, gt, mask)], leaves, ref.item()


def _check(cfg, B, H, W, n_img, dn, arrs, leaves, ref, acc, ssum, grads, gdot):
    image, depth, normal, alpha = arrs[:4]
    gi, gd, gn, ga = grads
    want = (gi * image).sum(1, keepdims=True) + ga * alpha
    if gd is not None:
        want = want + gd * depth
    if gn is not None:
        want = want + (gn * normal).sum(1, keepdims=True)
    assert np.isfinite(gdot).all()
    assert np.abs(gdot - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-12) + 1e-12
    loss = float(acc.astype(np.float64).sum()) + cfg.lambda_ssim * (B / n_img) * (1 - float(ssum[0]) / (B * 3 * H * W))
    assert abs(loss - ref) <= 2e-5 * abs(ref), (loss, ref)
    for got, leaf, name in ((gi, leaves[0], "image"), (gd, leaves[1], "depth"), (gn, leaves[2], "normal"),
                            (ga, leaves[3], "alpha")):
        if got is None:
            assert leaf.grad is None
            continue
        r = leaf.grad.numpy()
        rel = np.abs(got.astype(np.float64) - r).sum() / (np.abs(r).sum() + 1e-12)
        assert rel <= 1e-4, (name, rel)


def _ptr(a):
    return None if a is None else a.ctypes.data


def _weights(cfg, B, n_img, H, W):
    from dimo_amd.image_loss import loss_weights
    w = loss_weights(cfg, B, n_img, H, W)
    return [w[k] for k in ("w_mask", "w_smooth_x", "w_smooth_y", "w_bilat_x", "w_bilat_y")]


@pytest.mark.parametrize("B,H,W,share,dn", SHAPES)
def test_emulated_two_kernel_losses_vs_reference_assembly(B, H, W, share, dn):
    cfg, n_img, wts, arrs, leaves, ref = _inputs(B, H, W, share, dn)
    image, depth, normal, alpha, gt, mask = arrs
    ssum = np.full(1, np.nan, np.float32)
    partials = np.full((3, B, 3, H, W), np.nan, np.float32)
    assert Lz().dimo_ssim_forward(B, 3, H, W, 1, _ptr(image), _ptr(gt), _ptr(ssum), _ptr(partials), None) == 0
    coef = np.array([-cfg.lambda_ssim * B / n_img], np.float32)
    sg = np.full((B, 3, H, W), np.nan, np.float32)
    assert Lz().dimo_ssim_backward(B, 3, H, W, 1, _ptr(image), _ptr(gt), _ptr(partials), _ptr(coef), _ptr(sg), None) == 0
    # (the fused forward + backward must give the same sum and gradient image)
    ssum2, sg2 = np.full(1, np.nan, np.float32), np.full((B, 3, H, W), np.nan, np.float32)
    assert Lz().dimo_ssim_forward_backward(B, 3, H, W, 1, _ptr(image), _ptr(gt), _ptr(coef), _ptr(ssum2), _ptr(sg2), None) == 0
    assert abs(ssum2[0] - ssum[0]) <= 1e-5 * abs(ssum[0])
    assert np.abs(sg2 - sg).sum() <= 1e-5 * np.abs(sg).sum()
    acc = np.zeros(512, np.float32)
    w_mse = (C.c_float * B)(*[cfg.lambda_mse * w / (3 * H * W) for w in wts])
    gdot = np.full((B, 1, H, W), np.nan, np.float32)
    gi, ga = np.full_like(image, np.nan), np.full_like(alpha, np.nan)
    gd = np.full_like(depth, np.nan) if dn[0] else None
    gn = np.full_like(normal, np.nan) if dn[1] else None
    rc = Lz().dimo_image_loss(B, H, W, _ptr(image), _ptr(depth) if dn[0] else None, _ptr(normal) if dn[1] else None,
                              _ptr(alpha), _ptr(gt), _ptr(mask), 0, w_mse, *_weights(cfg, B, n_img, H, W), _ptr(sg),
                              _ptr(acc), _ptr(gi), _ptr(gd), _ptr(gn), _ptr(ga), _ptr(gdot), None, None, None)
    assert rc == 0
    _check(cfg, B, H, W, n_img, dn, arrs, leaves, ref, acc, ssum, (gi, gd, gn, ga), gdot)


@pytest.mark.parametrize("B,H,W,share,dn", SHAPES)
def test_emulated_one_pass_ssim_and_image_losses_vs_reference_assembly(B, H, W, share, dn):
    cfg, n_img, wts, arrs, leaves, ref = _inputs(B, H, W, share, dn)
    image, depth, normal, alpha, gt, mask = arrs
    ssum = np.zeros(1, np.float32)
    coef = np.array([-cfg.lambda_ssim * B / n_img], np.float32)
    acc = np.zeros(512, np.float32)
    w_mse = (C.c_float * B)(*[cfg.lambda_mse * w / (3 * H * W) for w in wts])
    gdot = np.full((B, 1, H, W), np.nan, np.float32)
    gi, ga = np.full_like(image, np.nan), np.full_like(alpha, np.nan)
    gd = np.full_like(depth, np.nan) if dn[0] else None
    gn = np.full_like(normal, np.nan) if dn[1] else None
    gts = [np.ascontiguousarray(gt[b]) for b in range(B)]  # per-image pointer lists, as the trainer hands them over
    masks = [mask.copy() for _ in range(B)]
    gt_list = (C.c_void_p * B)(*[x.ctypes.data for x in gts])
    mask_list = (C.c_void_p * B)(*[x.ctypes.data for x in masks])
    rc = Lz().dimo_ssim_image_loss(B, H, W, _ptr(image), _ptr(depth) if dn[0] else None, _ptr(normal) if dn[1] else None,
                                   _ptr(alpha), None, None, 0, w_mse, *_weights(cfg, B, n_img, H, W), _ptr(coef),
                                   _ptr(ssum), _ptr(acc), _ptr(gi), _ptr(gd), _ptr(gn), _ptr(ga), _ptr(gdot), gt_list,
                                   mask_list, None)
    assert rc == 0
    _check(cfg, B, H, W, n_img, dn, arrs, leaves, ref, acc, ssum, (gi, gd, gn, ga), gdot)


@pytest.mark.parametrize("order", ["reverse", "random:7"])
def test_emulated_one_pass_losses_under_other_fiber_schedules(order, monkeypatch):
    monkeypatch.setenv("SIMT_ORDER", order)
    test_emulated_one_pass_ssim_and_image_losses_vs_reference_assembly(2, 17, 125, 1.0, (True, True))
    test_emulated_two_kernel_losses_vs_reference_assembly(3, 96, 80, 0.75, (False, True))
