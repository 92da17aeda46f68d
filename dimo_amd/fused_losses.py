"""The image losses of a motion's batch as ONE autograd node on the fused HIP kernels (csrc/ssim.hip,
csrc/image_loss.hip): value and gradient images come out of the forward's two launches, the backward is a scale by
the upstream scalar.  Same terms and weights as `Trainer.motion_loss`'s PyTorch formulation (main_train_dimo.py:
331-372, src/loss.py:64-106, 132-175); used by the reference-shaped (autograd) step, and term by term by the drop-ins in
`dimo_amd.losses`.  GPU only.
"""
import torch

from . import _lib
from .image_loss import LOSS_WORDS, fused_image_loss


class _MotionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, depth, normal, alpha, gts, masks, w_mse, weights, lam_ssim):
        """img [B,3,H,W] (the clamped renders), depth [B,1,H,W]|None, normal [B,3,H,W]|None, alpha [B,1,H,W]; gts /
        masks: lists of B [3,H,W] / [1,H,W] tensors (or stacked tensors); w_mse: B floats (already / 3HW); weights:
        image_loss.loss_weights(...); lam_ssim: lambda_ssim x this rank's share of the motion (0: no SSIM term)."""
        if not img.is_cuda:
            raise RuntimeError("dimo_amd.fused_losses needs GPU tensors (no CPU fallback in the product path)")
        B, _, H, W = img.shape
        dev, L, st = img.device, _lib.lib(), _lib.current_stream()
        imgc, alphac = img.detach().contiguous(), alpha.detach().contiguous()
        depthc = depth.detach().contiguous() if depth is not None else None
        normalc = normal.detach().contiguous() if normal is not None else None
        if not isinstance(gts, (list, tuple)):
            gts = list(gts)
        if not isinstance(masks, (list, tuple)):
            masks = list(masks)
        gts = [t.contiguous() for t in gts]
        masks = [t.contiguous() for t in masks]
        acc = torch.zeros(LOSS_WORDS + 1, dtype=torch.float32, device=dev)
        ssim_grad = None
        if lam_ssim:
            ssim_grad = torch.empty_like(imgc)
            coef = torch.full((1,), -lam_ssim, dtype=torch.float32, device=dev)
            ssum = acc[LOSS_WORDS:]
            if B <= 32:
                _lib.check(L.dimo_ssim_forward_backward_images(B, 3, H, W, 1 | 2, _lib.ptr(imgc), _lib.ptr_array(gts),
                                                               _lib.ptr(coef), _lib.ptr(ssum), _lib.ptr(ssim_grad), st),
                           "dimo_ssim_forward_backward_images")
            else:
                _lib.check(L.dimo_ssim_forward_backward(B, 3, H, W, 1 | 2, _lib.ptr(imgc), _lib.ptr(torch.stack(gts)),
                                                        _lib.ptr(coef), _lib.ptr(ssum), _lib.ptr(ssim_grad), st),
                           "dimo_ssim_forward_backward")
        gi, gd, gn, ga = fused_image_loss(imgc, depthc, normalc, alphac, gts, masks, list(w_mse), weights, ssim_grad,
                                          acc[:LOSS_WORDS], stream=st)
        loss = acc[:LOSS_WORDS].sum()
        if lam_ssim:
            loss = loss + lam_ssim * (1 - acc[LOSS_WORDS] / float(B * 3 * H * W))
        ctx.save_for_backward(gi, gd if gd is not None else gi.new_empty(0), gn if gn is not None else gi.new_empty(0), ga)
        ctx.has = (gd is not None, gn is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        gi, gd, gn, ga = ctx.saved_tensors
        return (gi * g, gd * g if ctx.has[0] else None, gn * g if ctx.has[1] else None, ga * g,
                None, None, None, None, None)


def motion_loss(img, depth, normal, alpha, gts, masks, w_mse, weights, lam_ssim):
    return _MotionLossFn.apply(img, depth, normal, alpha, gts, masks, w_mse, weights, lam_ssim)


_ZERO = dict(w_mask=0.0, w_smooth_x=0.0, w_smooth_y=0.0, w_bilat_x=0.0, w_bilat_y=0.0)
_CONST = {}


def _nchw(x):  # [B,H,W,C] (the reference's layout for the smoothness terms) -> [B,C,H,W]
    return x.permute(0, 3, 1, 2)


def _consts(B, H, W, dev):
    """Read-only zeros standing in for the terms a smoothness-only call does not have (alpha, mask) and a scratch
    plane for the alpha gradient nobody reads: made once per shape instead of two memsets per call."""
    key = (B, H, W, str(dev))
    c = _CONST.get(key)
    if c is None:
        if len(_CONST) > 32:
            _CONST.clear()
        c = _CONST[key] = (torch.zeros(B, 1, H, W, dtype=torch.float32, device=dev),
                           torch.zeros(1, H, W, dtype=torch.float32, device=dev),
                           torch.empty(B, 1, H, W, dtype=torch.float32, device=dev))
    return c


class _SmoothFn(torch.autograd.Function):
    """One smoothness term (depth: src/loss.py:64-83, normal: :86-106) on the fused image-loss kernel with every other
    weight zero: value and both gradient images from one launch."""

    @staticmethod
    def forward(ctx, img, x, is_depth, weights):
        if not img.is_cuda:
            raise RuntimeError("dimo_amd.fused_losses needs GPU tensors (no CPU fallback in the product path)")
        B, _, H, W = img.shape
        imgc, xc = img.detach().contiguous(), x.detach().contiguous()
        zero_alpha, zero_mask, scratch = _consts(B, H, W, img.device)
        acc = torch.zeros(LOSS_WORDS, dtype=torch.float32, device=img.device)
        g_img, g_x = torch.empty_like(imgc), torch.empty_like(xc)
        fused_image_loss(imgc, xc if is_depth else None, None if is_depth else xc, zero_alpha, imgc, zero_mask,
                         [0.0] * B, weights, None, acc,
                         out=(g_img, g_x if is_depth else None, None if is_depth else g_x, scratch))
        ctx.save_for_backward(g_img, g_x)
        return acc.sum()

    @staticmethod
    def backward(ctx, g):
        g_img, g_x = ctx.saved_tensors
        return g_img * g, g_x * g, None, None


def edge_aware_smoothness(depth_bhwc, rgb_bhwc):
    """src/loss.py:64-83 on the fused kernel: mean |d depth| exp(-mean_c |d rgb|) along x plus along y, differentiable
    w.r.t. depth and rgb.  depth [B,H,W,1], rgb [B,H,W,3] with values in [0, 1]."""
    B, H, W, _ = depth_bhwc.shape
    w = dict(_ZERO, w_smooth_x=1.0 / (B * H * (W - 1)) if W > 1 else 0.0,
             w_smooth_y=1.0 / (B * (H - 1) * W) if H > 1 else 0.0)
    return _SmoothFn.apply(_nchw(rgb_bhwc), _nchw(depth_bhwc), True, w)


def bilateral_normal_smoothness(normal_bhwc, rgb_bhwc):
    """src/loss.py:86-106 on the fused kernel: mean sqrt(1 + (|d n| exp(-3 mean_c |d rgb|))^2) along x plus along y."""
    B, H, W, _ = normal_bhwc.shape
    w = dict(_ZERO, w_bilat_x=1.0 / (3 * B * H * (W - 1)) if W > 1 else 0.0,
             w_bilat_y=1.0 / (3 * B * (H - 1) * W) if H > 1 else 0.0)
    return _SmoothFn.apply(_nchw(rgb_bhwc), _nchw(normal_bhwc), False, w)
