"""Drop-in for `fused_ssim.fused_ssim(img1, img2)` (main_test_dimo.py:29,979) and for the
pure-PyTorch `src.loss.ssim` the trainer calls (src/loss.py:132-175, main_train_dimo.py:343):
mean SSIM with an 11x11 sigma-1.5 window and zero padding, differentiable w.r.t. img1.

Runs dimo_ssim_forward / dimo_ssim_forward_backward (dimo_amd/csrc/ssim.hip).  No CPU fallback.
"""
import torch

from . import _lib


class _FusedSSIM(torch.autograd.Function):
    """With a gradient wanted, value and UNIT gradient come from one launch (dimo_ssim_forward_backward) and the
    backward is a scale by the upstream scalar; without, only the value kernel runs."""

    @staticmethod
    def forward(ctx, img1, img2):
        if not img1.is_cuda:
            raise RuntimeError("dimo_amd.fused_ssim needs GPU tensors (no CPU fallback in the product path)")
        img1c, img2c = img1.float().contiguous(), img2.detach().float().contiguous()
        B, C, H, W = img1c.shape
        ssum = torch.empty(1, dtype=torch.float32, device=img1.device)
        L, st = _lib.lib(), _lib.current_stream()
        if img1.requires_grad:
            one = torch.ones(1, dtype=torch.float32, device=img1.device)
            unit = torch.empty_like(img1c)
            _lib.check(L.dimo_ssim_forward_backward(B, C, H, W, 0, _lib.ptr(img1c), _lib.ptr(img2c), _lib.ptr(one),
                                                    _lib.ptr(ssum), _lib.ptr(unit), st), "dimo_ssim_forward_backward")
            ctx.save_for_backward(unit)
        else:
            _lib.check(L.dimo_ssim_forward(B, C, H, W, 0, _lib.ptr(img1c), _lib.ptr(img2c), _lib.ptr(ssum), None, st),
                       "dimo_ssim_forward")
        return (ssum / float(B * C * H * W)).reshape(())

    @staticmethod
    def backward(ctx, g):
        (unit,) = ctx.saved_tensors
        return unit * g.float(), None


def fused_ssim(img1, img2, padding="same", train=True):
    if padding != "same":
        raise NotImplementedError("only zero 'same' padding (the reference's src/loss.py:144 semantics)")
    from .losses import materialize  # (stand-ins of queued renders: an autograd Function needs the tensors)
    img1, img2 = materialize(img1), materialize(img2)
    if img1.dim() == 3:
        img1, img2 = img1[None], img2[None]
    return _FusedSSIM.apply(img1, img2)


def ssim(img1, img2, window_size=11, size_average=True):
    """src.loss.ssim signature."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("fused path implements window_size=11, size_average=True")
    return fused_ssim(img1, img2)
