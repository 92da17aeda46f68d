"""Image losses of the training step (main_train_dimo.py:328-372).

  ssim                                      -> fused HIP kernel (dimo_amd/csrc/ssim.hip); replaces src/loss.py:132-175
  compute_edge_aware_smoothness_loss        src/loss.py:64-83
  compute_bilateral_normal_smoothness_loss  src/loss.py:86-106

The two smoothness terms are short elementwise chains on [b^2, H, W, C] tensors and stay on
PyTorch-ROCm; SSIM (five 11x11 grouped convolutions in the reference) is the fused kernel.
"""
import torch

from .fused_ssim import ssim  # noqa: F401  (GPU only, no CPU fallback)


def _image_gradients(rgb):
    gx = torch.mean(torch.abs(rgb[..., :, :-1, :] - rgb[..., :, 1:, :]), -1, keepdim=True)
    gy = torch.mean(torch.abs(rgb[..., :-1, :, :] - rgb[..., 1:, :, :]), -1, keepdim=True)
    return gx, gy


def compute_edge_aware_smoothness_loss(depth, rgb):
    """depth [B,H,W,1], rgb [B,H,W,3]: mean |d depth| * exp(-mean_c |d rgb|) along x plus along y."""
    gx, gy = _image_gradients(rgb)
    dx = torch.abs(depth[..., :, :-1, :] - depth[..., :, 1:, :]) * torch.exp(-gx)
    dy = torch.abs(depth[..., :-1, :, :] - depth[..., 1:, :, :]) * torch.exp(-gy)
    return dx.mean() + dy.mean()


def compute_bilateral_normal_smoothness_loss(normal, rgb):
    """normal [B,H,W,3], rgb [B,H,W,3]: mean sqrt(1 + (|d n| exp(-3 mean_c |d rgb|))^2) along x plus along y."""
    gx, gy = _image_gradients(rgb)
    nx = torch.abs(normal[..., :, :-1, :] - normal[..., :, 1:, :]) * torch.exp(-3 * gx)
    ny = torch.abs(normal[..., :-1, :, :] - normal[..., 1:, :, :]) * torch.exp(-3 * gy)
    return torch.sqrt(1 + nx ** 2).mean() + torch.sqrt(1 + ny ** 2).mean()
