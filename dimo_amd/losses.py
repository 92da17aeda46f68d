"""Image losses of the training step (main_train_dimo.py:328-372).

  ssim                                      -> fused HIP kernel (dimo_amd/csrc/ssim.hip); replaces src/loss.py:132-175
  compute_edge_aware_smoothness_loss        src/loss.py:64-83
  compute_bilateral_normal_smoothness_loss  src/loss.py:86-106

On the GPU all three run on the fused HIP kernels (dimo_amd/fused_losses.py: value and gradient in one launch; SSIM is
five 11x11 grouped convolutions in the reference, each smoothness term a chain of ~25 elementwise launches).  The
fused smoothness kernels read rgb as clamp(rgb, 0, 1) -- the reference passes the clamped render
(main_train_dimo.py:363-372 after latent_gs_renderer.py:1279); `assume_unit_range=False` keeps the PyTorch formulation
for any other input.  CPU tensors always take the PyTorch formulation (the oracle of the tests).
"""
import torch

from .fused_ssim import ssim  # noqa: F401  (GPU only, no CPU fallback)


def _image_gradients(rgb):
    gx = torch.mean(torch.abs(rgb[..., :, :-1, :] - rgb[..., :, 1:, :]), -1, keepdim=True)
    gy = torch.mean(torch.abs(rgb[..., :-1, :, :] - rgb[..., 1:, :, :]), -1, keepdim=True)
    return gx, gy


def compute_edge_aware_smoothness_loss(depth, rgb, assume_unit_range=True):
    """depth [B,H,W,1], rgb [B,H,W,3]: mean |d depth| * exp(-mean_c |d rgb|) along x plus along y."""
    if assume_unit_range and depth.is_cuda and depth.dim() == 4 and depth.dtype == torch.float32:
        from .fused_losses import edge_aware_smoothness
        return edge_aware_smoothness(depth, rgb)
    gx, gy = _image_gradients(rgb)
    dx = torch.abs(depth[..., :, :-1, :] - depth[..., :, 1:, :]) * torch.exp(-gx)
    dy = torch.abs(depth[..., :-1, :, :] - depth[..., 1:, :, :]) * torch.exp(-gy)
    return dx.mean() + dy.mean()


def compute_bilateral_normal_smoothness_loss(normal, rgb, assume_unit_range=True):
    """normal [B,H,W,3], rgb [B,H,W,3]: mean sqrt(1 + (|d n| exp(-3 mean_c |d rgb|))^2) along x plus along y."""
    if assume_unit_range and normal.is_cuda and normal.dim() == 4 and normal.dtype == torch.float32:
        from .fused_losses import bilateral_normal_smoothness
        return bilateral_normal_smoothness(normal, rgb)
    gx, gy = _image_gradients(rgb)
    nx = torch.abs(normal[..., :, :-1, :] - normal[..., :, 1:, :]) * torch.exp(-3 * gx)
    ny = torch.abs(normal[..., :-1, :, :] - normal[..., 1:, :, :]) * torch.exp(-3 * gy)
    return torch.sqrt(1 + nx ** 2).mean() + torch.sqrt(1 + ny ** 2).mean()
