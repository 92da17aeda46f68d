"""Image losses of the training step (main_train_dimo.py:328-372).

  ssim                                      -> fused HIP kernel (dimo_amd/csrc/ssim.hip); replaces src/loss.py:132-175
  compute_edge_aware_smoothness_loss        src/loss.py:64-83
  compute_bilateral_normal_smoothness_loss  src/loss.py:86-106

On the GPU all three can run on the fused HIP kernels (dimo_amd/fused_losses.py: value and gradient in one launch; SSIM
is five 11x11 grouped convolutions in the reference, each smoothness term a chain of ~25 elementwise launches).  The
fused smoothness kernels read rgb as clamp(rgb, 0, 1) -- the reference passes the clamped render
(main_train_dimo.py:363-372 after latent_gs_renderer.py:1279) -- so they are used only when rgb is known to lie in
[0, 1]: `assume_unit_range=True`, or (default, None) rgb is the image `Renderer.render` returned or a view /
concatenation of it.  Any other input, dtype or layout keeps the PyTorch formulation with the reference's exact
semantics.  CPU tensors always take the PyTorch formulation (the oracle of the tests).
"""
import torch

from .fused_ssim import ssim  # noqa: F401  (GPU only, no CPU fallback)


def materialize(x):
    """The tensor behind a `dimo_amd.batched_render.LazyTensor` (identity for tensors)."""
    m = getattr(x, "materialize", None)
    return m() if m is not None and not isinstance(x, torch.Tensor) else x


def _image_gradients(rgb):
    gx = torch.mean(torch.abs(rgb[..., :, :-1, :] - rgb[..., :, 1:, :]), -1, keepdim=True)
    gy = torch.mean(torch.abs(rgb[..., :-1, :, :] - rgb[..., 1:, :, :]), -1, keepdim=True)
    return gx, gy


def _fused_ok(x, rgb, channels, assume_unit_range):
    """The fused kernel reads rgb as clamp(rgb, 0, 1): it stands in for the PyTorch formulation only when rgb is KNOWN
    to lie in [0, 1] -- the caller says so, or rgb is (a view / concatenation of) the clamped image `Renderer.render`
    returned (`LazyTensor.unit_range`) -- and the operands have the layout it is written for."""
    if assume_unit_range is None:
        assume_unit_range = bool(getattr(rgb, "unit_range", False))
    return bool(assume_unit_range and x.is_cuda and x.dim() == 4 and rgb.dim() == 4 and x.dtype == torch.float32
                and rgb.dtype == torch.float32 and x.shape[-1] == channels and rgb.shape[-1] == 3
                and tuple(x.shape[:3]) == tuple(rgb.shape[:3]))


def compute_edge_aware_smoothness_loss(depth, rgb, assume_unit_range=None):
    """depth [B,H,W,1], rgb [B,H,W,3]: mean |d depth| * exp(-mean_c |d rgb|) along x plus along y.
    `assume_unit_range`: True = rgb lies in [0, 1] (fused kernel), False = PyTorch formulation, None = decided from
    where rgb comes from (see `_fused_ok`)."""
    if _fused_ok(depth, rgb, 1, assume_unit_range):
        from .fused_losses import edge_aware_smoothness
        return edge_aware_smoothness(materialize(depth), materialize(rgb))
    depth, rgb = materialize(depth), materialize(rgb)
    gx, gy = _image_gradients(rgb)
    dx = torch.abs(depth[..., :, :-1, :] - depth[..., :, 1:, :]) * torch.exp(-gx)
    dy = torch.abs(depth[..., :-1, :, :] - depth[..., 1:, :, :]) * torch.exp(-gy)
    return dx.mean() + dy.mean()


def compute_bilateral_normal_smoothness_loss(normal, rgb, assume_unit_range=None):
    """normal [B,H,W,3], rgb [B,H,W,3]: mean sqrt(1 + (|d n| exp(-3 mean_c |d rgb|))^2) along x plus along y."""
    if _fused_ok(normal, rgb, 3, assume_unit_range):
        from .fused_losses import bilateral_normal_smoothness
        return bilateral_normal_smoothness(materialize(normal), materialize(rgb))
    normal, rgb = materialize(normal), materialize(rgb)
    gx, gy = _image_gradients(rgb)
    nx = torch.abs(normal[..., :, :-1, :] - normal[..., :, 1:, :]) * torch.exp(-3 * gx)
    ny = torch.abs(normal[..., :-1, :, :] - normal[..., 1:, :, :]) * torch.exp(-3 * gy)
    return torch.sqrt(1 + nx ** 2).mean() + torch.sqrt(1 + ny ** 2).mean()
