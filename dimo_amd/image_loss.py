"""Raw (non-autograd) front end of dimo_image_loss (dimo_amd/csrc/image_loss.hip): all per-image training
losses of one motion's batch and the gradient images the rasterizer backward consumes, in one kernel.
Loss definitions: main_train_dimo.py:331-372, src/loss.py:64-106.  GPU only."""
import ctypes as C

import torch

from . import _lib


def loss_weights(cfg, n_local, n_img, H, W, depth_on=None, normal_on=None):
    """Folds lambda, the 1/numel of each mean-type term and this rank's share n_local/n_img of the motion's
    images into per-element weights (see Trainer.motion_loss for the autograd formulation).  `depth_on` /
    `normal_on` override cfg.add_depth / cfg.add_normal (the terms start after depth/normal_reg_start_iter)."""
    share = n_local / n_img
    B = n_local
    depth_on = cfg.add_depth if depth_on is None else depth_on
    normal_on = cfg.add_normal if normal_on is None else normal_on
    return dict(
        w_mask=cfg.lambda_mask * share / (B * H * W),
        w_smooth_x=(cfg.lambda_smooth * share / (B * H * (W - 1))) if depth_on and W > 1 else 0.0,
        w_smooth_y=(cfg.lambda_smooth * share / (B * (H - 1) * W)) if depth_on and H > 1 else 0.0,
        w_bilat_x=(cfg.lambda_bilateral * share / (3 * B * H * (W - 1))) if normal_on and W > 1 else 0.0,
        w_bilat_y=(cfg.lambda_bilateral * share / (3 * B * (H - 1) * W)) if normal_on and H > 1 else 0.0,
    )


LOSS_WORDS = 512  # include/dimo_hip.h: DIMO_LOSS_WORDS


def fused_image_loss(image, depth, normal, alpha, gt, mask, w_mse, weights, ssim_grad, loss_accum, out=None,
                     stream=None, g_dot=None):
    """image[B,3,H,W] (raw, unclamped) depth[B,1,H,W]|None normal[B,3,H,W]|None alpha[B,1,H,W] gt[B,3,H,W]
    mask [1,H,W] (shared) or [B,1,H,W] -- or `gt` / `mask` as LISTS of B separate [3,H,W] / [1,H,W] tensors (no
    stacking copy); w_mse: python list of B floats (already divided by 3HW).
    Adds the loss to `loss_accum` (LOSS_WORDS floats; the loss is their sum) and returns (g_image, g_depth|None, g_normal|None, g_alpha).
    `stream`: raw stream handle to launch on (default: torch's current stream).  `g_dot` (optional [B,1,H,W]):
    receives sum_channels(gradient x rendered value) per pixel for the rasterizer backward."""
    if not image.is_cuda:
        raise RuntimeError("dimo_amd.image_loss needs GPU tensors (no CPU fallback in the product path)")
    B, _, H, W = image.shape
    if loss_accum.numel() < LOSS_WORDS:
        raise ValueError(f"loss_accum needs {LOSS_WORDS} floats (dimo_hip.h: DIMO_LOSS_WORDS)")
    new = lambda ref: torch.empty_like(ref)
    if out is None:
        g_image, g_alpha = new(image), new(alpha)
        g_depth = new(depth) if depth is not None else None
        g_normal = new(normal) if normal is not None else None
    else:
        g_image, g_depth, g_normal, g_alpha = out
    gt_list = mask_list = None
    if isinstance(gt, (list, tuple)):
        gt_list, gt = _lib.ptr_array(gt), None
    if isinstance(mask, (list, tuple)):
        mask_list, mask = _lib.ptr_array(mask), None
    per_image = 1 if (mask is not None and mask.dim() == 4 and mask.shape[0] == B and B > 1) else 0
    w_arr = (C.c_float * B)(*w_mse)
    _lib.check(_lib.lib().dimo_image_loss(
        B, H, W, _lib.ptr(image), _lib.ptr(depth), _lib.ptr(normal), _lib.ptr(alpha), _lib.ptr(gt), _lib.ptr(mask),
        per_image, w_arr, weights["w_mask"], weights["w_smooth_x"], weights["w_smooth_y"], weights["w_bilat_x"],
        weights["w_bilat_y"], _lib.ptr(ssim_grad), _lib.ptr(loss_accum), _lib.ptr(g_image), _lib.ptr(g_depth),
        _lib.ptr(g_normal), _lib.ptr(g_alpha), _lib.ptr(g_dot), gt_list, mask_list,
        stream if stream is not None else _lib.current_stream()),
        "dimo_image_loss")
    return g_image, g_depth, g_normal, g_alpha


def fused_ssim_image_loss(image, depth, normal, alpha, gt, mask, w_mse, weights, ssim_coef, ssim_sum, loss_accum,
                          out=None, stream=None, g_dot=None):
    """`fused_image_loss` with the SSIM term inside (dimo_ssim_image_loss: ONE tile pass, no SSIM gradient image in
    between).  `ssim_coef`: 1-element device tensor, dL/d(mean SSIM of the batch) = -lambda_ssim x share;
    `ssim_sum`: 1-element device tensor (zeroed by the caller) that receives the sum of the SSIM map.  Everything
    else as `fused_image_loss`."""
    if not image.is_cuda:
        raise RuntimeError("dimo_amd.image_loss needs GPU tensors (no CPU fallback in the product path)")
    B, _, H, W = image.shape
    if loss_accum.numel() < LOSS_WORDS:
        raise ValueError(f"loss_accum needs {LOSS_WORDS} floats (dimo_hip.h: DIMO_LOSS_WORDS)")
    new = lambda ref: torch.empty_like(ref)
    if out is None:
        g_image, g_alpha = new(image), new(alpha)
        g_depth = new(depth) if depth is not None else None
        g_normal = new(normal) if normal is not None else None
    else:
        g_image, g_depth, g_normal, g_alpha = out
    gt_list = mask_list = None
    if isinstance(gt, (list, tuple)):
        gt_list, gt = _lib.ptr_array(gt), None
    if isinstance(mask, (list, tuple)):
        mask_list, mask = _lib.ptr_array(mask), None
    per_image = 1 if (mask is not None and mask.dim() == 4 and mask.shape[0] == B and B > 1) else 0
    w_arr = (C.c_float * B)(*w_mse)
    _lib.check(_lib.lib().dimo_ssim_image_loss(
        B, H, W, _lib.ptr(image), _lib.ptr(depth), _lib.ptr(normal), _lib.ptr(alpha), _lib.ptr(gt), _lib.ptr(mask),
        per_image, w_arr, weights["w_mask"], weights["w_smooth_x"], weights["w_smooth_y"], weights["w_bilat_x"],
        weights["w_bilat_y"], _lib.ptr(ssim_coef), _lib.ptr(ssim_sum), _lib.ptr(loss_accum), _lib.ptr(g_image),
        _lib.ptr(g_depth), _lib.ptr(g_normal), _lib.ptr(g_alpha), _lib.ptr(g_dot), gt_list, mask_list,
        stream if stream is not None else _lib.current_stream()),
        "dimo_ssim_image_loss")
    return g_image, g_depth, g_normal, g_alpha
