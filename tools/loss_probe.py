"""The image-loss kernels alone, per launch of B x 3 x 512^2 (one motion's batch):
  two kernels : SSIM value + gradient (ssim_fused) then the fused image losses (image_loss)
  one pass    : dimo_ssim_image_loss (ssim_loss_tile_kernel)
    python tools/loss_probe.py [B ...]        """
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from dimo_amd import _lib
from dimo_amd.image_loss import fused_image_loss, fused_ssim_image_loss, LOSS_WORDS
L = _lib.lib()
st = _lib.current_stream()
H = W = 512


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


for B in [int(a) for a in sys.argv[1:]] or [4, 8]:
    g = torch.Generator().manual_seed(B)
    img = (torch.rand(B, 3, H, W, generator=g) * 1.2 - 0.1).cuda()
    gt = torch.rand(B, 3, H, W, generator=g).cuda()
    dep, alp = torch.rand(B, 1, H, W, generator=g).cuda(), torch.rand(B, 1, H, W, generator=g).cuda()
    nrm, mask = torch.randn(B, 3, H, W, generator=g).cuda(), (torch.rand(B, 1, H, W, generator=g) > 0.5).float().cuda()
    coef, ssum, gs = torch.tensor([-0.2], device="cuda"), torch.zeros(1, device="cuda"), torch.empty_like(img)
    acc, gdot = torch.zeros(LOSS_WORDS, device="cuda"), torch.empty_like(alp)
    wts = dict(w_mask=1e-6, w_smooth_x=1e-6, w_smooth_y=1e-6, w_bilat_x=1e-6, w_bilat_y=1e-6)
    out = (torch.empty_like(img), torch.empty_like(dep), torch.empty_like(nrm), torch.empty_like(alp))
    w_mse = [1e-6] * B

    def ssim():
        _lib.check(L.dimo_ssim_forward_backward(B, 3, H, W, 1 | 2, _lib.ptr(img), _lib.ptr(gt), _lib.ptr(coef),
                                                _lib.ptr(ssum), _lib.ptr(gs), st), "ssim")

    def losses():
        fused_image_loss(img, dep, nrm, alp, gt, mask, w_mse, wts, gs, acc, out=out, g_dot=gdot)

    def one_pass():
        fused_ssim_image_loss(img, dep, nrm, alp, gt, mask, w_mse, wts, coef, ssum, acc, out=out, g_dot=gdot)

    t_s, t_l, t_o = timed(ssim), timed(losses), timed(one_pass)
    print("B = %d: SSIM value + gradient %.1f us + image losses %.1f us = %.1f us | one pass %.1f us"
          % (B, t_s, t_l, t_s + t_l, t_o))
