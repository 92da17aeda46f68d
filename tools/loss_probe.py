"""SSIM (one launch: value + gradient) and the fused image losses alone, per launch of B x 3 x 512^2:
    python tools/loss_probe.py [B ...]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import ctypes as C
import torch
from dimo_amd import _lib
from dimo_amd.image_loss import fused_image_loss, LOSS_WORDS
L = _lib.lib()
st = _lib.current_stream()
H = W = 512
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

    def once():
        _lib.check(L.dimo_ssim_forward_backward(B, 3, H, W, 1, _lib.ptr(img), _lib.ptr(gt), _lib.ptr(coef),
                                                _lib.ptr(ssum), _lib.ptr(gs), st), "ssim")
        fused_image_loss(img, dep, nrm, alp, gt, mask, [1e-6] * B, wts, gs, acc, out=out, g_dot=gdot)
    for _ in range(5): once()
    torch.cuda.synchronize()
    L.dimo_timing_select(None); L.dimo_timing_enable(1)
    for _ in range(30): once()
    torch.cuda.synchronize(); L.dimo_timing_enable(0)
    res = {}
    for name in (b"ssim_fwd", b"image_loss"):
        ms, n = C.c_double(0), C.c_int64(0)
        L.dimo_timing_read(name, C.byref(ms), C.byref(n))
        res[name.decode()] = 1e3 * ms.value / max(n.value, 1)
    print("B = %d: SSIM value + gradient %.1f us, image losses %.1f us per launch" % (B, res["ssim_fwd"], res["image_loss"]))
