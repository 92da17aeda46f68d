"""Isolated image_loss kernel time (4 x 512^2 batch): python tools/loss_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from types import SimpleNamespace
from dimo_amd.image_loss import fused_image_loss, loss_weights
B, H, W = 4, 512, 512
d = "cuda"
img, dep, nrm, al = torch.rand(B, 3, H, W, device=d), torch.rand(B, 1, H, W, device=d), torch.rand(B, 3, H, W, device=d), torch.rand(B, 1, H, W, device=d)
gt, mask, sg = torch.rand(B, 3, H, W, device=d), torch.rand(1, H, W, device=d), torch.rand(B, 3, H, W, device=d)
cfg = SimpleNamespace(lambda_mask=1.0, lambda_smooth=0.1, lambda_bilateral=0.1, add_depth=True, add_normal=True)
wts = loss_weights(cfg, B, B, H, W)
acc = torch.zeros(512, device=d)
out = fused_image_loss(img, dep, nrm, al, gt, mask, [1e-6] * B, wts, sg, acc)
f = lambda: fused_image_loss(img, dep, nrm, al, gt, mask, [1e-6] * B, wts, sg, acc, out=out)
for _ in range(10): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): f()
e1.record(); torch.cuda.synchronize()
print(f"image_loss: {e0.elapsed_time(e1) * 10:.1f} us per call")
