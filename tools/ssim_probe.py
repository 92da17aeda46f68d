"""Isolated SSIM kernel times: python tools/ssim_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from dimo_amd import _lib
L, st = _lib.lib(), _lib.current_stream()
B, C, H, W = 4, 3, 512, 512
a, b = torch.rand(B, C, H, W, device="cuda"), torch.rand(B, C, H, W, device="cuda")
coef = torch.tensor([-0.1], device="cuda")
s, part, g = torch.empty(1, device="cuda"), torch.empty(3, B, C, H, W, device="cuda"), torch.empty(B, C, H, W, device="cuda")
def two():
    L.dimo_ssim_forward(B, C, H, W, 1, _lib.ptr(a), _lib.ptr(b), _lib.ptr(s), _lib.ptr(part), st)
    L.dimo_ssim_backward(B, C, H, W, 1, _lib.ptr(a), _lib.ptr(b), _lib.ptr(part), _lib.ptr(coef), _lib.ptr(g), st)
def one():
    L.dimo_ssim_forward_backward(B, C, H, W, 1, _lib.ptr(a), _lib.ptr(b), _lib.ptr(coef), _lib.ptr(s), _lib.ptr(g), st)
for name, f in (("fwd+bwd", two), ("fused", one), ("fwd+bwd", two), ("fused", one)):
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 10:.1f} us per call")
