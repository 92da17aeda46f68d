"""Isolated dimo_knn timing, seeded and unseeded, Morton-sorted and random queries: python tools/knn_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from dimo_amd.knn_cuda import knn_points
from dimo_amd.densify import morton_order
torch.manual_seed(0)
N, M = 100000, 512
q = (torch.rand(N, 3, device="cuda") - 0.5)
ref = (torch.rand(M, 3, device="cuda") - 0.5)
def t(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, qq in (("random order", q), ("morton order", q[morton_order(q)].contiguous())):
    _, idx = knn_points(ref, qq, 4)
    print(name, "unseeded %.1f us" % t(lambda: knn_points(ref, qq, 4)), "seeded %.1f us" % t(lambda: knn_points(ref, qq, 4, seed=idx)))
