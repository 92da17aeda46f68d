"""Does the step slow down as training moves the Gaussians, and does re-sorting them (Morton order) bring it back?
    python tools/drift_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
def rate(n=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.train_step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(30): tr.train_step()
print("steps 30-130: %.3f ms" % rate())
g = tr.renderer.gaussians
x0 = g._xyz.detach().clone()
for k in range(4):
    for _ in range(400): tr.train_step()
    r_before = rate()
    tot = pol.last_r_mean if hasattr(pol, "last_r_mean") else None
    g = tr.renderer.gaussians
    print("after ~%d steps: %.3f ms/step, N %d, R mean %s, max scale %.3f, mean opacity %.3f" % (
        530 + 500 * k, r_before, g._xyz.shape[0], tot, float(g.get_scaling.max()), float(g.get_opacity.mean())))
g.sort_spatially()
print("after re-sort: %.3f ms" % rate())
