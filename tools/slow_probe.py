import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
from dimo_amd import _lib
import ctypes as C
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
for _ in range(1150): tr.train_step()
torch.cuda.synchronize()
L = _lib.lib()
L.dimo_timing_select(None); L.dimo_timing_enable(1)
for _ in range(10): tr.train_step()
torch.cuda.synchronize()
L.dimo_timing_enable(0)
t = bench.read_timing()
print({k: round(v[0] / max(v[1], 1), 4) for k, v in t.items()})
g = tr.renderer.gaussians
d = (g._xyz.detach() ** 2).sum(1).sqrt()
print("N", g._xyz.shape[0], "|x| max %.2f  99.9%% %.2f" % (float(d.max()), float(d.kthvalue(int(0.999 * len(d))).values)), "scale max %.2f" % float(g.get_scaling.max()))
import gc
def rate(n=60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.train_step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("slow phase: %.3f ms/step" % rate(), "allocated %.2f GB reserved %.2f GB" % (torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
tr._exec = None; gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
print("after dropping the executor + empty_cache: %.3f ms/step" % rate(), "reserved %.2f GB" % (torch.cuda.memory_reserved() / 1e9))
