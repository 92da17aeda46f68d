"""Step time early and deep into a run (after the first stage-s2 prune), for the executor mode in the environment."""
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
for _ in range(50): tr.train_step()
a = rate()
for _ in range(1100): tr.train_step()
b = rate()
print("streams=%s joint=%s: early %.3f ms/step, late %.3f ms/step, capacity %d, N %d" % (
    os.environ.get("DIMO_EXEC_STREAMS", "-2"), os.environ.get("DIMO_JOINT_BWD", "1"), a, b, pol.capacity,
    tr.renderer.gaussians._xyz.shape[0]))
