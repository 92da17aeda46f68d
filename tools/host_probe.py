"""Is the bench step host-bound?  Host time to ENQUEUE a step against the device time it takes:
    python tools/host_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(30):
    tr.train_step()
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(k):
        tr.train_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.3f ms per step, device finished %.3f ms after the last enqueue; %.3f ms per step in all"
          % (1e3 * (t1 - t0) / k, 1e3 * (t2 - t1), 1e3 * (t2 - t0) / k))
# the same with the device idle at every step (pure host cost of a step)
ts = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("host cost of one step with an idle device: median %.3f ms, min %.3f" % (1e3 * ts[len(ts) // 2], 1e3 * ts[0]))
