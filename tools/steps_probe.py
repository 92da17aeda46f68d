import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
evs = []
t0 = time.perf_counter(); host = []
for i in range(60):
    tr.train_step()
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
d = [evs[i].elapsed_time(evs[i+1]) for i in range(len(evs)-1)]
print("device ms per step:", " ".join("%.2f" % x for x in d))
print("host   ms per step:", " ".join("%.2f" % ((host[i+1]-host[i])*1e3) for i in range(len(host)-1)))
