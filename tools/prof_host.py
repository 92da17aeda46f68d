import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
for _ in range(3): tr.train_step()
torch.cuda.synchronize()
# (a) wall per step, (b) host time to enqueue (no sync inside except cap.check)
t0=time.perf_counter()
for _ in range(5): tr.train_step()
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('enqueue per step ms', (t1-t0)/5*1e3, 'drain ms', (t2-t1)*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.train_step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
