"""Host-side profile of the reference-shaped (drop-in surface) step: python tools/dropin_host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tr, _ = bench.make_trainer(dev, 0, 1, 100000, 512, capacity=False, direct=False)
for _ in range(5):
    tr.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.train_step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("20 steps: host enqueue %.2f ms/step, with device drain %.2f ms/step" % (50 * t_host, 50 * t_all))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.train_step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
