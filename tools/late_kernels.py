"""Per-kernel-group times (library HIP-event timers, ms per launch) of the bench workload at several points of a long
run: python tools/late_kernels.py [steps ...]   (default 30 700 1400 2100)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import bench
from dimo_amd import _lib

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
L = _lib.lib()
marks = [int(a) for a in sys.argv[1:]] or [30, 700, 1400, 2100]
done = 0
import time
for m in marks:
    while done < m:
        tr.train_step()
        done += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.train_step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 20
    done += 20
    L.dimo_timing_select(None)
    L.dimo_timing_enable(1)
    for _ in range(4):
        tr.train_step()
    torch.cuda.synchronize()
    L.dimo_timing_enable(0)
    done += 4
    t = bench.read_timing()
    pol.poll(lag=0)
    print("after %5d steps: %.3f ms/step, N %d, R mean %.0f | " % (done, ms, tr.renderer.gaussians._xyz.shape[0], pol.last_r_mean or 0)
          + " ".join("%s %.0f" % (k, 1e3 * v[0] / v[1]) for k, v in t.items() if v[1]))
