"""Main-stream phase durations of the direct train step (HIP events), averaged over steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
for _ in range(5): tr.train_step()
torch.cuda.synchronize()
acc, n = {}, 20
t0 = time.perf_counter()
for _ in range(n):
    tr.marks = []
    s = torch.cuda.Event(enable_timing=True); s.record()
    tr.train_step()
    marks = [("step_begin", s)] + tr.marks
    torch.cuda.synchronize()
    for (a, ea), (b, eb) in zip(marks[:-1], marks[1:]):
        acc[b] = acc.get(b, 0.0) + ea.elapsed_time(eb)
print("wall ms/step (with per-step sync):", (time.perf_counter() - t0) / n * 1e3)
for k, v in acc.items(): print(f"{k:28s} {v/n:7.3f} ms")
