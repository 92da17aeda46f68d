"""Clocks and power deep into a run (after the first prune of stage s2), where a step takes twice as long:
    python tools/slow_probe.py"""
import os, sys, time, subprocess, threading
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        return " | ".join(l.split(":", 1)[-1].strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Power (W)")))
    except Exception as e:
        return repr(e)
def phase(label, n):
    stop = []
    samples = []
    def poll():
        while not stop:
            samples.append(smi()); time.sleep(0.3)
    th = threading.Thread(target=poll); th.start()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.train_step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    stop.append(1); th.join()
    print(label, "%.3f ms/step" % dt)
    for s_ in samples[1:4]: print("    ", s_)
for _ in range(50): tr.train_step()
phase("steps 50-850", 800)
for _ in range(300): tr.train_step()
phase("steps 1150-1950", 800)
