"""Is a long run's slowdown the training state or the GPU's clocks?  Frozen parameters (Adam no-op, no pruning), 4000
steps, rate per 250 steps, plus rocm-smi clocks if available."""
import os, sys, time, subprocess
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
tr.cfg.density_end_iter_s2 = 0
frozen = os.environ.get("FROZEN", "1") == "1"
if frozen:
    import types
    g = tr.renderer.gaussians
    orig = g.update_learning_rate
    def no_lr(self, *a, **k):
        orig(*a, **k)
        for grp in self.optimizer.param_groups: grp["lr"] = 0.0
    g.update_learning_rate = types.MethodType(no_lr, g)
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "Temperature (Sensor junction)"))]
        return " | ".join(keep)[:300]
    except Exception as e:
        return repr(e)
for k in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(250): tr.train_step()
    torch.cuda.synchronize()
    print("steps %4d-%4d: %.3f ms/step" % (250 * k, 250 * k + 250, (time.perf_counter() - t0) / 250 * 1e3), smi() if k % 4 == 3 else "")
