"""Host-side profile of the reference's loop body through the drop-in surface (tests/reference_step.py):
python tools/literal_loop_profile.py [--log 0|1] [--res 512] [--num-pts 100000]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--log", type=int, default=0)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--num-pts", type=int, default=100000)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()

from dimo_amd.rasterizer import CapacityPolicy
from tests.reference_step import ReferenceLoop
from dimo_amd.renderer import Renderer
from dimo_amd.synth import SyntheticTargets, init_synthetic_model
from dimo_amd.trainer import TrainConfig

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg = TrainConfig(num_pts=a.num_pts, resolution=a.res, motions_per_step=2, views_per_step=2, frames_per_step=2)
cfg.progressive_resolution = a.res <= 512
rd = Renderer(sh_degree=0, white_background=True, radius=cfg.radius, num_latent_code=cfg.num_motions, add_normal=True,
              device=dev, capacity=CapacityPolicy(initial=max(1 << 20, 40 * a.num_pts)))
init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=0, regime="trained", num_latent=cfg.num_motions)
rd.gaussians.sort_spatially()
rd.gaussians.training_setup(cfg)
loop = ReferenceLoop(cfg, rd, SyntheticTargets(a.res, dev, seed=0), log_scalars=bool(a.log))
loop.step = 1000
for _ in range(5):
    loop.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loop.train_step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("%d steps: host enqueue %.2f ms/step, with device drain %.2f ms/step -> %.0f frames/s"
      % (a.steps, 1e3 * t_host / a.steps, 1e3 * t_all / a.steps, 8 * a.steps / t_all))
# kernel launches per step
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        loop.train_step()
    torch.cuda.synchronize()
ev = prof.key_averages()
kern = [e for e in ev if e.device_type.name != "CPU"]
print("device kernels per step: %.0f, device time per step %.3f ms" %
      (sum(e.count for e in kern) / 3, sum(e.device_time_total for e in kern) / 3e3))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=35, max_name_column_width=60))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    loop.train_step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
