"""Stage s1 on the HIP pipeline alone (bench.py's `s1_frames_per_s` workload), more steps:
    python tools/s1_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from dimo_amd.rasterizer import CapacityPolicy
from dimo_amd.renderer import Renderer
from dimo_amd.synth import init_synthetic_model
from dimo_amd.trainer import TrainConfig, Trainer
dev = torch.device("cuda", 0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
c1 = TrainConfig(num_pts=512, resolution=512, motions_per_step=2, views_per_step=2, frames_per_step=2, stage="s1",
                 FPS_iter=10 ** 9, position_lr_max_steps=500)
rd1 = Renderer(sh_degree=0, white_background=True, radius=c1.radius, num_latent_code=c1.num_motions, add_normal=True,
               device=dev, capacity=CapacityPolicy(initial=1 << 22))
init_synthetic_model(rd1, c1.num_pts, c1.num_cpts, seed=0, regime="trained", num_latent=c1.num_motions)
rd1.gaussians._r = torch.nn.Parameter(torch.full((1, 1), -3.2, device=dev))
t1 = Trainer(c1, rd1)
t1.step = 1100
for _ in range(10):
    t1.train_step()
torch.cuda.synchronize()
for rep in range(3):
    ts = time.perf_counter()
    n = sum(t1.train_step() for _ in range(k))
    torch.cuda.synchronize()
    dt = time.perf_counter() - ts
    print("s1: %.0f frames/s, %.3f ms per step (%d steps)" % (n / dt, 1e3 * dt / k, k))
