"""Walks the reference's two-stage schedule end to end on synthetic data (stage s1 with FPS and densification, then
stage s2 through the 128 / 256 / 512 render sizes and a prune) and prints the step time per phase:
    python tools/schedule_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from dimo_amd.rasterizer import CapacityPolicy
from dimo_amd.renderer import Renderer
from dimo_amd.synth import init_synthetic_model
from dimo_amd.trainer import TrainConfig, Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
def run(tr, n, label):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = 0
    for _ in range(n): r += tr.train_step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    g = tr.renderer.gaussians
    print("%-34s step %5d  N %6d  %.3f ms/step  %.0f frames/s  loss %.1f  skipped %d" % (
        label, tr.step, g._xyz.shape[0], dt / n * 1e3, r / dt, tr.last_loss.item(), tr.skipped_steps))
# ---- stage s1: the TimeNet moves the Gaussians; FPS to num_cpts at step 0, densification every 100 steps
c1 = TrainConfig(num_pts=20000, resolution=512, stage="s1", motions_per_step=2, views_per_step=2, frames_per_step=2)
rd = Renderer(sh_degree=0, white_background=True, radius=c1.radius, num_latent_code=c1.num_motions, add_normal=True,
              device=dev, capacity=CapacityPolicy(initial=1 << 22))
init_synthetic_model(rd, c1.num_pts, c1.num_cpts, seed=0, regime="trained", num_latent=c1.num_motions)
g = rd.gaussians
g._r = torch.nn.Parameter(torch.full((1, 1), -3.2, device=dev))
tr = Trainer(c1, rd)
for k in range(6):
    run(tr, 100, "s1 (FPS at 0, densify every 100)")
# ---- stage s2 from step 0: 100 k Gaussians, progressive render size, prune at step 1000
c2 = TrainConfig(num_pts=100000, resolution=512, motions_per_step=2, views_per_step=2, frames_per_step=2)
rd2 = Renderer(sh_degree=0, white_background=True, radius=c2.radius, num_latent_code=c2.num_motions, add_normal=True,
               device=dev, capacity=CapacityPolicy(initial=4000000))
init_synthetic_model(rd2, c2.num_pts, c2.num_cpts, seed=0, regime="trained", num_latent=c2.num_motions)
tr2 = Trainer(c2, rd2)
for n, label in ((300, "s2 128^2 (steps 1-300)"), (150, "s2 256^2 (301-450)"), (300, "s2 512^2 (451-750)"),
                 (300, "s2 512^2 (751-1050, prune at 1000)"), (300, "s2 512^2 (1051-1350)")):
    run(tr2, n, label)
