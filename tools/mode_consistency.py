"""Runs N training steps of the bench workload and saves the flat parameters: python tools/mode_consistency.py out.pt [steps]
(compare runs under different DIMO_EXEC_STREAMS / DIMO_INORDER_LOSSES settings: schedules of the same kernels)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for _ in range(n): tr.train_step()
torch.cuda.synchronize()
p = tr.renderer.gaussians.flat_params.detach().cpu()
print("skipped", tr.skipped_steps, "finite", bool(torch.isfinite(p).all()), "loss", float(tr.last_loss))
torch.save(p, sys.argv[1])
