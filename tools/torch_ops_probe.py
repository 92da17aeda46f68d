"""Which torch (non-library) kernels a training step launches: python tools/torch_ops_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
for _ in range(5): tr.train_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3): tr.train_step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=40, max_name_column_width=60, max_src_column_width=90))
