"""Which lines of the step call into torch (host overhead / small kernels): python tools/torch_calls_probe.py"""
import collections, os, sys, traceback
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
for _ in range(5): tr.train_step()
torch.cuda.synchronize()
counts = collections.Counter()
def wrap(owner, name):
    orig = getattr(owner, name)
    def f(*a, **k):
        fr = [x for x in traceback.extract_stack(limit=6)[:-1] if 'torch_calls_probe' not in x.filename]
        loc = " < ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-3:][::-1])
        counts[(name, loc)] += 1
        return orig(*a, **k)
    setattr(owner, name, f)
for n in ("tensor", "stack", "zeros", "zeros_like", "empty", "empty_like", "cat", "as_tensor", "ones", "full"):
    wrap(torch, n)
for n in ("to", "copy_", "fill_", "zero_", "max", "item", "clone", "contiguous", "cpu", "sum", "__add__", "__iadd__", "__mul__", "add_", "mul_", "detach", "view", "__getitem__", "float"):
    wrap(torch.Tensor, n)
steps = 3
for _ in range(steps): tr.train_step()
torch.cuda.synchronize()
for (name, loc), c in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f"{c / steps:6.1f}/step  {name:12s} {loc}")
