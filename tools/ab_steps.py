"""Wall time per step in blocks (variance check): python tools/ab_steps.py [blocks] [steps_per_block]"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
for _ in range(5): tr.train_step()
torch.cuda.synchronize()
sel = os.environ.get("AB_TIMING")
if sel is not None:
    from dimo_amd import _lib
    L = _lib.lib()
    L.dimo_timing_select(sel.encode() if sel else None)
    L.dimo_timing_enable(1)
for b in range(blocks):
    t0 = time.perf_counter()
    for _ in range(steps): tr.train_step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"block {b}: {1e3*(t2-t0)/steps:.3f} ms/step  (enqueue {1e3*(t1-t0)/steps:.3f} ms/step, drain {1e3*(t2-t1):.2f} ms)  skipped={tr.skipped_steps}")
