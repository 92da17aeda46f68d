"""Workload for the rocprofv3 --pmc passes: a calibration copy of known size (1 GiB read + 1 GiB write with
16-byte lanes) followed by a few training steps of the bench workload.  See tools/pmc_summarise.py."""
import os
import sys

# rocprofv3 --pmc SERIALISES kernel dispatches: a stream that waits for a value another stream has yet to write
# (hipStreamWaitValue32, the executor's default cross-stream dependency) then never sees its producer scheduled --
# the counter passes run with event dependencies
os.environ["DIMO_XSTREAM"] = "event"

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if "--no-calibration" not in sys.argv:
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
    for _ in range(3):
        dst = src.clone()  # vectorized copy kernel: 1 GiB in, 1 GiB out
    torch.cuda.synchronize()
regime = sys.argv[sys.argv.index("--regime") + 1] if "--regime" in sys.argv else "trained"
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512, regime=regime)
if "--default-schedule" not in sys.argv:
    tr._joint_bwd = True  # the counters are read per launch of the roofline kernel: ONE launch over the step's 8 renders
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 4
for _ in range(steps):
    tr.train_step()
torch.cuda.synchronize()
