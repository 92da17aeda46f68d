"""frames/s of the reference's own loop body through the drop-in surface (bench.py: dropin_figures), alone:
    python tools/literal_fps.py [select]      ("select": rows of a pending batch by `select` per image, as before round 5)"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import bench

if "select" in sys.argv[1:]:
    from dimo_amd import batched_render as br
    br.LazyTensor._row = lambda self, i: self.materialize()[i]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = bench.dropin_figures(dev, 100000, 512, (2, 2, 2), steps=20)
print(json.dumps({"rows": "select" if "select" in sys.argv[1:] else "unbind", "headline": out["dropin_frames_per_s"],
                  **out["dropin_detail"]}))
