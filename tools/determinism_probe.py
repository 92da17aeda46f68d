"""Which parts of a step's gradient bucket are bit-reproducible run to run?  The same step (same parameters, same
triples) taken several times without an optimizer update; the flat gradient bucket compared group by group.

    python tools/determinism_probe.py [--num-pts 100000 --res 512 --repeats 4]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-pts", type=int, default=100000)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--repeats", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tr, pol = bench.make_trainer(dev, 0, 1, a.num_pts, a.res)
    g = tr.renderer.gaussians
    for _ in range(3):
        tr.train_step()
    torch.cuda.synchronize()
    triples = tr.sample()
    tr.optimizer.step = lambda *x, **k: None
    grads, losses = [], []
    for _ in range(a.repeats):
        g.zero_grad()
        tr.step = 1000
        tr.train_step(triples)
        torch.cuda.synchronize()
        grads.append(g.flat_grads.detach().clone())
        losses.append(float(tr.last_loss))
    out = {"loss_values": losses, "groups": {}}
    base = g.flat_params.data_ptr()
    for grp in tr.optimizer.param_groups:
        lo = min((p.data_ptr() - base) // 4 for p in grp["params"])
        hi = max((p.data_ptr() - base) // 4 + p.numel() for p in grp["params"])
        ref = grads[0][lo:hi]
        diffs = [float((x[lo:hi] - ref).abs().max()) for x in grads[1:]]
        out["groups"][grp["name"]] = {"identical": all(d == 0.0 for d in diffs), "max_abs_diff": max(diffs),
                                      "max_abs": float(ref.abs().max())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
