"""Where a reference-shaped (drop-in surface) step spends its time: python tools/dropin_probe.py [--capacity]
Phases of Trainer._forward_backward_autograd, each bracketed by a device sync, mean over 5 steps (ms)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tr, _ = bench.make_trainer(dev, 0, 1, 100000, 512, capacity="--capacity" in sys.argv, direct=False)
for _ in range(3):
    tr.train_step()
torch.cuda.synchronize()
acc = {}


def lap(name, t0):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0)
    return time.perf_counter()


K = 5
t_all = time.perf_counter()
for _ in range(K):
    t0 = time.perf_counter()
    g = tr.renderer.gaussians
    tr.step += 1
    g.update_learning_rate(tr.step, tr.stage)
    tr.find_knn(4)
    triples = tr.sample()
    t0 = lap("knn+sample", t0)
    deforms = tr.batched_deform(triples)
    t0 = lap("timenet_fwd(torch)", t0)
    outs = {}
    by_motion = {}
    for (m, v, f) in triples:
        out = tr.render_triple(m, v, f, deform=deforms.get((m, v, f)))
        gt, mask = tr.target(m, v, f)
        w = 1.0 if (v == 0 or f == 0) else 0.5
        rec = by_motion.setdefault(m, ([], [], [], []))
        rec[0].append(out), rec[1].append(gt), rec[2].append(mask), rec[3].append(w)
    t0 = lap("renders_fwd", t0)
    loss = None
    n_img = max(1, len(triples) // max(1, len({t[0] for t in triples})))
    for m, (o, gts, masks, ws) in by_motion.items():
        lm = tr.motion_loss(o, gts, masks, ws, n_img)
        loss = lm if loss is None else loss + lm
    t0 = lap("losses_fwd(torch)", t0)
    loss.backward()
    t0 = lap("backward", t0)
    cap = tr.renderer.capacity
    if cap is not None:
        cap.check()
    tr.optimizer.step()
    g.zero_grad()
    t0 = lap("adam", t0)
tot = time.perf_counter() - t_all
print("capacity policy:", "--capacity" in sys.argv, "| %.2f ms/step, %.0f frames/s" % (1e3 * tot / K, 8 * K / tot))
for k, v in acc.items():
    print("  %-22s %7.3f ms" % (k, 1e3 * v / K))
