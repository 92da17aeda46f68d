"""Timeline of the blend backward's work items (dimo_debug_blend_trace):
    DIMO_BWD_TRACE=1 python -m dimo_amd.csrc.build --force && python tools/bwd_trace.py
Prints, for one 4-render launch of the bench workload: item count, sum / max of the item durations, the launch's
span, achieved concurrency, and duration statistics by bucket index."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from dimo_amd import _lib

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
regime = sys.argv[sys.argv.index("--regime") + 1] if "--regime" in sys.argv else "trained"
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512, regime=regime)
print("regime:", regime)
for _ in range(4):
    tr.train_step()
torch.cuda.synchronize()
L = _lib.lib()
cap = 1 << 18
buf = torch.zeros(cap * 4, dtype=torch.int64, device=dev)
import ctypes as C


def timed_step(label):
    L.dimo_timing_select(b"blend_bwd,blend_fwd")
    L.dimo_timing_enable(1)
    tr.train_step()
    torch.cuda.synchronize()
    for name in (b"blend_bwd", b"blend_fwd"):
        ms, cnt = C.c_double(0), C.c_int64(0)
        L.dimo_timing_read(name, C.byref(ms), C.byref(cnt))
        print(label, name.decode(), "device time %.3f ms over %d launches" % (ms.value, cnt.value))
    L.dimo_timing_enable(0)


timed_step("trace off:")
L.dimo_debug_blend_trace(_lib.ptr(buf), cap)
timed_step("trace on: ")
n = L.dimo_debug_blend_trace(None, 0)
t = buf.cpu().numpy().view(np.uint64).reshape(-1, 4)[:n]
start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
xcc = ((t[:, 2] >> np.uint64(56)) & np.uint64(0xf)).astype(np.int64)
render = ((t[:, 2] >> np.uint64(48)) & np.uint64(0xff)).astype(np.int64)
bucket = (t[:, 2] & np.uint64(0xfff)).astype(np.int64)
n_rec = (t[:, 3] >> np.uint64(48)).astype(np.int64)
n_quad = ((t[:, 3] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
hw = (t[:, 3] & np.uint64(0xffff0000)).astype(np.int64)
n_useful = (t[:, 3] & np.uint64(0xffff)).astype(np.int64)
print(f"quadrant visits {n_quad.sum()}, of which some pixel used the entry in {n_useful.sum()} "
      f"({n_useful.sum() / max(1, n_quad.sum()):.3f}); records {n_rec.sum()}")
d = end - start
print(f"{n} items, duration mean {d.mean():.0f} ticks, max {d.max()}")
# quadrant visits per record, item by item: a record's 13-value wave reduction runs ONCE per
# (record, tile) whatever the number of quadrants it was evaluated on
ratio = n_quad[n_rec > 0] / n_rec[n_rec > 0]
hist, edges = np.histogram(ratio, bins=[1.0, 1.25, 1.5, 2.0, 2.5, 3.0, 3.5, 4.01])
print("items by mean quadrant visits per record:",
      ", ".join(f"[{edges[i]:.2f}, {edges[i + 1]:.2f}): {hist[i] / max(1, len(ratio)):.3f}" for i in range(len(hist))),
      f"| records weighted mean {n_quad.sum() / max(1, n_rec.sum()):.2f}")
for b in range(0, 18):
    m = bucket == b
    if m.any():
        print(f"   bucket {b}: {m.sum():5d} items, duration mean {d[m].mean():8.0f} max {d[m].max():8d}, records/item "
              f"{n_rec[m].mean():5.1f}, quadrant visits/record {n_quad[m].sum() / max(1, n_rec[m].sum()):.2f}")
# timeline on the chip-wide 100 MHz clock: split the step's launches by gaps in the start times
order = np.argsort(start)
gaps = np.diff(start[order])
cuts = [0] + [i + 1 for i in np.nonzero(gaps > 3000)[0]] + [len(order)]
for li in range(len(cuts) - 1):
    part = order[cuts[li]:cuts[li + 1]]
    if len(part) < 100:
        continue
    s_, e_ = start[part], end[part]
    span = e_.max() - s_.min()
    ev = np.concatenate([np.stack([s_, np.ones_like(s_)], 1), np.stack([e_, -np.ones_like(e_)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    alive = np.cumsum(ev[:, 1])
    edges = np.linspace(s_.min(), e_.max(), 21)
    prof = [int(alive[min(len(alive) - 1, np.searchsorted(ev[:, 0], x))]) for x in edges[:-1]]
    print(f"launch {li}: {len(part)} items, span {span / 100:.1f} us, mean item {d[part].mean() / 100:.1f} us, max {d[part].max() / 100:.1f} us, "
          f"sum/span = {d[part].sum() / span:.0f} waves alive on average, peak {int(alive.max())}")
    print("   waves alive at 5% steps:", prof)
    xs = xcc[part]
    print("   items per XCD:", np.bincount(xs, minlength=8).tolist(), " last end per XCD (us after start):",
          [round(float(e_[xs == x].max() - s_.min()) / 100, 1) if (xs == x).any() else None for x in range(8)])
