"""Per-workgroup phase trace of the binning kernels (library built with DIMO_BIN_TRACE=1; GPU box):
    DIMO_EXEC_STREAMS=0 python tools/bin_trace.py
One training step of the bench workload is traced; for every kernel: its span (first start -> last end), and per
mark interval the mean / p90 / max over the workgroups, in microseconds (s_memrealtime: 100 MHz)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

import bench
from dimo_amd import _lib

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
for _ in range(30):
    tr.train_step()
torch.cuda.synchronize()
L = _lib.lib()
cap = 1 << 16
buf = torch.zeros(cap * 32, dtype=torch.int64, device=dev)
assert L.dimo_debug_bin_trace(buf.data_ptr(), cap) >= 0
tr.train_step()
torch.cuda.synchronize()
n = L.dimo_debug_bin_trace(None, 0)
rec = buf.cpu().numpy().reshape(cap, 32)[:min(n, cap)]
# bucket_sort's marks: start, first item's words in, then one per item of the work list the workgroup took (its two low
# bits say what the item was), end
names = {1: "level1", 2: "level1_scatter (until round 4)", 3: "bucket_sort", 4: "level2_count (until round 4)", 5: "level2_fill"}
print("records", n)
for kid, name in names.items():
    r = rec[(rec[:, 0] >> 56) == kid]
    if not len(r):
        continue
    # launches: group by start time gaps (a step has one or two launches per kernel)
    t = r[:, 1:].astype(np.float64)
    t[t == 0] = np.nan
    t0 = np.nanmin(t[:, 0])
    t = (t - t0) / 100.0  # us
    starts = t[:, 0]
    order = np.argsort(starts)
    cuts = np.where(np.diff(starts[order]) > 50.0)[0]
    groups = np.split(order, cuts + 1)
    for gi, g in enumerate(groups):
        tg = t[g]
        end = np.nanmax(tg, axis=1)
        print(f"{name} launch {gi}: {len(g)} workgroups, span {np.nanmax(end) - np.nanmin(tg[:, 0]):.1f} us; "
              f"workgroup life mean {np.mean(end - tg[:, 0]):.1f} p90 {np.percentile(end - tg[:, 0], 90):.1f} max {np.max(end - tg[:, 0]):.1f}; "
              f"start spread {np.max(tg[:, 0]) - np.min(tg[:, 0]):.1f}")
        nm = int(np.max(np.sum(~np.isnan(tg), axis=1)))
        for k in range(1, min(nm, 14)):
            d = tg[:, k] - tg[:, k - 1]
            d = d[~np.isnan(d)]
            if len(d):
                print(f"     mark {k - 1}->{k}: n {len(d):5d} mean {d.mean():6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f}")
        if kid == 3:  # bucket_sort: the items by kind (two low bits of the mark that ends them)
            kinds = {1: "bucket sorted in LDS", 2: "slice", 3: "byte passes (or an idle slice of such a bucket)", 0: "skipped"}
            raw = r[g][:, 1:]
            for kd, kname in kinds.items():
                d = []
                for row in raw:
                    ts = row[row != 0]
                    for k in range(2, len(ts) - 1):  # (the last mark is the flush: no kind)
                        if (int(ts[k]) & 3) == kd:
                            d.append((int(ts[k]) - int(ts[k - 1])) / 100.0)
                if d:
                    d = np.array(d)
                    print(f"     items '{kname}': n {len(d)} mean {d.mean():.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f} sum {d.sum():.0f} us")
        # the slowest workgroup's marks
        w = int(np.argmax(end - tg[:, 0]))
        print("     slowest workgroup", int(r[g[w], 0] & 0xffffffff), "render", int((r[g[w], 0] >> 48) & 0xff), "marks",
              " ".join("%.1f" % x for x in (tg[w] - tg[w, 0]) if not np.isnan(x)))
