"""Bucket statistics of the binning's level 1 on the bench workload (GPU box): python tools/bin_stats.py
Sizes of the (supertile, depth bin) buckets of every render slot after a few steps, slices, and the largest sub-bin
(entries that rank themselves quadratically) per bucket with the kernels' own sub-bin map."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import bench
from dimo_amd import _lib

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    tr.train_step()
torch.cuda.synchronize()
ex = tr._exec
L = _lib.lib()
o = (C.c_size_t * 9)()
assert L.dimo_debug_bin_layout(ex.N, ex.r_cap, ex.H, ex.W, o) == 0
bk_off, l1tmp_off, l1_off, nbuckets, lg, NS, TOT, START, NSLICE = [int(x) for x in o]
for si in range(0, 8, 3):
    s = ex.slots[si]
    bk = s["geom"][bk_off:bk_off + 4 * (START + 2048 + 2048)].view(torch.int32).cpu().numpy().astype("int64") & 0xFFFFFFFF
    lo, shift = int(bk[0]), int(bk[1])
    start = bk[START:START + nbuckets]
    # totals are cleared by level2_fill: sizes from consecutive starts (supertile lists are padded to 256)
    l1 = s["bin"][l1_off:l1_off + 16 * int(ex.r_cap)].view(torch.int32).view(-1, 4)
    order = start.argsort()
    print(f"slot {si}: lo={lo:#x} shift={shift} buckets={nbuckets} slices={int(bk[NSLICE])}")
    sizes = []
    worst = 0
    for j, b in enumerate(order[:-1]):
        n = int(start[order[j + 1]] - start[b])
        sizes.append(n)
    import numpy as np
    sizes = np.array(sizes)
    print("   bucket sizes (incl. padding at list ends): mean %.0f  p50 %d  p90 %d  p99 %d  max %d  >2048: %d" % (
        sizes.mean(), np.percentile(sizes, 50), np.percentile(sizes, 90), np.percentile(sizes, 99), sizes.max(), (sizes > 2048).sum()))
    # largest sub-bin per bucket with the kernel's map (sorted entries: keys ascending inside a bucket)
    big = 0
    fat = []
    for j, b in enumerate(order[:-1]):
        n = int(start[order[j + 1]] - start[b])
        if n < 64:
            continue
        keys = (l1[int(start[b]):int(start[b]) + n, 1].cpu().numpy().astype("int64")) & 0xFFFFFFFF
        dbin = int(b) & ((1 << lg) - 1)
        sh = min(shift, 31)
        klo = lo + (dbin << sh)
        off, bits = (128, 7) if dbin == 0 else (0, 8)
        ssh = sh - bits if sh > bits else 0
        f = np.where(keys < klo, 0, np.minimum(off + ((keys - klo) >> ssh), 255))
        m = np.bincount(f, minlength=256).max()
        fat.append(m)
    fat = np.array(fat)
    print("   largest sub-bin per bucket: mean %.0f  p90 %d  p99 %d  max %d  >512: %d" % (
        fat.mean(), np.percentile(fat, 90), np.percentile(fat, 99), fat.max(), (fat > 512).sum()))
