import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
for _ in range(5): tr.train_step()
torch.cuda.synchronize()
ex = tr._exec
N = 100000
align = lambda x: (x + 255) // 256 * 256
n = N; o = 0
def adv(sz):
    global o
    start = o; o = align(o + sz); return start
splat = adv(n*64); rect = adv(n*8); tiles = adv(n*4); offsets = adv(n*4); flags = adv(n); total = adv(16)
nb = (n + 255)//256
bsums = adv(3*(nb+1)*4); nka = adv(n*8); nva = adv(n*4); nkb = adv(n*8); nvb = adv(n*4)
bk = o
for i in range(8):
    gbuf = ex.slots[i]["geom"]
    w = gbuf[bk:bk + 4*(4 + 260 + 256 + 256)].view(torch.int32).cpu().numpy().view(np.uint32)
    tot = gbuf[total:total+16].view(torch.int32).cpu().numpy().view(np.uint32)
    bs = gbuf[bsums:bsums + 3*(nb+1)*4].view(torch.int32).cpu().numpy().view(np.uint32)
    bmin, bmax = bs[nb+1:nb+1+nb], bs[2*(nb+1):2*(nb+1)+nb]
    base = w[4:4+129]; sizes = np.diff(base)
    print("render", i, "kmin %x shift %d nslice %d total %s" % (w[0], w[1], w[2], tot), "block min range %x..%x max range %x..%x" % (bmin.min(), bmin.max(), bmax.min(), bmax.max()), "bins: max", sizes.max(), "first", sizes[:3], "last nonzero", sizes[sizes>0][-3:], "used", (sizes>0).sum())
