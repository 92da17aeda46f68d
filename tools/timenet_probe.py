"""Runs the fused TimeNet forward/backward alone (benchmark batch: 4 pairs x 512 control points) for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from dimo_amd.deform import TimeNet
from dimo_amd.fused_timenet import FusedTimeNet
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = TimeNet().to(dev)
f = FusedTimeNet(net)
P, M = 4, 512
c = torch.rand(M, 3, device=dev) - 0.5
tab = torch.randn(8, 32, device=dev)
gx, gr = torch.randn(P, M, 3, device=dev), torch.randn(P, M, 4, device=dev)
gc, gt = torch.zeros(M, 3, device=dev), torch.zeros_like(tab)
import ctypes as C
from dimo_amd import _lib
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for i in range(n + 3):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L.dimo_timing_select(None); L.dimo_timing_enable(1)
    f.forward(c, [0.1, 0.2, 0.3, 0.4], tab, [0, 1, 2, 3])
    f.backward(gx, gr, gc, gt)
torch.cuda.synchronize()
L.dimo_timing_enable(0)
for name in (b"timenet_fwd", b"timenet_bwd"):
    ms, k = C.c_double(0), C.c_int64(0)
    L.dimo_timing_read(name, C.byref(ms), C.byref(k))
    print(name.decode(), "%.1f us per call (%d calls)" % (1e3 * ms.value / max(k.value, 1), k.value))
print(f"fwd+bwd {1e3 * (time.perf_counter() - t0) / n:.3f} ms per iteration")
