import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dimo_amd import _lib
L = _lib.lib()
for N in (100000, 1000000):
    pts = (torch.randn(N, 3, device="cuda") * 0.3).contiguous()
    out = torch.empty(N, device="cuda")
    ws = torch.empty(L.dimo_dist2_workspace_bytes(N), dtype=torch.uint8, device="cuda")
    for name, fn in (("grid", lambda: L.dimo_dist2_grid(N, pts.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream())),
                     ("brute", lambda: L.dimo_dist2(N, pts.data_ptr(), out.data_ptr(), _lib.current_stream()))):
        if name == "brute" and N > 200000:
            continue
        fn(); torch.cuda.synchronize()
        t = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(N, name, "%.3f ms" % (1e3 * (time.perf_counter() - t)))
