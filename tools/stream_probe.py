"""Does running independent renders on several HIP streams fill the GPU better?  8 raster fwd+bwd on 1/2/4/8 streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dimo_amd import rasterizer as rz
from dimo_amd.deform import fused_skinning
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
tr.find_knn()
g = tr.renderer.gaussians
cams = [tr.cams.get(0, tr.azimuths[i % 9], 2, 512, 512) for i in range(8)]
with torch.no_grad():
    ins = []
    for i in range(8):
        dx, dq = g._timenet(g._c_xyz, tr.source_time[i], g.latent_code(i))
        ins.append([t.detach() for t in fused_skinning(g._xyz, g._rotation, g._scaling, g._opacity, g._c_xyz, g._c_radius, dx, dq, g.neighbor_dists, g.neighbor_indices)])
gr = [torch.randn(3, 512, 512, device=dev), torch.randn(1, 512, 512, device=dev), torch.randn(3, 512, 512, device=dev), torch.randn(1, 512, 512, device=dev)]
cap = rz.CapacityPolicy(initial=1 << 21)
fdc = g._features_dc.detach()
def run(S):
    streams = [torch.cuda.Stream() for _ in range(S)]
    main = torch.cuda.current_stream()
    def once():
        for s in streams: s.wait_stream(main)
        for i in range(8):
            with torch.cuda.stream(streams[i % S]):
                pts, rot, sc, op = ins[i]
                *_, st = rz.raster_forward(pts, fdc, None, op, sc, rot, None, tr.renderer._settings(cams[i], 1.0, None), True, cap)
                rz.raster_backward(st, *gr)
        for s in streams: main.wait_stream(s)
    for _ in range(3): once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): once()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    cap.check()
    print(f"streams={S}: {dt*1e3:.2f} ms per 8 renders fwd+bwd  ({dt/8*1e3:.3f} ms/render)")
for S in (1, 2, 4, 8): run(S)
