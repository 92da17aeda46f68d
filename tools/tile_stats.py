"""Per-tile workload statistics of the bench scene (list lengths, depth reached by early termination)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from dimo_amd import rasterizer as rz
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
tr, pol = bench.make_trainer(dev, 0, 1, 100000, 512)
tr.find_knn()
cam = tr.cams.get(0, tr.azimuths[0], 2, 512, 512)
tr.renderer.capacity = None
with torch.no_grad():
    out = tr.renderer.render(cam, time=tr.source_time[3], stage="s2", latent_index=0)
g = tr.renderer.gaussians
from dimo_amd.deform import fused_skinning
dx, dq = g._timenet(g._c_xyz, tr.source_time[3], g.latent_code(0))
pts, rot, sc, op = fused_skinning(g._xyz, g._rotation, g._scaling, g._opacity, g._c_xyz, g._c_radius, dx, dq, g.neighbor_dists, g.neighbor_indices)
s = tr.renderer._settings(cam, 1.0, None)
*_, st = rz.raster_forward(pts.detach(), g._features_dc.detach(), None, op.detach(), sc.detach(), rot.detach(), None, s, True, None)
ins = rz.inspect_state((st.geom, st.bin_ws, st.img_ws), st.N, 512, 512, st.r_cap)
ranges = ins["ranges"].cpu().numpy().astype(np.int64); n = ranges[:,1]-ranges[:,0]
nc = ins["n_contrib"].cpu().numpy().astype(np.int64).reshape(32,16,32,16).transpose(0,2,1,3).reshape(1024,256)
mx = nc.max(1)
print("R", n.sum(), "tiles nonempty", (n>0).sum(), "list len mean/50/90/99/max", n[n>0].mean(), np.percentile(n[n>0],[50,90,99]), n.max())
print("max n_contrib per tile mean/50/90/99/max", mx[n>0].mean(), np.percentile(mx[n>0],[50,90,99]), mx.max(), " sum(max_last)/R", mx.sum()/n.sum())
print("mean n_contrib per pixel (nonempty tiles)", nc[n>0].mean(), "batches per tile mean", np.ceil(mx[n>0]/256).mean())
radii = st.radii.cpu().numpy(); print("radius mean/50/90/max", radii[radii>0].mean(), np.percentile(radii[radii>0],[50,90]), radii.max())
tt = ins["tiles_touched"].cpu().numpy(); print("tiles touched mean", tt[tt>0].mean())
