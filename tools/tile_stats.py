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

# ---- visit / activity statistics of the blend kernels (what the per-quadrant culling leaves to do) ----
sp = ins["splat"].float()            # [N,16]: x y A B C opacity r g b depth nx ny nz
vs = ins["vals_sorted"].long()
rg = ins["ranges"].long()
tot = dict(R=0, visits_fwd=0, actw_fwd=0, actl_fwd=0, visits_bwd=0, actw_bwd=0, actl_bwd=0)
ncd = ins["n_contrib"].long().reshape(32, 16, 32, 16).permute(0, 2, 1, 3).reshape(1024, 16, 16)
yy, xx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
quad = ((yy >= 8).long() * 2 + (xx >= 8).long()).reshape(-1)           # quadrant (= wave) of each pixel of a tile
for t in range(1024):
    lo, hi = int(rg[t, 0]), int(rg[t, 1])
    if hi == lo:
        continue
    e = sp[vs[lo:hi]]
    tx, ty = t % 32, t // 32
    px = (tx * 16 + xx).reshape(-1).float(); py = (ty * 16 + yy).reshape(-1).float()
    dx = e[:, 0:1] - px[None]; dy = e[:, 1:2] - py[None]
    power = -0.5 * (e[:, 2:3] * dx * dx + e[:, 4:5] * dy * dy) - e[:, 3:4] * dx * dy
    alpha = torch.clamp(e[:, 5:6] * torch.exp(power), max=0.99)
    act = (power <= 0) & (alpha >= 1.0 / 255.0)                         # [L,256]
    last = ncd[t].reshape(-1)                                           # n_contrib per pixel
    idx = torch.arange(hi - lo, device=dev)[:, None]
    live_f = idx < last[None]                                           # entries a pixel still looks at (fwd ~ bwd)
    # quadrant mask as the kernels compute it
    A, B, C_, o = e[:, 2], e[:, 3], e[:, 4], e[:, 5]
    det = A * C_ - B * B
    tau = 2.0 * torch.log(torch.clamp(o * 255.0, min=1e-9)) * 1.02 + 0.05
    x0, y0 = tx * 16.0 - e[:, 0], ty * 16.0 - e[:, 1]
    def edge_min(P, Q, R, fixed, lo_, hi_):
        tt = torch.minimum(torch.maximum(-Q * fixed / R, lo_), hi_)
        return P * fixed * fixed + 2 * Q * fixed * tt + R * tt * tt
    def reach(u0, u1, v0, v1):
        inside = (u0 <= 0) & (u1 >= 0) & (v0 <= 0) & (v1 >= 0)
        m = torch.minimum(torch.minimum(edge_min(A, B, C_, u0, v0, v1), edge_min(A, B, C_, u1, v0, v1)),
                          torch.minimum(edge_min(C_, B, A, v0, u0, u1), edge_min(C_, B, A, v1, u0, u1)))
        return inside | (m <= tau)
    ul, ur, vt, vb = (x0 - 0.5, x0 + 7.5), (x0 + 7.5, x0 + 15.5), (y0 - 0.5, y0 + 7.5), (y0 + 7.5, y0 + 15.5)
    qm = torch.stack([reach(*ul, *vt), reach(*ur, *vt), reach(*ul, *vb), reach(*ur, *vb)], 1) & (o * 255.0 >= 1.0)[:, None]
    qm = qm | (det <= 0)[:, None]
    tot["R"] += hi - lo
    for q in range(4):
        pm = quad == q
        lastq = last[pm].max()
        vis = qm[:, q] & (idx[:, 0] < lastq)
        a = act[:, pm] & live_f[:, pm]
        aw = a.any(1)
        tot["visits_fwd"] += int(vis.sum()); tot["actw_fwd"] += int((aw & vis).sum()); tot["actl_fwd"] += int(a[vis].sum())
        visb = qm[:, q] & (idx[:, 0] < last.max())
        tot["visits_bwd"] += int(visb.sum()); tot["actw_bwd"] += int((aw & visb).sum()); tot["actl_bwd"] += int(a[visb].sum())
print(tot)
print("fwd: visits/4R %.3f  active waves/visits %.3f  active lanes/(64*active waves) %.3f" % (
    tot["visits_fwd"] / (4 * tot["R"]), tot["actw_fwd"] / tot["visits_fwd"], tot["actl_fwd"] / (64 * tot["actw_fwd"])))
print("bwd: visits/4R %.3f  active waves/visits %.3f  active lanes/(64*active waves) %.3f" % (
    tot["visits_bwd"] / (4 * tot["R"]), tot["actw_bwd"] / tot["visits_bwd"], tot["actl_bwd"] / (64 * tot["actw_bwd"])))
