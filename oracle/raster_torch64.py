"""TEST INFRASTRUCTURE ONLY -- an INDEPENDENT float64 restatement of the published 3D-Gaussian-splatting
rasterizer (project -> EWA splat -> tile membership -> front-to-back alpha blending) in plain torch, differentiated
by autograd.  It shares no code with oracle/raster_ref.c or dimo_amd/csrc/preprocess.hip (whose small helpers were
once transcribed from one another): the rotation comes from the textbook axis-angle/quaternion sandwich product, the
covariance from explicit matrix products, the gradients from autograd instead of a hand-written backward.  Tests
compare BOTH the C oracle and the HIP kernels with it on <= a few hundred Gaussians (tests/test_oracle_kat.py,
tests/test_gpu_kat.py).

PARITY UNPINNED like the C oracle (the CUDA sources are absent from /root/reference): this file follows the same
published algorithm -- Kerbl et al. 2023, sections 4-6 and appendix A; alpha / depth channels as in the
ashawkey fork; normal channel ASSUMED as documented in raster_ref.c -- and the reference's call-site contract
(renderer/latent_gs_renderer.py:1132-1163, 1255-1277).  Conventions: row-vector matrices as MiniCam builds them
(p_view = [x y z 1] @ viewmatrix), quaternions (w, x, y, z), SH degree 0 colour = max(C0 * sh + 0.5, 0).
"""
import math

import torch

TILE = 16
SH_C0 = 0.28209479177387814


def rotation_from_quaternion(q):
    """Quaternion (w, x, y, z) -> rotation matrix, via v' = q v q* applied to the three basis vectors.  The
    published kernel does NOT normalise (the caller hands it unit quaternions: `get_rotation`), and its gradient is
    that of the polynomial I + 2 w [v]x + 2 [v]x^2 -- which the Rodrigues form below is, term for term."""
    w, v = q[..., :1], q[..., 1:]
    cols = []
    for k in range(3):
        e = torch.zeros_like(v)
        e[..., k] = 1.0
        # Rodrigues form of the sandwich product: e + 2 w (v x e) + 2 v x (v x e)
        t = 2.0 * torch.cross(v, e, dim=-1)
        cols.append(e + w * t + torch.cross(v, t, dim=-1))
    return torch.stack(cols, dim=-1)  # columns = images of the basis vectors


def project(means3D, scales, rotations, opacities, shs, view, proj, campos, tanfovx, tanfovy, H, W, scale_mod=1.0):
    """Per-Gaussian quantities; everything differentiable except the integer outputs."""
    N = means3D.shape[0]
    dt = means3D.dtype
    hom = torch.cat([means3D, torch.ones(N, 1, dtype=dt)], dim=1)
    p_view = hom @ view
    p_clip = hom @ proj
    ndc = p_clip[:, :3] / (p_clip[:, 3:4] + 1e-7)
    depth = p_view[:, 2]
    in_front = depth > 0.2
    Rm = rotation_from_quaternion(rotations)
    S = torch.diag_embed((scale_mod * scales) ** 2)
    Sigma = Rm @ S @ Rm.transpose(1, 2)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    z = depth
    lim_x, lim_y = 1.3 * tanfovx, 1.3 * tanfovy
    # EWA Jacobian at the view-space point with x/z, y/z clamped to 1.3 tan(fov/2).  The published backward treats a
    # clamped coordinate as a CONSTANT (x_grad_mul = 0: it drops d(lim z)/dz as well), so it is detached here; an
    # unclamped one is the view coordinate itself.
    rx, ry = (p_view[:, 0] / z).detach(), (p_view[:, 1] / z).detach()
    tx = torch.where(rx.abs() > lim_x, (torch.clamp(rx, -lim_x, lim_x) * z).detach(), p_view[:, 0])
    ty = torch.where(ry.abs() > lim_y, (torch.clamp(ry, -lim_y, lim_y) * z).detach(), p_view[:, 1])
    zero = torch.zeros_like(z)
    J = torch.stack([torch.stack([fx / z, zero, -fx * tx / (z * z)], -1),
                     torch.stack([zero, fy / z, -fy * ty / (z * z)], -1)], dim=1)  # [N, 2, 3]
    Wr = view[:3, :3].T  # world -> view rotation acting on column vectors
    M = J @ Wr
    cov2 = M @ Sigma @ M.transpose(1, 2) + 0.3 * torch.eye(2, dtype=dt)
    a, b, c = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam.detach()))
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    x0 = torch.clamp(((px.detach() - radius) / TILE).to(torch.int64), 0, gx)
    x1 = torch.clamp(((px.detach() + radius + TILE - 1) / TILE).to(torch.int64), 0, gx)
    y0 = torch.clamp(((py.detach() - radius) / TILE).to(torch.int64), 0, gy)
    y1 = torch.clamp(((py.detach() + radius + TILE - 1) / TILE).to(torch.int64), 0, gy)
    visible = in_front & (det.detach() != 0) & ((x1 - x0) * (y1 - y0) > 0)
    color = torch.clamp(SH_C0 * shs[:, 0, :] + 0.5, min=0.0)
    # normal (assumption, see raster_ref.c): axis of the smallest scale, turned towards the camera, in view space
    k = torch.argmin(scales.detach(), dim=1)  # first minimum = lowest axis on ties
    n_world = Rm[torch.arange(N), :, k]
    towards = ((campos[None] - means3D) * n_world).sum(1)
    n_world = torch.where((towards < 0)[:, None], -n_world, n_world)
    n_view = n_world @ view[:3, :3]
    return dict(px=px, py=py, conic=conic, depth=depth, color=color, normal=n_view, opacity=opacities.reshape(-1),
                radius=torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32), visible=visible,
                rect=torch.stack([x0, y0, x1, y1], 1), cov2=cov2)


def blend(g, bg, H, W):
    """Front-to-back compositing with the published thresholds; a Gaussian reaches a pixel only through the tiles of
    its rectangle.  Order: view depth (as float32 bits, like the sort key), ties by index."""
    dt = g["px"].dtype
    idx = torch.nonzero(g["visible"]).reshape(-1)
    key = g["depth"].detach()[idx].to(torch.float32)
    order = idx[torch.sort(key, stable=True).indices]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    tile_x, tile_y = (xs / TILE).floor().to(torch.int64), (ys / TILE).floor().to(torch.int64)
    T = torch.ones(H, W, dtype=dt)
    alive = torch.ones(H, W, dtype=torch.bool)
    C = torch.zeros(3, H, W, dtype=dt)
    D = torch.zeros(H, W, dtype=dt)
    Nn = torch.zeros(3, H, W, dtype=dt)
    A = torch.zeros(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    seen = torch.zeros(H, W, dtype=torch.int64)  # entries of the pixel's tile list walked so far
    for i in order.tolist():
        x0, y0, x1, y1 = g["rect"][i].tolist()
        in_tiles = (tile_x >= x0) & (tile_x < x1) & (tile_y >= y0) & (tile_y < y1)
        walk = in_tiles & alive
        seen = seen + walk.to(torch.int64)
        dx, dy = g["px"][i] - xs, g["py"][i] - ys
        ca, cb, cc = g["conic"][i]
        power = -0.5 * (ca * dx * dx + cc * dy * dy) - cb * dx * dy
        alpha = torch.clamp(g["opacity"][i] * torch.exp(power), max=0.99)
        hit = walk & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
        test_T = T * (1.0 - alpha)
        stop = hit & (test_T.detach() < 1e-4)
        add = hit & ~stop
        w = torch.where(add, alpha * T, torch.zeros_like(T))
        C = C + g["color"][i][:, None, None] * w
        D = D + g["depth"][i] * w
        Nn = Nn + g["normal"][i][:, None, None] * w
        A = A + w
        T = torch.where(add, test_T, T)
        n_contrib = torch.where(add, seen, n_contrib)
        alive = alive & ~stop
    image = C + T * bg[:, None, None]
    return dict(image=image, depth=D[None], normal=Nn, alpha=A[None], final_T=T, n_contrib=n_contrib)


def render(means3D, scales, rotations, opacities, shs, view, proj, campos, bg, tanfovx, tanfovy, H, W, scale_mod=1.0):
    g = project(means3D, scales, rotations, opacities, shs, view, proj, campos, tanfovx, tanfovy, H, W, scale_mod)
    out = blend(g, bg, H, W)
    out["geom"] = g
    return out


def closed_form_conic_on_axis(z, focal, sx, sy, theta):
    """A Gaussian on the optical axis at view depth z whose first two principal axes lie in the image plane, rotated
    by theta about the viewing direction: 2D covariance = (f/z)^2 R(theta) diag(sx^2, sy^2) R(theta)^T + 0.3 I."""
    k = (focal / z) ** 2
    c, s = math.cos(theta), math.sin(theta)
    a = k * (c * c * sx * sx + s * s * sy * sy) + 0.3
    b = k * (c * s * (sx * sx - sy * sy))
    d = k * (s * s * sx * sx + c * c * sy * sy) + 0.3
    det = a * d - b * b
    return d / det, -b / det, a / det
