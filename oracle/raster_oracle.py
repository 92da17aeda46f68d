"""ctypes front end of the CPU oracle (oracle/raster_ref.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by dimo_amd/ (the product path).

PARITY UNPINNED for the rasterizer (see raster_ref.c header): the reference's
CUDA rasterizers are absent from /root/reference and it has no tests.

Provides
  * stage-by-stage numpy entry points (preprocess / bin / blend fwd / blend bwd /
    preprocess bwd) in fp32 or fp64,
  * `OracleRasterize`, a torch.autograd.Function on CPU tensors with the
    diff_gauss call surface (renderer/latent_gs_renderer.py:1255-1266),
  * knn / dist2 restatements.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
NFEAT = 7
TILE = 16


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    out = os.path.join(_HERE, "_build", "libraster_ref_f32.so")
    src = os.path.join(_HERE, "raster_ref.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib(f64=False):
    key = "f64" if f64 else "f32"
    if key not in _LIBS:
        path = os.path.join(_HERE, "_build", f"libraster_ref_{key}.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ref_preprocess_forward.restype = C.c_int64
        L.ref_bin.restype = C.c_int
        assert L.ref_sizeof_real() == (8 if f64 else 4)
        _LIBS[key] = L
    return _LIBS[key]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _rt(f64):
    return (np.float64, C.c_double) if f64 else (np.float32, C.c_float)


def _arr(x, dt):
    return None if x is None else np.ascontiguousarray(np.asarray(x, dtype=dt))


def preprocess_forward(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, scale_mod,
                       view, proj, campos, tanfovx, tanfovy, H, W, deg, f64=False):
    dt, ct = _rt(f64)
    L = lib(f64)
    means3D = _arr(means3D, dt).reshape(-1, 3)
    N = means3D.shape[0]
    shs = _arr(shs, dt)
    M = 0 if shs is None else shs.reshape(N, -1, 3).shape[1]
    colors_precomp = _arr(colors_precomp, dt)
    opacities = _arr(opacities, dt).reshape(-1)
    scales, rotations, cov3D_precomp = _arr(scales, dt), _arr(rotations, dt), _arr(cov3D_precomp, dt)
    view, proj, campos = _arr(view, dt).reshape(16), _arr(proj, dt).reshape(16), _arr(campos, dt).reshape(3)
    st = dict(
        N=N, M=M, deg=deg, H=H, W=W, f64=f64,
        means3D=means3D, shs=shs, colors_precomp=colors_precomp, opacities=opacities, scales=scales,
        rotations=rotations, cov3D_precomp=cov3D_precomp, scale_mod=float(scale_mod), view=view, proj=proj,
        campos=campos, tanfovx=float(tanfovx), tanfovy=float(tanfovy),
        radii=np.zeros(N, np.int32), xy=np.zeros((N, 2), dt), conic_op=np.zeros((N, 4), dt),
        feat=np.zeros((N, NFEAT), dt), rect=np.zeros((N, 4), np.int32), tiles_touched=np.zeros(N, np.uint32),
        offsets=np.zeros(N, np.uint32), clamped=np.zeros((N, 3), np.uint8), cov3d=np.zeros((N, 6), dt),
        normal_axis_sign=np.zeros((N, 2), np.int8),
    )
    R = L.ref_preprocess_forward(
        N, deg, M, H, W, _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales), _p(rotations),
        _p(cov3D_precomp), ct(scale_mod), _p(view), _p(proj), _p(campos), ct(tanfovx), ct(tanfovy),
        _p(st["radii"]), _p(st["xy"]), _p(st["conic_op"]), _p(st["feat"]), _p(st["rect"]), _p(st["tiles_touched"]),
        _p(st["offsets"]), _p(st["clamped"]), _p(st["cov3d"]), _p(st["normal_axis_sign"]))
    st["R"] = int(R)
    return st


def bin_tiles(st):
    L = lib(st["f64"])
    R, H, W = st["R"], st["H"], st["W"]
    T = ((W + TILE - 1) // TILE) * ((H + TILE - 1) // TILE)
    st["keys_unsorted"] = np.zeros(max(R, 1), np.uint64)
    st["vals_unsorted"] = np.zeros(max(R, 1), np.uint32)
    st["keys_sorted"] = np.zeros(max(R, 1), np.uint64)
    st["vals_sorted"] = np.zeros(max(R, 1), np.uint32)
    st["ranges"] = np.zeros((T, 2), np.uint32)
    rc = L.ref_bin(st["N"], H, W, C.c_int64(R), _p(st["radii"]), _p(st["feat"]), _p(st["rect"]), _p(st["offsets"]),
                   _p(st["keys_unsorted"]), _p(st["vals_unsorted"]), _p(st["keys_sorted"]), _p(st["vals_sorted"]),
                   _p(st["ranges"]))
    assert rc == 0
    for k in ("keys_unsorted", "vals_unsorted", "keys_sorted", "vals_sorted"):
        st[k] = st[k][:R]
    return st


def blend_forward(st, bg):
    dt, _ = _rt(st["f64"])
    L = lib(st["f64"])
    H, W = st["H"], st["W"]
    st["bg"] = _arr(bg, dt).reshape(3)
    st["out_color"] = np.zeros((3, H, W), dt)
    st["out_depth"] = np.zeros((1, H, W), dt)
    st["out_normal"] = np.zeros((3, H, W), dt)
    st["out_alpha"] = np.zeros((1, H, W), dt)
    st["final_T"] = np.zeros((H, W), dt)
    st["n_contrib"] = np.zeros((H, W), np.uint32)
    L.ref_blend_forward(H, W, _p(st["ranges"]), _p(st["vals_sorted"]), _p(st["xy"]), _p(st["conic_op"]),
                        _p(st["feat"]), _p(st["bg"]), _p(st["out_color"]), _p(st["out_depth"]), _p(st["out_normal"]),
                        _p(st["out_alpha"]), _p(st["final_T"]), _p(st["n_contrib"]))
    return st


def forward(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, scale_mod, view, proj,
            campos, bg, tanfovx, tanfovy, H, W, deg, f64=False):
    st = preprocess_forward(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, scale_mod,
                            view, proj, campos, tanfovx, tanfovy, H, W, deg, f64)
    bin_tiles(st)
    blend_forward(st, bg)
    return st


def blend_backward(st, dL_dcolor, dL_ddepth, dL_dnormal, dL_dalpha):
    dt, _ = _rt(st["f64"])
    L = lib(st["f64"])
    N, H, W = st["N"], st["H"], st["W"]
    g = dict(dL_dmean2D=np.zeros((N, 2), dt), dL_dconic=np.zeros((N, 3), dt), dL_dopacity=np.zeros(N, dt),
             dL_dfeat=np.zeros((N, NFEAT), dt))
    dc, dd, dn, da = (_arr(x, dt) for x in (dL_dcolor, dL_ddepth, dL_dnormal, dL_dalpha))
    L.ref_blend_backward(H, W, _p(st["ranges"]), _p(st["vals_sorted"]), _p(st["xy"]), _p(st["conic_op"]),
                         _p(st["feat"]), _p(st["bg"]), _p(st["final_T"]), _p(st["n_contrib"]), _p(dc), _p(dd), _p(dn),
                         _p(da), _p(g["dL_dmean2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dfeat"]))
    return g


def preprocess_backward(st, g):
    dt, ct = _rt(st["f64"])
    L = lib(st["f64"])
    N, M = st["N"], st["M"]
    g["dL_dmeans3D"] = np.zeros((N, 3), dt)
    g["dL_dshs"] = np.zeros((N, max(M, 1), 3), dt) if st["shs"] is not None else None
    g["dL_dcolors"] = np.zeros((N, 3), dt) if st["colors_precomp"] is not None else None
    g["dL_dscales"] = np.zeros((N, 3), dt)
    g["dL_drot"] = np.zeros((N, 4), dt)
    g["dL_dcov3D"] = np.zeros((N, 6), dt)
    L.ref_preprocess_backward(
        N, st["deg"], M, st["H"], st["W"], _p(st["means3D"]), _p(st["shs"]), _p(st["colors_precomp"]),
        _p(st["scales"]), _p(st["rotations"]), _p(st["cov3D_precomp"]), ct(st["scale_mod"]), _p(st["view"]),
        _p(st["proj"]), _p(st["campos"]), ct(st["tanfovx"]), ct(st["tanfovy"]), _p(st["radii"]), _p(st["cov3d"]),
        _p(st["clamped"]), _p(st["normal_axis_sign"]), _p(g["dL_dmean2D"]), _p(g["dL_dconic"]), _p(g["dL_dfeat"]),
        _p(g["dL_dmeans3D"]), _p(g["dL_dshs"]), _p(g["dL_dcolors"]), _p(g["dL_dscales"]), _p(g["dL_drot"]),
        _p(g["dL_dcov3D"]))
    return g


def backward(st, dL_dcolor, dL_ddepth, dL_dnormal, dL_dalpha):
    g = blend_backward(st, dL_dcolor, dL_ddepth, dL_dnormal, dL_dalpha)
    return preprocess_backward(st, g)


def knn(ref, query, k, f64=False):
    """knn_cuda.KNN(k, transpose_mode=True) on [M,3] / [N,3] -> (dist[N,k], idx[N,k] int64)."""
    dt, _ = _rt(f64)
    ref, query = _arr(ref, dt).reshape(-1, 3), _arr(query, dt).reshape(-1, 3)
    N = query.shape[0]
    dist, idx = np.zeros((N, k), dt), np.zeros((N, k), np.int64)
    lib(f64).ref_knn(ref.shape[0], N, k, _p(ref), _p(query), _p(dist), _p(idx))
    return dist, idx


def dist2(points, f64=False):
    """simple_knn._C.distCUDA2 on [N,3] -> [N]."""
    dt, _ = _rt(f64)
    pts = _arr(points, dt).reshape(-1, 3)
    out = np.zeros(pts.shape[0], dt)
    lib(f64).ref_dist2(pts.shape[0], _p(pts), _p(out))
    return out


# --------------------------------------------------------------------------- torch front end (CPU)
def _torch():
    import torch
    return torch


class _OracleRasterizeFn:
    """Built lazily so that importing this module does not import torch."""
    _cls = None

    @classmethod
    def get(cls):
        if cls._cls is not None:
            return cls._cls
        torch = _torch()

        class OracleRasterize(torch.autograd.Function):
            @staticmethod
            def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                        settings):
                f64 = means3D.dtype == torch.float64
                n = lambda t: None if t is None or t.numel() == 0 else t.detach().cpu().numpy()
                st = forward(n(means3D), n(shs), n(colors_precomp), n(opacities), n(scales), n(rotations),
                             n(cov3D_precomp), settings.scale_modifier, n(settings.viewmatrix),
                             n(settings.projmatrix), n(settings.campos), n(settings.bg), settings.tanfovx,
                             settings.tanfovy, settings.image_height, settings.image_width, settings.sh_degree, f64)
                ctx.st = st
                ctx.has = (shs is not None, colors_precomp is not None, cov3D_precomp is not None)
                t = lambda a: torch.from_numpy(a.copy())
                ctx.mark_non_differentiable(*[])
                return (t(st["out_color"]), t(st["out_depth"]), t(st["out_normal"]), t(st["out_alpha"]),
                        t(st["radii"]))

            @staticmethod
            def backward(ctx, g_color, g_depth, g_normal, g_alpha, _g_radii):
                st = ctx.st
                dt = np.float64 if st["f64"] else np.float32
                H, W = st["H"], st["W"]
                z = lambda g, c: np.zeros((c, H, W), dt) if g is None else g.detach().cpu().numpy()
                g = backward(st, z(g_color, 3), z(g_depth, 1), z(g_normal, 3), z(g_alpha, 1))
                t = lambda a: None if a is None else torch.from_numpy(a)
                N = st["N"]
                m2d = np.zeros((N, 3), dt)
                m2d[:, :2] = g["dL_dmean2D"]
                has_shs, has_col, has_cov = ctx.has
                return (t(g["dL_dmeans3D"]), t(m2d), t(g["dL_dshs"]) if has_shs else None,
                        t(g["dL_dcolors"]) if has_col else None, t(g["dL_dopacity"].reshape(N, 1)),
                        None if has_cov else t(g["dL_dscales"]), None if has_cov else t(g["dL_drot"]),
                        t(g["dL_dcov3D"]) if has_cov else None, None)

        cls._cls = OracleRasterize
        return OracleRasterize


def rasterize_torch(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings):
    """CPU torch front end with diff_gauss' return order minus `extra`:
    (image, depth, normal, alpha, radii)."""
    return _OracleRasterizeFn.get().apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                          cov3D_precomp, settings)
