"""Runs the binning chain of dimo_amd/csrc/binning.hip on the CPU SIMT emulation (tests/simt/build.py) for the
not-gpu tests: inputs are what the projection kernel leaves per Gaussian (tile rectangle, tiles touched, depth bits,
the per-block words), outputs the per-tile lists.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None
PRE_BLOCK = 256



def workspace(nbytes, fill=0x5A, align=256):
    """`nbytes` of host memory on a 256-byte boundary, as hipMalloc and torch's allocator hand out (include/dimo_hip.h:
    the kernels use 16- and 64-byte vector accesses on their workspaces).  Under AddressSanitizer
    (tools/emulated_asan.sh) the slack in front of and behind the view is poisoned, so an access one byte past the
    workspace is still reported."""
    raw = np.full(nbytes + align, fill, np.uint8)
    off = (-raw.ctypes.data) % align
    view = raw[off:off + nbytes]
    poison = getattr(C.CDLL(None), "__asan_poison_memory_region", None) if os.environ.get("SIMT_ASAN") else None
    if poison is not None:
        poison.argtypes = [C.c_void_p, C.c_size_t]
        poison(raw.ctypes.data, off)
        poison(raw.ctypes.data + off + nbytes, align - off)
    return view

def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_build.build(target="binning"))
        L.simt_geom_layout.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
        L.simt_bin_layout.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        L.simt_supertile_shift.argtypes = [C.c_int, C.c_int]
        L.simt_bin_instances.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.simt_bin_instances_batched.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t,
                                                 C.c_size_t, C.c_void_p]
        L.simt_bwd_scratch_bytes.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.simt_bwd_scratch_bytes.restype = C.c_size_t
        L.simt_depth_keys.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _layouts(N, H, W, R_cap):
    L = lib()
    g = (C.c_size_t * 10)()
    b = (C.c_size_t * 14)()
    L.simt_geom_layout(N, g)
    L.simt_bin_layout(N, R_cap, H, W, b)
    G = dict(zip(("rect", "tiles", "offsets", "total", "block_sums", "key32", "bk", "bytes", "nb", "bk_tot"), [int(x) for x in g]))
    B = dict(zip(("vals", "ranges", "totals", "order", "bytes", "T", "cap", "l1tmp", "l1list", "meta", "grpbase", "grpinfo",
                  "cntu", "l1cap"), [int(x) for x in b]))
    return G, B


def make_geom(rect, tiles, key32, H, W, G):
    """The geometry workspace as preprocess_fwd leaves it (preprocess.hip:125-157)."""
    N = len(tiles)
    ssh = lib().simt_supertile_shift(H, W)
    assert ssh >= 0
    r16, t32, key, sums = geom_words(rect, tiles, key32, ssh, G["nb"])
    geom = np.zeros(G["bytes"], np.uint8)
    geom[G["rect"]:G["rect"] + 8 * N] = r16.reshape(-1).view(np.uint8)
    geom[G["tiles"]:G["tiles"] + 4 * N] = t32.view(np.uint8)
    geom[G["key32"]:G["key32"] + 4 * N] = key.view(np.uint8)
    geom[G["block_sums"]:G["block_sums"] + sums.nbytes] = sums.reshape(-1).view(np.uint8)
    return geom


def _read(buf, off, n, dt=np.uint32):
    return buf[off:off + n * np.dtype(dt).itemsize].view(dt).copy()


def run_binning(rect, tiles, key32, H, W, R_cap=None, n_batched=0, poison=True):
    """rect int[N,4] (tile units: x0 y0 x1 y1), tiles uint32[N], key32 uint32[N] (depth bits).  Returns a dict (or a list
    of dicts, one per render, for n_batched > 0: the batched kernels over n copies of the inputs)."""
    rect, tiles, key32 = np.asarray(rect), np.asarray(tiles, np.uint32), np.asarray(key32, np.uint32)
    N = len(tiles)
    R = int(tiles.astype(np.int64).sum())
    if R_cap is None:
        R_cap = max(R, 1)
    G, B = _layouts(N, H, W, R_cap)
    L = lib()
    n = max(n_batched, 1)
    geoms = [make_geom(rect, tiles, key32, H, W, G) for _ in range(n)]
    # (stale contents from an earlier use of the workspace must not matter)
    bins = [workspace(B["bytes"], 0xA5 if poison else 0) for _ in range(n)]
    if n_batched:
        sb = int(L.simt_bwd_scratch_bytes(R_cap, H, W))
        scr = [workspace(sb, 0x5A) for _ in range(n)]
        arr = lambda bufs: (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        totals = np.zeros(2 * n, np.uint32)
        rc = L.simt_bin_instances_batched(N, H, W, R_cap, n, arr(geoms), arr(bins), arr(scr), G["bytes"], B["bytes"], sb,
                                          totals.ctypes.data)
    else:
        rc = L.simt_bin_instances(N, H, W, R_cap, geoms[0].ctypes.data, bins[0].ctypes.data)
    assert rc == 0, rc
    out = []
    for i in range(n):
        tot = _read(geoms[i], G["total"], 4)
        Rk = min(int(tot[0]), R_cap)
        dk = np.zeros(max(Rk, 1), np.uint32)  # the instances' depth bits: gathered on request (not stored per instance)
        if not int(tot[1]):
            assert L.simt_depth_keys(N, H, W, R_cap, geoms[i].ctypes.data, bins[i].ctypes.data, dk.ctypes.data) == 0
        d = dict(R=int(tot[0]), overflow=int(tot[1]), entries=int(tot[2]), offsets=_read(geoms[i], G["offsets"], N),
                 dkeys=dk[:Rk], vals=_read(bins[i], B["vals"], Rk),
                 ranges=_read(bins[i], B["ranges"], 2 * B["T"]).reshape(-1, 2), order=_read(bins[i], B["order"], B["T"]),
                 bk_tot=_read(geoms[i], G["bk_tot"], 2 * 2048), bk=_read(geoms[i], G["bk"], 8),
                 l1list=_read(bins[i], B["l1list"], 4 * min(B["l1cap"], 4 * N + 65536)).reshape(-1, 4),
                 l1tmp=_read(bins[i], B["l1tmp"], 4 * min(B["l1cap"], 4 * N + 65536)).reshape(-1, 4),
                 meta=_read(bins[i], B["meta"], 16), totals=_read(bins[i], B["totals"], B["T"]))
        if n_batched:
            d["totals_out"] = totals[2 * i:2 * i + 2].copy()
            d["flags"] = scr[i][sb - ((B["cap"] + 255) // 256 * 256):][:Rk].copy()
        out.append(d)
    return out if n_batched else out[0]


def expected(rect, tiles, key32, H, W):
    """The published order in numpy: instances sorted by (tile, depth bits), stable in the Gaussian index."""
    TILE = 16
    tx = (W + TILE - 1) // TILE
    T = tx * ((H + TILE - 1) // TILE)
    ids, tl = [], []
    for i in np.nonzero(np.asarray(tiles) > 0)[0]:
        x0, y0, x1, y1 = (int(v) for v in rect[i])
        yy, xx = np.mgrid[y0:y1, x0:x1]
        t = (yy * tx + xx).reshape(-1)
        tl.append(t), ids.append(np.full(len(t), i, np.int64))
    if not ids:
        return dict(R=0, dkeys=np.zeros(0, np.uint32), vals=np.zeros(0, np.uint32), ranges=np.zeros((T, 2), np.uint32))
    ids, tl = np.concatenate(ids), np.concatenate(tl)
    k = (tl.astype(np.uint64) << np.uint64(32)) | np.asarray(key32, np.uint64)[ids]
    o = np.argsort(k, kind="stable")
    ks, vs = k[o], ids[o]
    tiles_sorted = (ks >> np.uint64(32)).astype(np.int64)
    ranges = np.zeros((T, 2), np.uint32)
    cnt = np.bincount(tiles_sorted, minlength=T)
    end = np.cumsum(cnt)
    nz = cnt > 0
    ranges[nz, 0] = (end - cnt)[nz]
    ranges[nz, 1] = end[nz]
    return dict(R=len(ks), dkeys=(ks & np.uint64(0xFFFFFFFF)).astype(np.uint32), vals=vs.astype(np.uint32), ranges=ranges)


def fuzz_config(rng, max_instances=1_500_000):
    """One random binning input (H, W, rect, tiles, key32): image sizes from 16^2 to 1300 x 2048 (every supertile edge,
    partial supertiles, more and fewer than 32 buckets), 1 to 20 000 Gaussians, rectangles of a tile / a few tiles / the
    whole image, visible fractions down to 5 %, depths uniform / one value / three values / a thin slab with floaters."""
    while True:
        H = int(rng.choice([16, 33, 64, 100, 128, 200, 256, 400, 512, 777, 1024, 1300]))
        W = int(rng.choice([16, 48, 64, 96, 128, 250, 256, 512, 640, 1024, 2048]))
        tx, ty = (W + 15) // 16, (H + 15) // 16
        N = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 3000, 8000, 20000]))
        mode, km = int(rng.integers(0, 5)), int(rng.integers(0, 4))
        cx, cy = rng.integers(0, tx, N), rng.integers(0, ty, N)
        if mode == 0:
            ext = rng.integers(1, 3, (N, 2))
        elif mode == 1:
            ext = rng.integers(1, max(2, min(tx, ty)), (N, 2))
        elif mode == 2:
            ext = np.stack([np.full(N, tx), np.full(N, ty)], 1)
        elif mode == 3:
            ext = rng.integers(1, 6, (N, 2))
        else:
            ext = np.where(rng.random((N, 1)) < 0.02, max(tx, ty), rng.integers(1, 3, (N, 2)))
        x0 = np.clip(cx - ext[:, 0] // 2, 0, tx - 1)
        y0 = np.clip(cy - ext[:, 1] // 2, 0, ty - 1)
        x1, y1 = np.clip(x0 + ext[:, 0], 1, tx), np.clip(y0 + ext[:, 1], 1, ty)
        rect = np.stack([x0, y0, np.maximum(x1, x0 + 1), np.maximum(y1, y0 + 1)], 1).astype(np.int32)
        vis = rng.random(N) < rng.choice([1.0, 0.9, 0.5, 0.05])
        tiles = np.where(vis, (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1]), 0).astype(np.uint32)
        if km == 0:
            key = rng.uniform(0.3, 50.0, N).astype(np.float32).view(np.uint32)
        elif km == 1:
            key = np.full(N, np.float32(2.5).view(np.uint32))
        elif km == 2:
            key = rng.choice(np.array([1.0, 1.5, 3.0], np.float32), N).view(np.uint32)
        else:
            k = rng.normal(2.0, 0.01, N).astype(np.float32)
            k[: max(1, N // 500)] = 90.0
            k[-1] = 0.21
            key = np.abs(k).astype(np.float32).view(np.uint32)
        if int(tiles.astype(np.int64).sum()) <= max_instances:
            return H, W, rect, tiles, key, (H, W, N, mode, km)


def geom_words(rect, tiles, key32, ssh, nb):
    """What preprocess_fwd leaves for the binning (preprocess.hip:125-157), as numpy arrays: rect u16 [N,4], tiles u32,
    key32 u32 (0xffffffff without tiles), the four per-block rows [4, nb + 1]."""
    N = len(tiles)
    vis = tiles > 0
    key = np.where(vis, key32, 0xFFFFFFFF).astype(np.uint32)
    r16 = np.where(vis[:, None], rect, 0).astype(np.uint16)
    pad = nb * PRE_BLOCK - N
    x0, y0, x1, y1 = (rect[:, i].astype(np.int64) for i in range(4))
    ent = np.where(vis, (((x1 - 1) >> ssh) - (x0 >> ssh) + 1) * (((y1 - 1) >> ssh) - (y0 >> ssh) + 1), 0)
    blk = lambda a, fill: np.concatenate([a, np.full(pad, fill, a.dtype)]).reshape(nb, PRE_BLOCK)
    sums = np.zeros((4, nb + 1), np.uint32)
    sums[0, :nb] = blk(tiles.astype(np.uint32), 0).sum(1)
    sums[1, :nb] = blk(key, 0xFFFFFFFF).min(1)
    sums[2, :nb] = blk(np.where(vis, key, 0).astype(np.uint32), 0).max(1)
    sums[3, :nb] = blk(ent.astype(np.uint32), 0).sum(1)
    return r16, tiles.astype(np.uint32), key, sums
