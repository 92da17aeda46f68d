"""No-GPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/dimo_hip.h declares; size/layout queries (host-only code) behave; ops fail loudly
on CPU tensors (no silent fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dimo_amd.csrc.build import build
    build()
    from dimo_amd import _lib
    return _lib.lib()


def test_header_symbols_all_exported(lib):
    from dimo_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dimo_hip.h")).read()
    declared = set(re.findall(r"\b(dimo_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} not exported"
    assert lib.dimo_version().startswith(b"dimo_hip gfx950")


def test_layout_queries(lib):
    g, b, i = (C.c_size_t * 6)(), (C.c_size_t * 3)(), (C.c_size_t * 2)()
    assert lib.dimo_raster_geom_layout(1000, g) == 0
    assert lib.dimo_raster_bin_layout(50000, 128, 96, b) == 0
    assert lib.dimo_raster_img_layout(128, 96, i) == 0
    for arr in (g, b, i):
        offs = list(arr)
        assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert g[1] - g[0] >= 1000 * 64 and b[1] - b[0] >= 50000 * 4 and i[1] - i[0] >= 128 * 96 * 4
    assert lib.dimo_raster_geom_bytes(1000) > g[5]
    assert lib.dimo_raster_bin_bytes(1000, 50000, 128, 96) > b[2]
    # the bucket regions of the unsorted level-1 array follow the model's size (64 KB per (supertile, depth bin) bucket):
    # a 100 k-Gaussian model at 512^2 reserves 512 of them, not the layout's limit of 2048
    small, big = lib.dimo_raster_bin_bytes(100_000, 1 << 20, 512, 512), lib.dimo_raster_bin_bytes(2_000_000, 1 << 20, 512, 512)
    assert big - small == (2048 - 512) * 4096 * 16
    assert lib.dimo_raster_geom_bytes(0) > 0  # empty scenes still get a valid workspace
    assert lib.dimo_raster_bin_layout(-1, 128, 96, b) == -1  # DIMO_E_ARG


def test_no_cpu_fallback():
    from dimo_amd import rasterizer as rz
    from dimo_amd.fused_ssim import fused_ssim
    from dimo_amd.knn_cuda import KNN
    from dimo_amd.simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        distCUDA2(torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        KNN(4, True)(torch.zeros(1, 8, 3), torch.zeros(1, 5, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fused_ssim(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 16, 16))
    s = rz.GaussianRasterizationSettings(16, 16, 0.3, 0.3, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                         torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rz.GaussianRasterizerNormal(s)(means3D=torch.zeros(2, 3), means2D=torch.zeros(2, 3), shs=torch.zeros(2, 1, 3),
                                       colors_precomp=None, opacities=torch.zeros(2, 1), scales=torch.ones(2, 3),
                                       rotations=torch.ones(2, 4), cov3Ds_precomp=None, extra_attrs=None)


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (parity claims depend on it)."""
    pkg = os.path.join(ROOT, "dimo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)
                assert "raster_oracle" not in src and "raster_ref" not in src, os.path.join(dp, f)
