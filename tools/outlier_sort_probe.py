"""Level-1 binning cost (count + scatter + bucket sort) when a few far outliers stretch the key range:
    python tools/outlier_sort_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import ctypes as C
import numpy as np
import torch
from dimo_amd import _lib
from tests.scenes import random_scene, camera_np
from tests.test_gpu_raster import _run_hip
L = _lib.lib()
cam = camera_np(0.0, W=256, H=256)
for far in (0.0, 30.0, 1000.0, 1e5, -1.0):
    sc = random_scene(100000, seed=1, scale=0.01)
    if far:
        view = np.asarray(cam["view"], np.float64)
        axis = (view[:3, 2] / np.linalg.norm(view[:3, 2])).astype(np.float32)
        sc["means3D"][:8] += far * axis
    _run_hip(sc, cam, (0, 0, 0), 0)
    L.dimo_timing_select(None); L.dimo_timing_enable(1)
    for _ in range(5): _run_hip(sc, cam, (0, 0, 0), 0)
    torch.cuda.synchronize(); L.dimo_timing_enable(0)
    tot = {}
    for name in (b"scan", b"emit", b"sort"):
        ms, n = C.c_double(0), C.c_int64(0)
        L.dimo_timing_read(name, C.byref(ms), C.byref(n))
        tot[name.decode()] = 1e3 * ms.value / max(n.value, 1)
    print("outliers at +%g: level-1 count %.1f + scatter %.1f + bucket sort %.1f = %.1f us per render"
          % (far, tot["scan"], tot["emit"], tot["sort"], sum(tot.values())))
