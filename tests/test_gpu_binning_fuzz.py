"""GPU: the tile binning ALONE (dimo_debug_bin_instances) on seeded random inputs written straight into the geometry
workspace -- shapes the projection of a plausible scene does not produce (every Gaussian over the whole image, one depth
for 20 000 Gaussians, 5 % visible, images of one tile) -- against a numpy restatement of the published order: ids and tile
ranges bit for bit.  The same generator drives tests/test_binning_emulated.py on the CPU SIMT emulation."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.simt import harness as hz

pytestmark = pytest.mark.gpu


def _run(L, rect, tiles, key, H, W, dev):
    N = len(tiles)
    R = int(tiles.astype(np.int64).sum())
    cap = max(R, 1)
    g = (C.c_size_t * 10)()
    b = (C.c_size_t * 3)()
    assert L.dimo_debug_bin_geom_layout(N, H, W, g) == 0 and L.dimo_raster_bin_layout(cap, H, W, b) == 0
    G = dict(zip(("rect", "tiles", "offsets", "total", "block_sums", "key32", "bk", "bytes", "nb", "ssh"), [int(x) for x in g]))
    r16, t32, k32, sums = hz.geom_words(rect, tiles, key, G["ssh"], G["nb"])
    host = np.zeros(G["bytes"], np.uint8)
    host[G["rect"]:G["rect"] + 8 * N] = r16.reshape(-1).view(np.uint8)
    host[G["tiles"]:G["tiles"] + 4 * N] = t32.view(np.uint8)
    host[G["key32"]:G["key32"] + 4 * N] = k32.view(np.uint8)
    host[G["block_sums"]:G["block_sums"] + sums.nbytes] = sums.reshape(-1).view(np.uint8)
    geom = torch.from_numpy(host).to(dev)
    bin_ws = torch.full((int(L.dimo_raster_bin_bytes(N, cap, H, W)),), 0xA5, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    assert L.dimo_debug_bin_instances(N, H, W, cap, geom.data_ptr(), bin_ws.data_ptr(), s) == 0
    dk = torch.zeros(cap, dtype=torch.int32, device=dev)
    assert L.dimo_raster_depth_keys(N, H, W, cap, geom.data_ptr(), bin_ws.data_ptr(), dk.data_ptr(), s) == 0
    torch.cuda.synchronize(dev)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    total = geom[G["total"]:G["total"] + 16].view(torch.int32).cpu().numpy().view(np.uint32)
    vals = bin_ws[int(b[0]):int(b[0]) + 4 * cap].view(torch.int32).cpu().numpy().view(np.uint32)[:R]
    ranges = bin_ws[int(b[1]):int(b[1]) + 8 * T].view(torch.int32).cpu().numpy().view(np.uint32).reshape(T, 2)
    return total, vals, ranges, dk.cpu().numpy().view(np.uint32)[:R]


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15])
def test_binning_fuzz_against_numpy(seed):
    from dimo_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    for it in range(12):
        H, W, rect, tiles, key, what = hz.fuzz_config(rng, max_instances=3_000_000)
        e = hz.expected(rect, tiles, key, H, W)
        total, vals, ranges, dk = _run(L, rect, tiles, key, H, W, dev)
        assert int(total[0]) == e["R"] and int(total[1]) == 0, (seed, it, what)
        assert np.array_equal(ranges, e["ranges"]), ("ranges", seed, it, what)
        assert np.array_equal(vals, e["vals"]), ("order", seed, it, what)
        assert np.array_equal(dk, e["dkeys"]), ("depth bits", seed, it, what)
