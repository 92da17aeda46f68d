"""The TimeNet kernels (dimo_amd/csrc/timenet.hip as hipcc compiles it: the one-launch fp32-MFMA forward and dgrad
chain, the weight packers, the 64 x 64 weight-gradient tiles with the embedding backward, the per-layer GEMM path for
other shapes) run on the CPU SIMT emulation (tests/simt/) against the TimeNet module in float64 with autograd -- the
cases and the 1e-4 bar of tests/test_gpu_timenet.py.  The three MFMA builtins are wave rendezvous in the shim that
apply the matrix cores' documented operand layouts (which lane holds which element of A, B and D), so a wrong operand
mapping in the kernels fails here as it would on the GPU.  No GPU; the GPU tests stay the parity tests proper."""
import copy
import ctypes as C

import pytest
import torch

from dimo_amd.fused_timenet import MAX_LAYERS, TimeNetDesc, linear_layers
from tests.simt import build as simt_build
from tests.test_gpu_timenet import _close, _make, _reference

_T = None


def Tn():
    global _T
    if _T is None:
        lib = C.CDLL(simt_build.build(target="timenet"))
        p, i, z = C.c_void_p, C.c_int, C.c_size_t
        lib.dimo_timenet_workspace_bytes.argtypes = [p, i, i]
        lib.dimo_timenet_workspace_bytes.restype = z
        lib.dimo_timenet_forward.argtypes = [p, i, i, p, p, p, p, p, p, p, z, p]
        lib.dimo_timenet_backward.argtypes = [p, i, i, p, p, p, p, p, p, p, z, p]
        _T = lib
    return _T


class EmulatedTimeNet:
    """dimo_amd.fused_timenet.FusedTimeNet's calls on host tensors."""

    def __init__(self, net):
        self.net, self.layers = net, linear_layers(net)
        assert len(self.layers) <= MAX_LAYERS
        d = self.desc = TimeNetDesc()
        d.D, d.W = len(net.deformnet), net.deformnet[0].out_features
        d.skip = net.skips[0] if net.skips else -1
        d.pts_freqs, d.time_freqs = net.pts_ch, net.times_ch
        d.latent_dim = net.input_ch - 6 * net.pts_ch - 2 * net.times_ch
        for i, lin in enumerate(self.layers):
            for t in (lin.weight, lin.bias):
                t.grad = torch.zeros_like(t)
            d.weight[i], d.bias[i] = lin.weight.data_ptr(), lin.bias.data_ptr()
            d.g_weight[i], d.g_bias[i] = lin.weight.grad.data_ptr(), lin.bias.grad.data_ptr()

    def forward(self, c_xyz, times, table, rows):
        P, M = len(times), c_xyz.shape[0]
        n = Tn().dimo_timenet_workspace_bytes(C.addressof(self.desc), P, M)
        self.ws = torch.full((n,), 0x5A, dtype=torch.uint8)
        self.times = (C.c_float * P)(*[float(t) for t in times])
        self.rows = (C.c_int * P)(*[int(r) for r in rows])
        self.shape = (P, M)
        dx, dr = torch.full((P, M, 3), float("nan")), torch.full((P, M, 4), float("nan"))
        rc = Tn().dimo_timenet_forward(C.addressof(self.desc), P, M, c_xyz.data_ptr(), self.times, table.data_ptr(),
                                       self.rows, dx.data_ptr(), dr.data_ptr(), self.ws.data_ptr(), n, None)
        assert rc == 0
        return dx, dr

    def backward(self, gx, gr, g_c, g_tab):
        P, M = self.shape
        rc = Tn().dimo_timenet_backward(C.addressof(self.desc), P, M, gx.data_ptr(), gr.data_ptr(), self.times, self.rows,
                                        g_c.data_ptr(), g_tab.data_ptr() if g_tab is not None else None,
                                        self.ws.data_ptr(), self.ws.numel(), None)
        assert rc == 0


@pytest.mark.parametrize("D,W,skips,L,P,M", [
    (8, 256, (4,), 32, 2, 96),    # DIMO's TimeNet (the one-launch kernels), a small batch
    (8, 256, (4,), 32, 3, 53),    # ragged rows, a latent row used twice
    (8, 256, (4,), 32, 2, 250),   # 500 rows: the weight-gradient tiles walk SEVERAL 64-row stages per K chunk (their
                                  # "previous stage consumed" barrier was untested below ~320 rows: tools/mutate_emulated.py)
    (3, 64, (), 8, 2, 70),        # no skip, narrow: the per-layer GEMM path
    (4, 128, (1,), 0, 1, 33),     # no latent code
    (4, 256, (1,), 0, 2, 40),     # one-launch path: early skip, no latent code (72 embedding columns)
    (3, 256, (), 8, 1, 17),       # one-launch path: no skip layer, a single partial 16-row block
    (2, 256, (0,), 16, 2, 64),    # one-launch path: the skip right after the first layer
])
def test_emulated_timenet_forward_backward_matches_float64_autograd(D, W, skips, L, P, M):
    net = _make(D, W, skips, L, seed=D * 7 + W)
    g = torch.Generator().manual_seed(5)
    c_xyz = (torch.rand(M, 3, generator=g) - 0.5).contiguous()
    T = max(P, 3)
    table = torch.randn(T, max(L, 1), generator=g)[:, :L].contiguous() if L else torch.zeros(T, 0)
    rows = [(2 * p) % T for p in range(P)] if P != 3 else [2, 0, 2]
    times = [0.1 + 0.27 * p for p in range(P)]
    gx, gr = torch.randn(P, M, 3, generator=g), torch.randn(P, M, 4, generator=g)
    dx_ref, dr_ref, gc_ref, gt_ref, gp_ref, gx, gr = _reference(net, c_xyz, times, table, rows, gx, gr)
    net_e = copy.deepcopy(net)
    f = EmulatedTimeNet(net_e)
    tab = table if L else torch.zeros(T, 1)
    dx, dr = f.forward(c_xyz, times, tab, rows)
    _close(dx, dx_ref, "d_xyz")
    _close(dr, dr_ref, "d_rot")
    g_c, g_tab = torch.zeros(M, 3), torch.zeros_like(tab)
    f.backward(gx.float().contiguous(), gr.float().contiguous(), g_c, g_tab if L else None)
    _close(g_c, gc_ref, "g_c_xyz")
    if L:
        _close(g_tab, gt_ref, "g_latent_table")
    for name, p in net_e.named_parameters():
        _close(p.grad, gp_ref[name], name)
    # gradients are ADDED: a second backward doubles them
    f.backward(gx.float().contiguous(), gr.float().contiguous(), g_c, g_tab if L else None)
    _close(g_c, 2 * gc_ref, "g_c_xyz x2")
    _close(net_e.deformnet[0].weight.grad, 2 * gp_ref["deformnet.0.weight"], "deformnet.0.weight x2")


@pytest.mark.parametrize("order", ["reverse", "random:7"])
def test_emulated_timenet_under_other_fiber_schedules(order, monkeypatch):
    """The one-launch kernels run 16 waves per workgroup through a chain of layers with LDS hand-overs: every barrier
    they need must be there when the fibers take turns in reverse or shuffled order."""
    monkeypatch.setenv("SIMT_ORDER", order)
    test_emulated_timenet_forward_backward_matches_float64_autograd(8, 256, (4,), 32, 3, 53)
    test_emulated_timenet_forward_backward_matches_float64_autograd(8, 256, (4,), 32, 2, 250)
