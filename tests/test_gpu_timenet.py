"""dimo_timenet_forward / dimo_timenet_backward (fp32 MFMA GEMM chains) against the TimeNet module evaluated in
float64 on the CPU with autograd.  The module itself is pinned to the reference's TimeNet by the golden fixtures
(tests/test_host_golden.py); tolerance 1e-4 of the largest reference magnitude (north_star's float bar)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(D, W, skips, latent_dim, seed):
    from dimo_amd.deform import TimeNet
    torch.manual_seed(seed)
    net = TimeNet(D=D, W=W, skips=skips, latent_code_dim=latent_dim)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # the zero-initialised head outputs would hide most of the backward
        for lin in (net.pts_layers[2], net.rot_layers[2]):
            lin.weight.copy_(0.1 * torch.randn(lin.weight.shape, generator=g))
            lin.bias.copy_(0.1 * torch.randn(lin.bias.shape, generator=g))
        for lin in list(net.deformnet) + [net.pts_layers[0], net.rot_layers[0]]:
            lin.bias.copy_(0.05 * torch.randn(lin.bias.shape, generator=g))
    return net


def _robust_rows(n64, c, t, lat, margin=2e-5):
    """Rows (pair, control point) none of whose ReLU inputs lies within `margin` of the kink: a pre-activation of
    +1e-7 in fp32 and -1e-7 in fp64 flips a whole gradient path, which is round-off, not a kernel property.  The
    other rows get a zero output gradient in BOTH implementations (about 5 % of the rows)."""
    smallest = []
    hooks = [lin.register_forward_hook(lambda _m, _i, out: smallest.append(out.detach().abs().amin(-1)))
             for lin in list(n64.deformnet) + [n64.pts_layers[0], n64.rot_layers[0]]]
    with torch.no_grad():
        n64(c[None], t, lat, t_apply=True)
    for h in hooks:
        h.remove()
    return torch.stack(smallest).amin(0) > margin  # [P, M]


def _reference(net, c_xyz, times, table, rows, gx, gr):
    """Returns the float64 results and the (row-masked) output gradients to feed the kernels with."""
    n64 = copy.deepcopy(net).double()
    c = c_xyz.double().requires_grad_(True)
    tab = table.double().requires_grad_(True)
    P, M = len(times), c.shape[0]
    t = torch.tensor(times, dtype=torch.float64)[:, None, None].expand(P, M, 1)
    lat = tab[torch.tensor(rows)][:, None, :].expand(P, M, -1)
    keep = _robust_rows(n64, c.detach(), t, lat.detach())
    assert float(keep.float().mean()) > 0.7
    gx, gr = gx * keep[..., None], gr * keep[..., None]
    dx, dr = n64(c[None], t, lat, t_apply=True)
    ((dx * gx.double()).sum() + (dr * gr.double()).sum()).backward()
    grads = {k: p.grad for k, p in n64.named_parameters()}
    return dx.detach(), dr.detach(), c.grad, tab.grad, grads, gx.contiguous(), gr.contiguous()


def _close(got, ref, name, tol=1e-4):
    ref = ref.double()
    err = float((got.double().cpu() - ref).abs().max())
    scale = max(float(ref.abs().max()), 1e-6)
    assert err <= tol * scale, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("D,W,skips,L,P,M", [
    (8, 256, (4,), 32, 4, 512),   # DIMO's TimeNet at the benchmark's batch
    (8, 256, (4,), 32, 3, 203),   # ragged rows, a latent row used twice
    (3, 64, (), 8, 2, 70),        # no skip, narrow
    (4, 128, (1,), 0, 1, 33),     # no latent code
    (4, 256, (1,), 0, 2, 40),     # one-launch path: early skip, no latent code (72 embedding columns)
    (3, 256, (), 8, 1, 17),       # one-launch path: no skip layer, a single partial 16-row block
    (2, 256, (0,), 16, 2, 64),    # one-launch path: the skip right after the first layer
])
def test_timenet_forward_backward_matches_float64_autograd(D, W, skips, L, P, M):
    from dimo_amd.fused_timenet import FusedTimeNet
    net = _make(D, W, skips, L, seed=D * 7 + W)
    g = torch.Generator().manual_seed(5)
    c_xyz = torch.rand(M, 3, generator=g) - 0.5
    T = max(P, 3)
    table = torch.randn(T, max(L, 1), generator=g)[:, :L].contiguous() if L else torch.zeros(T, 0)
    rows = [(2 * p) % T for p in range(P)] if P != 3 else [2, 0, 2]
    times = [0.1 + 0.27 * p for p in range(P)]
    gx, gr = torch.randn(P, M, 3, generator=g), torch.randn(P, M, 4, generator=g)
    dx_ref, dr_ref, gc_ref, gt_ref, gp_ref, gx, gr = _reference(net, c_xyz, times, table, rows, gx, gr)

    dev = torch.device("cuda")
    net_d = copy.deepcopy(net).to(dev)
    f = FusedTimeNet(net_d)
    tab_d = table.to(dev) if L else torch.zeros(T, 1, device=dev)
    dx, dr = f.forward(c_xyz.to(dev), times, tab_d, rows)
    _close(dx, dx_ref, "d_xyz")
    _close(dr, dr_ref, "d_rot")
    g_c = torch.zeros(M, 3, device=dev)
    g_tab = torch.zeros_like(tab_d)
    f.backward(gx.to(dev), gr.to(dev), g_c, g_tab if L else None)
    torch.cuda.synchronize()
    _close(g_c, gc_ref, "g_c_xyz")
    if L:
        _close(g_tab, gt_ref, "g_latent_table")
    for name, p in net_d.named_parameters():
        _close(p.grad, gp_ref[name], name)
    # gradients are ADDED: a second backward doubles them
    f.backward(gx.to(dev), gr.to(dev), g_c, g_tab if L else None)
    _close(g_c, 2 * gc_ref, "g_c_xyz x2")
    _close(net_d.deformnet[0].weight.grad, 2 * gp_ref["deformnet.0.weight"], "deformnet.0.weight x2")


def test_timenet_matches_eager_module_on_device():
    """Same numbers as the PyTorch module the autograd path uses (rocBLAS GEMMs), to fp32 round-off."""
    from dimo_amd.fused_timenet import FusedTimeNet
    dev = torch.device("cuda")
    net = _make(8, 256, (4,), 32, seed=3).to(dev)
    g = torch.Generator().manual_seed(11)
    M, P = 512, 5
    c_xyz = (torch.rand(M, 3, generator=g) - 0.5).to(dev)
    table = torch.randn(7, 32, generator=g).to(dev)
    rows, times = [6, 1, 1, 0, 3], [0.0, 0.25, 0.5, 0.75, 1.0]
    t = torch.tensor(times, device=dev)[:, None, None].expand(P, M, 1)
    lat = table[torch.tensor(rows, device=dev)][:, None, :].expand(P, M, -1)
    with torch.no_grad():
        ex, er = net(c_xyz[None], t, lat, t_apply=True)
    dx, dr = FusedTimeNet(net).forward(c_xyz, times, table, rows)
    assert float((dx - ex).abs().max()) <= 2e-5 * max(1.0, float(ex.abs().max()))
    assert float((dr - er).abs().max()) <= 2e-5 * max(1.0, float(er.abs().max()))


def test_timenet_argument_errors():
    from dimo_amd.fused_timenet import FusedTimeNet
    dev = torch.device("cuda")
    net = _make(8, 256, (4,), 32, seed=3).to(dev)
    f = FusedTimeNet(net)
    c = torch.zeros(4, 3, device=dev)
    with pytest.raises(ValueError):
        f.forward(c, [0.0] * 300, torch.zeros(300, 32, device=dev))  # > DIMO_TIMENET_MAX_PAIRS
    with pytest.raises(ValueError):
        f.forward(c, [0.0, 1.0], torch.zeros(2, 32, device=dev), [0, 5])  # latent row out of range
    with pytest.raises(RuntimeError):
        FusedTimeNet(net).backward(torch.zeros(1, 4, 3, device=dev), torch.zeros(1, 4, 4, device=dev))
    dx, dr = f.forward(c, [], torch.zeros(1, 32, device=dev), [])  # empty batch
    assert dx.shape == (0, 4, 3) and dr.shape == (0, 4, 4)
