"""`LazyTensor` (dimo_amd/batched_render.py): the stand-ins `Renderer.render` returns for a queued render's outputs on
the GPU (the call surface is renderer/latent_gs_renderer.py:1283-1293's dict).  The host logic is device independent:
here the producer is a CPU thunk and a counter tells when it ran."""
import torch

from dimo_amd.batched_render import LazyTensor, materialize
from dimo_amd.rasterizer import CapacityPolicy


def _lazy(value):
    calls = []

    def fn():
        calls.append(1)
        return value

    return LazyTensor(fn), calls


def test_nothing_runs_until_a_use_and_every_kind_of_use_runs_it_once():
    base = torch.arange(12.0).reshape(3, 4)
    uses = [
        lambda x: torch.stack([x, x]),            # torch function with the stand-in inside a list
        lambda x: torch.nn.functional.relu(x),     # torch.nn.functional
        lambda x: x.clamp(0, 1),                   # method
        lambda x: x * 2 + 1,                       # operators
        lambda x: 2 - x,                           # reflected operator
        lambda x: x > 3,                           # comparison (radii > 0)
        lambda x: x[1:, :2],                       # indexing
        lambda x: x.shape,                         # attribute
        lambda x: float(x.sum()),                  # the reference's per-render .item() style reads
        lambda x: torch.cat((x, base), dim=0),     # mixed with real tensors in a tuple
        lambda x: torch.where(x > 5, x, torch.zeros_like(x)),
        lambda x: len(x),
    ]
    for use in uses:
        lz, calls = _lazy(base.clone())
        assert not calls and "pending" in repr(lz)
        got, want = use(lz), use(base.clone())
        assert len(calls) == 1
        if isinstance(want, torch.Tensor):
            assert torch.equal(got, want)
        else:
            assert got == want
        use(lz)
        assert len(calls) == 1, "materialised twice"
        assert materialize(lz) is materialize(lz) and isinstance(materialize(lz), torch.Tensor)
    assert materialize(base) is base


def test_gradients_flow_through_a_stand_in():
    w = torch.randn(5, requires_grad=True)
    lz, calls = _lazy(w * 3.0)
    loss = (torch.stack([lz, lz]) ** 2).sum() + lz.sum()
    loss.backward()
    assert len(calls) == 1
    assert torch.allclose(w.grad, 2 * 2 * 9 * w.detach() + 3.0)


def test_kwargs_and_nested_containers_are_unwrapped():
    a, ca = _lazy(torch.ones(2, 3))
    b, cb = _lazy(torch.zeros(2, 3))
    out = torch.cat(tensors=[a, b], dim=1)
    assert out.shape == (2, 6) and ca == [1] and cb == [1]
    assert torch.equal(torch.add(a, other=b), torch.ones(2, 3))


def test_capacity_policy_takes_single_renders_and_batches():
    pol = CapacityPolicy(initial=1000, margin=1.5)
    pol.track(torch.tensor([400, 0], dtype=torch.int32))                       # one render's (R, overflow)
    pol.track(torch.tensor([[500, 0], [900, 0]], dtype=torch.int32))           # a batch of two
    assert pol.check() and pol.last_r_max == 900 and pol.capacity >= 900 * 1.5
    pol.track(torch.tensor([[100, 1]], dtype=torch.int32))                     # overflow flag set
    before = pol.capacity
    assert not pol.check() and pol.capacity >= before
    assert pol.check()  # nothing pending
