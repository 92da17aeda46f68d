"""LPIPS (VGG-16 variant, dimo_amd/lpips_vgg.py) against an independent functional restatement of the published
metric on random weights (the package and its weights are absent: parity with the reference's values is unpinned),
plus the properties any weights satisfy."""
import torch
import torch.nn.functional as F

from dimo_amd.lpips_vgg import LPIPS, _SLICES


def _functional(model, a, b):
    shift = torch.tensor([-0.030, -0.088, -0.188])[None, :, None, None]
    scale = torch.tensor([0.458, 0.448, 0.450])[None, :, None, None]
    ws = [(c.weight, c.bias) for c in model.convs]

    def feats(x):
        x = (x - shift) / scale
        out, k = [], 0
        for s, idxs in enumerate(_SLICES):
            if s:
                x = F.max_pool2d(x, kernel_size=2, stride=2)
            for _ in idxs:
                x = torch.relu(F.conv2d(x, ws[k][0], ws[k][1], padding=1))
                k += 1
            out.append(x)
        return out

    total = torch.zeros(a.shape[0], 1, 1, 1)
    for fa, fb, lin in zip(feats(a), feats(b), model.lins):
        na = fa / (fa.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = fb / (fb.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        total = total + ((na - nb) ** 2 * lin.weight.reshape(1, -1, 1, 1)).sum(1, keepdim=True).mean((2, 3), keepdim=True)
    return total


def test_matches_functional_restatement_and_properties():
    torch.manual_seed(0)
    m = LPIPS()
    assert [c.out_channels for c in m.convs] == [64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512]
    assert all(not p.requires_grad for p in m.parameters())
    a, b = torch.rand(2, 3, 32, 48), torch.rand(2, 3, 32, 48)
    d = m(a, b)
    assert d.shape == (2, 1, 1, 1)
    assert torch.allclose(d, _functional(m, a, b), rtol=1e-5, atol=1e-7)
    assert torch.all(d >= 0) and torch.all(m(a, a) == 0)              # non-negative lin weights; identity
    assert torch.allclose(m(a, b), m(b, a), rtol=1e-6)                # symmetric
    assert torch.allclose(m(a, b, normalize=True), m(2 * a - 1, 2 * b - 1), rtol=1e-6)
    x = a.clone().requires_grad_(True)
    m(x, b).mean().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0


def test_load_pretrained_takes_torchvision_and_package_key_names():
    torch.manual_seed(1)
    src, dst = LPIPS(), LPIPS()
    flat = [i for idxs in _SLICES for i in idxs]
    vgg = {}
    for conv, i in zip(src.convs, flat):
        vgg[f"features.{i}.weight"], vgg[f"features.{i}.bias"] = conv.weight.clone(), conv.bias.clone()
    lin = {f"lin{k}.model.1.weight": l.weight.clone() for k, l in enumerate(src.lins)}
    dst.load_pretrained(vgg, lin)
    a, b = torch.rand(1, 3, 16, 16), torch.rand(1, 3, 16, 16)
    assert torch.equal(src(a, b), dst(a, b))
