"""LPIPS (VGG-16 variant) as the reference uses it: `lpips.LPIPS(net='vgg')(render, gt).mean()` added to the
per-motion loss with `lambda_lpips` (main_train_dimo.py:33,150,339-341; configs/train_config.yaml:44).

`lpips` is a pip dependency of the reference (requirements.txt:5, UNPINNED; absent from /root/reference and from
this image, as are its pretrained weights).  This module restates the published metric -- Zhang et al., "The
Unreasonable Effectiveness of Deep Features as a Perceptual Metric", CVPR 2018; richzhang/PerceptualSimilarity
`LPIPS(net='vgg', version='0.1', lpips=True, spatial=False)`:

    x -> (x - shift) / scale                                  (ImageNet statistics mapped to inputs in [-1, 1])
      -> VGG-16 features at relu1_2, relu2_2, relu3_3, relu4_3, relu5_3   (64, 128, 256, 512, 512 channels)
    per layer:  unit-normalise over channels (eps 1e-10), squared difference, non-negative 1x1 convolution to one
                channel (the learned "lin" weights), spatial mean;   distance = sum over the five layers  [B,1,1,1]

PARITY UNPINNED: without the package or its weights no value of the reference's can be reproduced here; the tests
pin the computation against an independent functional restatement on random weights.  Supply the real weights with
`load_pretrained` (a torchvision VGG-16 `features.*` state dict and the package's `lin*.model.1.weight` tensors) and
the module computes the published metric.  All convolutions run through PyTorch-ROCm (MIOpen): the metric is off the
per-render path and GEMM-shaped, which is what the vendor library is for.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

# torchvision vgg16().features indices of the convolutions in each of the five slices (each followed by ReLU; a
# 2x2 max-pool opens slices 2-5)
_SLICES = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))
_CHANNELS = (64, 128, 256, 512, 512)
_SHIFT = (-0.030, -0.088, -0.188)
_SCALE = (0.458, 0.448, 0.450)


def normalize_tensor(feat, eps=1e-10):
    """lpips.normalize_tensor: unit length over the channel dimension."""
    return feat / (torch.sqrt(torch.sum(feat ** 2, dim=1, keepdim=True)) + eps)


class LPIPS(nn.Module):
    def __init__(self, net="vgg", pretrained=False):
        super().__init__()
        if net != "vgg":
            raise NotImplementedError("the reference uses net='vgg' (main_train_dimo.py:150)")
        if pretrained:
            raise RuntimeError("no pretrained LPIPS / VGG-16 weights are available offline: construct with "
                               "pretrained=False and call load_pretrained(vgg_features_state, lin_state)")
        self.register_buffer("shift", torch.tensor(_SHIFT)[None, :, None, None])
        self.register_buffer("scale", torch.tensor(_SCALE)[None, :, None, None])
        convs, cin = [], 3
        for cout, idxs in zip(_CHANNELS, _SLICES):
            for _ in idxs:
                convs.append(nn.Conv2d(cin, cout, 3, padding=1))
                cin = cout
        self.convs = nn.ModuleList(convs)
        self.lins = nn.ModuleList([nn.Conv2d(c, 1, 1, bias=False) for c in _CHANNELS])
        with torch.no_grad():  # the learned weights are non-negative (the package clamps them during training)
            for lin in self.lins:
                lin.weight.abs_()
        for p in self.parameters():  # a fixed metric: never trained with the scene
            p.requires_grad_(False)
        self.eval()

    def load_pretrained(self, vgg_features_state, lin_state):
        """vgg_features_state: torchvision vgg16 `features.<i>.weight|bias` (or `<i>.weight|bias`);
        lin_state: the lpips package's `lin<k>.model.1.weight` (or `lins.<k>.model.1.weight`) tensors."""
        flat = [i for idxs in _SLICES for i in idxs]
        with torch.no_grad():
            for conv, i in zip(self.convs, flat):
                for name in ("weight", "bias"):
                    t = vgg_features_state.get(f"features.{i}.{name}", vgg_features_state.get(f"{i}.{name}"))
                    if t is None:
                        raise KeyError(f"VGG-16 features.{i}.{name} missing")
                    getattr(conv, name).copy_(t)
            for k, lin in enumerate(self.lins):
                t = lin_state.get(f"lin{k}.model.1.weight", lin_state.get(f"lins.{k}.model.1.weight"))
                if t is None:
                    raise KeyError(f"lin{k}.model.1.weight missing")
                lin.weight.copy_(t)
        return self

    def features(self, x):
        outs, it = [], iter(self.convs)
        for k, idxs in enumerate(_SLICES):
            if k > 0:
                x = F.max_pool2d(x, 2, 2)
            for _ in idxs:
                x = F.relu(next(it)(x))
            outs.append(x)
        return outs

    def forward(self, in0, in1, normalize=False):
        """[B,3,H,W] x 2 -> [B,1,1,1].  normalize=True maps inputs from [0,1] to [-1,1] first (the reference calls
        it with the default False on [0,1] images, main_train_dimo.py:340 -- kept as is)."""
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        f0 = self.features((in0 - self.shift) / self.scale)
        f1 = self.features((in1 - self.shift) / self.scale)
        val = 0
        for a, b, lin in zip(f0, f1, self.lins):
            d = (normalize_tensor(a) - normalize_tensor(b)) ** 2
            val = val + lin(d).mean(dim=(2, 3), keepdim=True)
        return val
