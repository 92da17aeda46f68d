"""TimeNet forward / backward on the HIP library (dimo_amd/csrc/timenet.hip, include/dimo_hip.h) for a whole
step's batch of (motion, frame) pairs: two ctypes calls instead of ~190 eager launches.

`FusedTimeNet` wraps a `dimo_amd.deform.TimeNet` module (the checkpoint-compatible mirror of
renderer/latent_gs_renderer.py:184-245): it reads the module's parameters in place and ADDS the parameter
gradients to their `.grad` tensors (views into the flat gradient bucket during training).  GPU only.
"""
import ctypes as C

import torch

from . import _lib

MAX_LAYERS, MAX_PAIRS = 20, 256


class TimeNetDesc(C.Structure):
    _fields_ = [("D", C.c_int), ("W", C.c_int), ("skip", C.c_int), ("pts_freqs", C.c_int), ("time_freqs", C.c_int),
                ("latent_dim", C.c_int),
                ("weight", C.c_void_p * MAX_LAYERS), ("bias", C.c_void_p * MAX_LAYERS),
                ("g_weight", C.c_void_p * MAX_LAYERS), ("g_bias", C.c_void_p * MAX_LAYERS)]


def linear_layers(net):
    """The module's Linear layers in the order of dimo_timenet_desc.weight[]."""
    return list(net.deformnet) + [net.pts_layers[0], net.pts_layers[2], net.rot_layers[0], net.rot_layers[2]]


class FusedTimeNet:
    def __init__(self, net):
        p = next(net.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("FusedTimeNet needs a GPU (no CPU fallback in the product path)")
        if len(net.skips) > 1:
            raise ValueError("one skip connection supported (the reference uses skips=[4])")
        self.net, self.device, self.L = net, p.device, _lib.lib()
        self.layers = linear_layers(net)
        if len(self.layers) > MAX_LAYERS:
            raise ValueError("too many layers")
        self.desc = TimeNetDesc()
        d = self.desc
        d.D, d.W = len(net.deformnet), net.deformnet[0].out_features
        d.skip = net.skips[0] if net.skips else -1
        d.pts_freqs, d.time_freqs = net.pts_ch, net.times_ch
        d.latent_dim = net.input_ch - 6 * net.pts_ch - 2 * net.times_ch
        self._key = None
        self._ws, self._shape, self._times, self._rows = None, None, None, None

    def _refresh(self, need_grads):
        """Re-reads the parameter / gradient pointers only when they moved (re-flattened buckets, fresh .grad
        tensors): in steady state this is two data_ptr() calls per call, not a walk over twenty layers."""
        l0, ll = self.layers[0], self.layers[-1]
        if need_grads and (l0.weight.grad is None or ll.bias.grad is None):
            for lin in self.layers:
                for t in (lin.weight, lin.bias):
                    if t.grad is None:
                        t.grad = torch.zeros_like(t)
        key = (l0.weight.data_ptr(), ll.bias.data_ptr(),
               l0.weight.grad.data_ptr() if l0.weight.grad is not None else 0,
               ll.bias.grad.data_ptr() if ll.bias.grad is not None else 0)
        if key == self._key:
            return
        d = self.desc
        for i, lin in enumerate(self.layers):
            for t in (lin.weight, lin.bias):
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise ValueError("TimeNet parameters must be contiguous fp32")
            d.weight[i], d.bias[i] = lin.weight.data_ptr(), lin.bias.data_ptr()
            d.g_weight[i] = lin.weight.grad.data_ptr() if lin.weight.grad is not None else None
            d.g_bias[i] = lin.bias.grad.data_ptr() if lin.bias.grad is not None else None
        self._key = key

    def forward(self, c_xyz, times, latent_table, latent_rows=None):
        """c_xyz [M,3]; times: P python floats; latent_table [T,L]; latent_rows: P row indices (None: row p).
        Returns d_xyz [P,M,3], d_rot [P,M,4] (no autograd graph: call `backward` with their gradients)."""
        self._refresh(False)
        P, M = len(times), c_xyz.shape[0]
        if P > MAX_PAIRS:
            raise ValueError(f"at most {MAX_PAIRS} (motion, frame) pairs per call")
        c_xyz, latent_table = c_xyz.detach(), latent_table.detach()
        if not (c_xyz.is_contiguous() and latent_table.is_contiguous()):
            raise ValueError("contiguous inputs expected")
        if latent_rows is not None and (len(latent_rows) != P or max(latent_rows, default=0) >= latent_table.shape[0]):
            raise ValueError("latent_rows out of range")
        if latent_rows is None and latent_table.shape[0] < P:
            raise ValueError("latent_table needs one row per pair")
        nbytes = self.L.dimo_timenet_workspace_bytes(C.byref(self.desc), P, M)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._times = (C.c_float * max(P, 1))(*[float(t) for t in times])
        self._rows = (C.c_int * max(P, 1))(*[int(r) for r in latent_rows]) if latent_rows is not None else None
        self._shape = (P, M)
        d_xyz = torch.empty(P, M, 3, dtype=torch.float32, device=self.device)
        d_rot = torch.empty(P, M, 4, dtype=torch.float32, device=self.device)
        _lib.check(self.L.dimo_timenet_forward(C.byref(self.desc), P, M, _lib.ptr(c_xyz), self._times,
                                               _lib.ptr(latent_table), self._rows, _lib.ptr(d_xyz), _lib.ptr(d_rot),
                                               _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()),
                   "dimo_timenet_forward")
        return d_xyz, d_rot

    def backward(self, g_d_xyz, g_d_rot, g_c_xyz=None, g_latent_table=None):
        """Adds the gradients of the last `forward` to the parameters' .grad, g_c_xyz [M,3] and g_latent_table."""
        if self._shape is None:
            raise RuntimeError("backward without forward")
        self._refresh(True)
        P, M = self._shape
        for t, shp in ((g_d_xyz, (P, M, 3)), (g_d_rot, (P, M, 4))):
            if tuple(t.shape) != shp or not t.is_contiguous() or t.dtype != torch.float32:
                raise ValueError("gradient shape / layout mismatch")
        _lib.check(self.L.dimo_timenet_backward(C.byref(self.desc), P, M, _lib.ptr(g_d_xyz), _lib.ptr(g_d_rot),
                                                self._times, self._rows, _lib.ptr(g_c_xyz), _lib.ptr(g_latent_table),
                                                _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()),
                   "dimo_timenet_backward")
