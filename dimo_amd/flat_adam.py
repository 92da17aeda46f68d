"""`FlatAdam`: torch.optim.Adam semantics (renderer/latent_gs_renderer.py:460-476: 12 named groups, lr set per
step by `update_learning_rate`, eps 1e-15) executed as ONE HIP launch over the flat parameter / gradient buckets
(`dimo_flat_adam_step`, dimo_amd/csrc/adam.hip) instead of one multi-tensor launch per group.

It keeps the part of the optimizer interface the training path uses: `param_groups` (list of dicts with "name",
"lr", "params"), `step()`, `zero_grad()`, `defaults`, `state_dict()` / `load_state_dict()` of the moment buffers.
GPU only; the CPU/test path keeps torch.optim.Adam.
"""
import ctypes as C

import torch

from . import _lib


class FlatAdam:
    def __init__(self, groups, flat_params, flat_grads, betas=(0.9, 0.999), eps=1e-15):
        if not flat_params.is_cuda:
            raise RuntimeError("FlatAdam needs the flat buckets on the GPU (no CPU fallback in the product path)")
        self.param_groups = groups
        self.defaults = dict(lr=0.0, betas=betas, eps=eps)
        self.flat_params, self.flat_grads = flat_params, flat_grads
        self.exp_avg = torch.zeros_like(flat_params)
        self.exp_avg_sq = torch.zeros_like(flat_params)
        self.pre_step = None  # callable run before every step (GaussianModel: renders queued behind render() run first)
        self.finals = 0    # launches that finished a step (whole bucket, or the "tail" part): they clear zero_extra
        self.launches = 0  # every step() call, applied or skipped on the device
        self.skipped_host = 0  # skipped launches the host knows of (Trainer: read back with a lag of one step)
        # the device's own count of skipped launches (two words written alternately, see adam.hip): the bias
        # corrections use launches - skipped without the host ever reading the overflow flag
        self._skipped = torch.zeros(2, dtype=torch.int32, device=flat_params.device)
        # groups must tile the bucket in order, at most 3 padding floats (16-byte alignment) between tensors
        # (GaussianModel.flatten_parameters builds it that way); padding is updated with the group it precedes
        base = flat_params.data_ptr()
        ends, o = [], 0
        for g in groups:
            for p in g["params"]:
                if p.numel() == 0:
                    continue
                at = (p.data_ptr() - base) // 4
                assert 0 <= at - o < 4 and (p.data_ptr() - base) % 4 == 0, \
                    "parameter is not a view of the flat bucket in group order"
                o = at + p.numel()
            ends.append(o)
        assert 0 <= flat_params.numel() - o < 4
        if ends:
            ends[-1] = flat_params.numel()
        self._ends = (C.c_int64 * len(ends))(*ends)
        self._n_seg = len(ends)

    @property
    def step_count(self):
        """Updates actually applied, as far as the host knows (== torch.optim.Adam's `step` state)."""
        return self.launches - self.skipped_host

    @step_count.setter
    def step_count(self, value):
        self.launches = int(value)
        self.skipped_host = 0
        self._skipped.zero_()

    def step(self, skip_flags=None, zero_grad=False, report=None, zero_extra=None, part=None, stream=None):
        """skip_flags: optional int32 tensor [k, 2] (rasterizer `total` words: R, overflow) -- any non-zero
        overflow word turns this step into a no-op on the device.  report: optional (src int32 device tensor, dst
        PINNED int32 host tensor with >= 1 + src.numel() words, seq) -- the launch copies src to dst[1:] and then stores
        seq to dst[0] (CapacityPolicy polls it a step later: no copy engine, no event).  zero_extra: optional fp32
        device tensor cleared by the same launch.  part: None = the whole bucket; ("head", split) = only [0, split)
        -- the FIRST of the two launches of a step; ("tail", split) = [split, n), the second and final one (it counts
        the step, reports, clears): the per-Gaussian head of the bucket can be updated as soon as ITS gradients are
        final, under the TimeNet backward.  Both launches must see the same skip_flags.  stream: raw stream handle
        (default: torch's current stream)."""
        if part is None or part[0] == "head":
            if self.pre_step is not None:
                self.pre_step()
            self.launches += 1
        begin, end, final = 0, 0, 1
        if part is None or part[0] == "tail":
            self.finals += 1
        if part is not None:
            kind, split = part
            split = int(split)
            assert split % 4 == 0 and 0 < split < self.flat_params.numel()
            begin, end, final = (0, split, 0) if kind == "head" else (split, self.flat_params.numel(), 1)
        lrs = (C.c_float * self._n_seg)(*[float(g["lr"]) for g in self.param_groups])
        b1, b2 = self.defaults["betas"]
        if skip_flags is not None and skip_flags.numel() > 0:
            assert skip_flags.dtype == torch.int32 and skip_flags.is_contiguous()
            if skip_flags.dim() == 2 and skip_flags.shape[-1] == 2:  # rasterizer `total` words (R, overflow)
                fptr, nfl, fstride = skip_flags.data_ptr() + 4, skip_flags.shape[0], 2
            else:  # plain flag words
                fptr, nfl, fstride = skip_flags.data_ptr(), skip_flags.numel(), 1
        else:
            fptr, nfl, fstride = None, 0, 1
        if report is not None:
            src, dst, seq = report
            # (dst must be PINNED host memory -- not checked here: is_pinned() is a driver query per call)
            assert src.dtype == torch.int32 and src.is_contiguous() and src.is_cuda
            assert dst.dtype == torch.int32 and not dst.is_cuda and dst.numel() >= 1 + src.numel()
            rsrc, rwords, rdst, rseq = src.data_ptr(), src.numel(), dst.data_ptr(), int(seq) & 0xffffffff
        else:
            rsrc, rwords, rdst, rseq = None, 0, None, 0
        if zero_extra is not None:
            assert zero_extra.dtype == torch.float32 and zero_extra.is_contiguous() and zero_extra.is_cuda
            zptr, zn = zero_extra.data_ptr(), zero_extra.numel()
        else:
            zptr, zn = None, 0
        _lib.check(_lib.lib().dimo_flat_adam_step(
            self.flat_params.numel(), _lib.ptr(self.flat_params), _lib.ptr(self.flat_grads), _lib.ptr(self.exp_avg),
            _lib.ptr(self.exp_avg_sq), self._n_seg, self._ends, lrs, b1, b2, self.defaults["eps"], self.launches,
            fptr, nfl, fstride, int(bool(zero_grad)), _lib.ptr(self._skipped), rsrc, rwords, rdst, rseq, zptr, zn,
            begin, end, final, stream if stream is not None else _lib.current_stream()),
            "dimo_flat_adam_step")

    def zero_grad(self, set_to_none=False):
        self.flat_grads.zero_()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lrs": [g["lr"] for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
