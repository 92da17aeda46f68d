"""Python face of the native step executor (dimo_amd/csrc/executor.hip, include/dimo_hip.h): persistent
per-render workspaces + two ctypes calls per step (forward of all renders, backward of all renders).

`StepExecutor` owns `max_renders` render slots.  A slot holds everything one render needs between forward and
backward (skinned Gaussians, rasterizer workspaces, per-render gradient buffers), allocated once and re-used every
step -- with 288 GB of HBM per MI355X eight slots at 100k Gaussians / 512^2 cost ~1.6 GB.  GPU only.
"""
import ctypes as C

import torch

from . import _lib

_fp, _vp = C.c_void_p, C.c_void_p


class StepCommon(C.Structure):
    _fields_ = [("N", C.c_int), ("M", C.c_int), ("H", C.c_int), ("W", C.c_int), ("with_normal", C.c_int),
                ("local_frame", C.c_int), ("R_cap", C.c_int64),
                ("xyz", _fp), ("rotation", _fp), ("scaling", _fp), ("opacity", _fp), ("f_dc", _fp),
                ("c_xyz", _fp), ("c_log_radius", _fp), ("nn_dist", _fp), ("nn_idx", _vp), ("bg", _fp),
                ("scale_modifier", C.c_float),
                ("g_xyz", _fp), ("g_rotation", _fp), ("g_scaling", _fp), ("g_opacity", _fp), ("g_f_dc", _fp),
                ("g_c_xyz", _fp), ("g_c_log_radius", _fp),
                ("lbs_scratch", _vp), ("lbs_scratch_bytes", C.c_size_t), ("geom_bytes", C.c_size_t),
                ("bin_bytes", C.c_size_t), ("img_bytes", C.c_size_t), ("bwd_scratch_bytes", C.c_size_t),
                ("stage1", C.c_int), ("log_r", _fp), ("g_log_r", _fp)]


class RenderDesc(C.Structure):
    _fields_ = [("view", _fp), ("proj", _fp), ("campos", _fp), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("d_xyz", _fp), ("d_rot", _fp), ("g_d_xyz", _fp), ("g_d_rot", _fp),
                ("out_color", _fp), ("out_depth", _fp), ("out_normal", _fp), ("out_alpha", _fp),
                ("g_color", _fp), ("g_depth", _fp), ("g_normal", _fp), ("g_alpha", _fp),
                ("pts", _fp), ("rot", _fp), ("scales", _fp), ("opac", _fp), ("radii", _vp),
                ("geom", _vp), ("bin", _vp), ("img", _vp), ("bwd_scratch", _vp),
                ("g_means3D", _fp), ("g_means2D", _fp), ("g_shs", _fp), ("g_opac", _fp), ("g_scales", _fp),
                ("g_rot", _fp), ("g_dot", _fp), ("totals_out", _vp)]


class StepExecutor:
    def __init__(self, N, M, H, W, max_renders, r_cap, device, n_streams=3):
        if torch.device(device).type != "cuda":
            raise RuntimeError("StepExecutor needs a GPU (no CPU fallback in the product path)")
        self.L = _lib.lib()
        self.N, self.M, self.H, self.W, self.max_renders, self.device = N, M, H, W, max_renders, device
        # n_streams: > 0 per-render chains on private streams; 0 batched on the caller's stream; < 0 batched
        # [first, first+count) ranges round-robin over |n_streams| private streams (see include/dimo_hip.h)
        self.batched = n_streams <= 0
        self.ranged = n_streams < 0
        self.handle = self.L.dimo_executor_create(n_streams)
        if not self.handle:
            raise RuntimeError("dimo_executor_create failed")
        f32 = dict(dtype=torch.float32, device=device)
        u8 = dict(dtype=torch.uint8, device=device)
        L = self.L
        self.geom_bytes, self.img_bytes = L.dimo_raster_geom_bytes(N), L.dimo_raster_img_bytes(H, W)
        self.lbs_scratch = torch.empty(L.dimo_deform_backward_scratch_bytes(N, M) * (max_renders if self.batched else 1),
                                       **u8)
        self.slots = []
        for _ in range(max_renders):
            s = dict(pts=torch.empty(N, 3, **f32), rot=torch.empty(N, 4, **f32), scales=torch.empty(N, 3, **f32),
                     opac=torch.empty(N, 1, **f32), radii=torch.empty(N, dtype=torch.int32, device=device),
                     geom=torch.empty(self.geom_bytes, **u8), img=torch.empty(self.img_bytes, **u8),
                     g_means3D=torch.empty(N, 3, **f32), g_means2D=torch.empty(N, 3, **f32),
                     g_shs=torch.empty(N, 1, 3, **f32), g_opac=torch.empty(N, 1, **f32),
                     g_scales=torch.empty(N, 3, **f32), g_rot=torch.empty(N, 4, **f32))
            self.slots.append(s)
        self.r_cap = 0
        self.resize_capacity(r_cap)
        self.common = StepCommon()
        # (R, overflow) of every slot's last render, written by the binning's last kernel (dimo_render_desc.totals_out)
        self.totals = torch.zeros(max_renders, 2, dtype=torch.int32, device=device)
        self.descs = (RenderDesc * max_renders)()
        for d, s in zip(self.descs, self.slots):
            for k in ("pts", "rot", "scales", "opac", "radii", "geom", "img", "bin", "bwd_scratch", "g_means3D",
                      "g_means2D", "g_shs", "g_opac", "g_scales", "g_rot"):
                setattr(d, k, s[k].data_ptr())
        for i, d in enumerate(self.descs):
            d.totals_out = self.totals.data_ptr() + 8 * i

    def destroy(self):
        """Releases the native executor (its events; the private streams belong to a per-device pool).  The caller
        makes sure the device has finished with the slots' buffers (`Trainer._executor` synchronises first)."""
        if getattr(self, "handle", None):
            self.L.dimo_executor_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                torch.cuda.synchronize()
                self.destroy()
        except Exception:
            pass

    def resize_capacity(self, r_cap):
        """(Re)allocates the instance-capacity dependent workspaces of every slot."""
        r_cap = int(r_cap)
        if r_cap == self.r_cap:
            return
        # the host may be a step ahead of the device, whose queued kernels (on private streams the caching allocator
        # knows nothing about) still use the old workspaces
        torch.cuda.synchronize()
        self.r_cap = r_cap
        self.bin_bytes = self.L.dimo_raster_bin_bytes(self.N, r_cap, self.H, self.W)
        self.bwd_bytes = self.L.dimo_raster_backward_scratch_bytes(self.N, r_cap)
        for s in self.slots:
            s["bin"] = torch.empty(self.bin_bytes, dtype=torch.uint8, device=self.device)
            s["bwd_scratch"] = torch.empty(self.bwd_bytes, dtype=torch.uint8, device=self.device)
        if hasattr(self, "descs"):
            for d, s in zip(self.descs, self.slots):
                d.bin, d.bwd_scratch = s["bin"].data_ptr(), s["bwd_scratch"].data_ptr()

    def total_words(self, n):
        """(R, overflow) words of the first n slots: ONE [n, 2] view (for CapacityPolicy / the skip flag)."""
        return self.totals[:n]

    def total_words_range(self, first, n):
        """(R, overflow) words of slots [first, first + n) as ONE [n, 2] view (no gather of the per-slot workspaces)."""
        return self.totals[first:first + n]

    def set_common(self, g, bg, with_normal, local_frame=True, scale_modifier=1.0, stage1=False):
        """stage1: direct deformation (stage s1) -- d_xyz of a render is [N, 3], scales = exp(g._r)."""
        c = self.common
        c.stage1 = int(bool(stage1))
        c.log_r = _lib.ptr(g._r) if stage1 else None
        c.g_log_r = _lib.ptr(g._r.grad) if stage1 else None
        c.N, c.M, c.H, c.W = self.N, self.M, self.H, self.W
        c.with_normal, c.local_frame, c.R_cap = int(with_normal), int(local_frame), self.r_cap
        p = _lib.ptr
        c.xyz, c.rotation, c.scaling, c.opacity, c.f_dc = p(g._xyz), p(g._rotation), p(g._scaling), p(g._opacity), p(g._features_dc)
        c.c_xyz, c.c_log_radius = p(g._c_xyz), p(g._c_radius)
        c.nn_dist = p(g.neighbor_dists) if not stage1 else None
        c.nn_idx = p(g.neighbor_indices) if not stage1 else None
        c.bg = p(bg)
        c.scale_modifier = float(scale_modifier)
        c.g_xyz, c.g_rotation, c.g_scaling, c.g_opacity = p(g._xyz.grad), p(g._rotation.grad), p(g._scaling.grad), p(g._opacity.grad)
        c.g_f_dc, c.g_c_xyz, c.g_c_log_radius = p(g._features_dc.grad), p(g._c_xyz.grad), p(g._c_radius.grad)
        c.lbs_scratch, c.lbs_scratch_bytes = p(self.lbs_scratch), self.lbs_scratch.numel()
        c.geom_bytes, c.bin_bytes, c.img_bytes, c.bwd_scratch_bytes = self.geom_bytes, self.bin_bytes, self.img_bytes, self.bwd_bytes

    def forward(self, n):
        _lib.check(self.L.dimo_executor_forward(self.handle, C.addressof(self.common), n, C.addressof(self.descs),
                                                _lib.current_stream()), "dimo_executor_forward")

    def forward_range(self, first, count):
        """Forward chain of renders [first, first + count) on a private stream (round-robin)."""
        _lib.check(self.L.dimo_executor_forward_range(self.handle, C.addressof(self.common), first, count,
                                                      C.addressof(self.descs), _lib.current_stream()),
                   "dimo_executor_forward_range")

    def join(self, first, count):
        _lib.check(self.L.dimo_executor_join(self.handle, first, count, _lib.current_stream()), "dimo_executor_join")

    def backward_launch(self, first, count):
        _lib.check(self.L.dimo_executor_backward_launch(self.handle, C.addressof(self.common), first, count,
                                                        C.addressof(self.descs), _lib.current_stream()),
                   "dimo_executor_backward_launch")

    def range_stream(self, first):
        """Raw handle of the private stream of the batched range starting at `first` (None if there is none)."""
        return self.L.dimo_executor_range_stream(self.handle, first) or None

    def backward_launch_in_order(self, first, count):
        _lib.check(self.L.dimo_executor_backward_launch_in_order(self.handle, C.addressof(self.common), first, count,
                                                                 C.addressof(self.descs), _lib.current_stream()),
                   "dimo_executor_backward_launch_in_order")

    def side_stream(self, which=0):
        """Raw handle of private stream `which`, waiting for the current stream's tail: the caller launches side work
        on it (then `side_done`) while it goes on enqueueing on the current stream."""
        s = self.L.dimo_executor_side_stream(self.handle, which, _lib.current_stream())
        if not s:
            raise RuntimeError("dimo_executor_side_stream failed")
        return s

    def side_done(self, which=0):
        _lib.check(self.L.dimo_executor_side_done(self.handle, which), "dimo_executor_side_done")

    def wait_side(self, which=0):
        _lib.check(self.L.dimo_executor_wait_side(self.handle, which, _lib.current_stream()), "dimo_executor_wait_side")

    def private_stream(self, which=0):
        return self.L.dimo_executor_private_stream(self.handle, which) or None

    def join_ranges(self, first, count, stream=None):
        _lib.check(self.L.dimo_executor_join_ranges(self.handle, first, count,
                                                    stream if stream is not None else _lib.current_stream()),
                   "dimo_executor_join_ranges")

    def backward_skinning_in_order(self, first, count):
        """The range's skinning backward behind its rasterizer backward, on the same stream (nothing shared is
        written); `backward_accumulate` over the step's renders then only folds."""
        _lib.check(self.L.dimo_executor_backward_skinning_in_order(self.handle, C.addressof(self.common), first, count,
                                                                   C.addressof(self.descs), _lib.current_stream()),
                   "dimo_executor_backward_skinning_in_order")

    def backward_launch_joint(self, first, count):
        _lib.check(self.L.dimo_executor_backward_launch_joint(self.handle, C.addressof(self.common), first, count,
                                                              C.addressof(self.descs), _lib.current_stream()),
                   "dimo_executor_backward_launch_joint")

    def backward_accumulate(self, first, count, stream=None):
        """`stream`: raw handle of the stream to run on (default: the current one)."""
        _lib.check(self.L.dimo_executor_backward_accumulate(self.handle, C.addressof(self.common), first, count,
                                                            C.addressof(self.descs),
                                                            stream if stream is not None else _lib.current_stream()),
                   "dimo_executor_backward_accumulate")
