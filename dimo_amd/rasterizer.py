"""Differentiable tile rasterizer on libdimo_hip -- the drop-in for the two CUDA
extensions DIMO imports:

  diff_gauss.GaussianRasterizer                   (6 outputs, default training path)
      settings renderer/latent_gs_renderer.py:1132-1147, call :1255-1266
  diff_gaussian_rasterization.GaussianRasterizer  (4 outputs)
      settings renderer/latent_gs_renderer.py:1149-1163, call :1268-1277

`dimo_amd.diff_gauss` / `dimo_amd.diff_gaussian_rasterization` re-export the
classes below under the reference's module names.

Everything runs on the caller's current stream through the C ABI (include/dimo_hip.h);
workspaces come from torch's caching allocator and live in the autograd ctx until
backward.  There is no CPU fallback: without the HIP library / a GPU tensor this raises.

Two layers:
  * `raster_forward` / `raster_backward` -- plain functions on tensors (no autograd), used by the
    trainer's direct pipeline, which chains the HIP kernels itself;
  * `_Rasterize` (autograd.Function) + the two `GaussianRasterizer` modules -- the reference's call surface.

Instance capacity.  The number of (Gaussian, tile) instances R is only known on the
device after projection.  `capacity=None` (default) reads it back (one stream sync per
render, like the CUDA original).  A `CapacityPolicy` instead sizes the sort buffers from
a running bound so the whole render is enqueued without a host round trip; an overflow
is flagged on the device and surfaced by `CapacityPolicy.check()`.
"""
import ctypes as C
import time
from typing import NamedTuple, Optional

import torch

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class CapacityPolicy:
    """Sync-free sizing of the instance buffers: capacity = margin * max(R seen), grown on overflow."""

    def __init__(self, initial=1 << 20, margin=1.5):
        self.capacity = int(initial)
        self.margin = float(margin)
        self._pending = []  # geom `total` views (R, overflow) of renders since the last check
        self.last_r_mean = self.last_r_max = None

    def next_capacity(self):
        return self.capacity

    def track(self, total_view):
        """total_view: the (R, overflow) words of ONE render ([2] or longer: the first two words count) or of several
        ([n, 2], e.g. StepExecutor.total_words)."""
        self._pending.append(total_view[:2][None] if total_view.dim() == 1 else total_view)

    def _update(self, tot):
        r_max = int(tot[:, 0].max())
        self.last_r_mean, self.last_r_max = float(tot[:, 0].float().mean()), r_max
        ok = int(tot[:, 1].max()) == 0
        need = int(r_max * self.margin) + 1024
        if need > self.capacity or not ok:
            # grown with headroom and in whole 2^20 steps: every change of the capacity re-allocates every render
            # slot's workspaces behind a device sync, and a scene whose instance count creeps up set a new record
            # -- and paid that -- on nearly every step (2.8 ms instead of 1.4 ms per step, host bound)
            self.capacity = max(self.capacity, -(-int(need * 1.25) // (1 << 20)) * (1 << 20))
        return ok

    def check(self):
        """One host sync for all renders since the last call. Returns True if every render fitted;
        on overflow the capacity is raised and the caller must redo the step."""
        if not self._pending:
            return True
        tot = (self._pending[0] if len(self._pending) == 1 else torch.cat(self._pending)).cpu()
        self._pending = []
        return self._update(tot)

    def collect_async(self, defer_copy=False):
        """Sync-free variant: returns the stacked device words [k, 2] (R, overflow) of the renders since the last
        call (for device-side consumers, e.g. the optimizer's skip flag) and starts a pinned-memory copy that
        `poll()` evaluates later.  defer_copy: the copy is started by start_copy() instead -- the trainer enqueues it
        BEHIND the optimizer's launch (a 4 us copy and its gap sat in front of Adam on the step's critical path)."""
        if not self._pending:
            return None
        tot = self._pending[0] if len(self._pending) == 1 else torch.cat(self._pending)
        self._pending = []
        if defer_copy:
            self._deferred = tot
            return tot
        self._start_copy(tot)
        return tot

    RING, RING_WORDS = 8, 1 + 2 * 256

    def collect_report(self):
        """Like `collect_async`, but the words reach the host THROUGH a kernel: returns (device words [k, 2], pinned
        host slot, sequence number) for `FlatAdam.step(report=...)`, whose launch copies the words into the slot and
        stores the sequence number behind them; `poll()` reads the slot a step later.  No copy engine, no event: the
        4 us device-to-host copy and the ~13 us gap behind it sat between one step's optimizer and the next step's first
        kernel.  None if nothing is pending (or the words do not fit a slot: use `collect_async`)."""
        if not self._pending:
            return None
        n = sum(t.shape[0] for t in self._pending)
        if 1 + 2 * n > self.RING_WORDS:
            return None
        tot = self._pending[0] if len(self._pending) == 1 else torch.cat(self._pending)
        self._pending = []
        if getattr(self, "_ring", None) is None:
            self._ring = torch.zeros(self.RING, self.RING_WORDS, dtype=torch.int32, pin_memory=True)
            self._slots = [self._ring[i] for i in range(self.RING)]
            self._slot_np = [s.numpy() for s in self._slots]  # (the host's view: plain memory reads in poll)
            self._seq = 0
        self._seq += 1
        slot = self._slots[self._seq % self.RING]
        self._inflight = getattr(self, "_inflight", [])
        self._inflight.append((slot, None, self._seq, int(tot.shape[0])))
        return tot.contiguous(), slot, self._seq

    def start_copy(self):
        tot, self._deferred = getattr(self, "_deferred", None), None
        if tot is not None:
            self._start_copy(tot)

    def _start_copy(self, tot):
        host = torch.empty(tot.shape, dtype=tot.dtype, pin_memory=True)
        host.copy_(tot, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._inflight = getattr(self, "_inflight", [])
        self._inflight.append((host, ev))

    def poll(self, lag=1):
        """Evaluates the asynchronous copies except the `lag` newest ones.  Returns the number of steps that had
        overflowed.  Called at the start of a step, lag = 1 leaves the copy of the step just enqueued alone and looks
        at the one before it, which finished a whole step ago -- the host never waits for the device and can run a
        full step ahead (with lag = 0 the call is a device sync in disguise: the event of the previous step's
        renders completes only when that step is all but done).  The device-side skip flag, not this read-back, is
        what keeps an overflowing step from being applied."""
        inflight = getattr(self, "_inflight", [])
        ready, self._inflight = (inflight[:-lag], inflight[-lag:]) if lag > 0 else (inflight, [])
        bad = 0
        for entry in ready:
            host, ev = entry[0], entry[1]
            if ev is None:  # a slot the optimizer's launch fills (collect_report): complete when its number is there
                seq, n = entry[2], entry[3]
                want = seq & 0xffffffff
                view = self._slot_np[seq % self.RING]
                # (normally there: the launch ran a step ago.  When the host is further ahead, wait for THAT launch
                # only -- like an event wait; a device-wide sync would drain the queue the host has built up)
                lost, deadline = False, None
                while (int(view[0]) & 0xffffffff) != want:
                    now = time.perf_counter()
                    if deadline is None:
                        deadline = now + 0.25
                    elif now > deadline:  # the launch that carries the report never ran (an optimizer step somebody
                        torch.cuda.synchronize()  # replaced or skipped): the device-side skip flag still guards the
                        lost = (int(view[0]) & 0xffffffff) != want  # update
                        break
                if lost:
                    self.lost_reports = getattr(self, "lost_reports", 0) + 1
                    continue
                host = torch.from_numpy(view[1:1 + 2 * n].reshape(n, 2).copy())
            else:
                ev.synchronize()
            if not self._update(host):
                bad += 1
        return bad


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


_TOTAL_OFFSET = {}


def _total_view(geom, N):
    off = _TOTAL_OFFSET.get(N)
    if off is None:
        arr = (C.c_size_t * 6)()
        _lib.lib().dimo_raster_geom_layout(N, arr)
        off = _TOTAL_OFFSET[N] = int(arr[5])
    return geom[off:off + 8].view(torch.int32)


class RasterState:
    """Everything the backward needs (inputs are kept by reference, workspaces by ownership)."""
    __slots__ = ("settings", "N", "M", "H", "W", "r_cap", "with_normal", "means3D", "shs", "colors_precomp",
                 "opacities", "scales", "rotations", "cov3D_precomp", "view", "proj", "campos", "bg", "radii", "geom",
                 "bin_ws", "img_ws")


def raster_forward(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings,
                   with_normal=True, capacity: Optional[CapacityPolicy] = None, out=None):
    """Runs preprocess + binning + blend.  `out` = optional (color[3,H,W], depth[1,H,W], normal[3,H,W]|None,
    alpha[1,H,W]) contiguous tensors to render into (e.g. slices of a batch buffer).
    Returns (color, depth, normal|None, alpha, radii, RasterState)."""
    L = _lib.lib()
    if not means3D.is_cuda:
        raise RuntimeError("dimo_amd rasterizer needs GPU tensors (no CPU fallback in the product path)")
    dev = means3D.device
    s = settings
    N = means3D.shape[0]
    H, W = int(s.image_height), int(s.image_width)
    means3D, opacities = _f32c(means3D), _f32c(opacities)
    shs, colors_precomp = _f32c(shs), _f32c(colors_precomp)
    scales, rotations, cov3D_precomp = _f32c(scales), _f32c(rotations), _f32c(cov3D_precomp)
    if N > 0:  # the CUDA wrappers pass "absent" as empty tensors
        if shs is not None and shs.numel() == 0:
            shs = None
        if colors_precomp is not None and colors_precomp.numel() == 0:
            colors_precomp = None
        if cov3D_precomp is not None and cov3D_precomp.numel() == 0:
            cov3D_precomp = None
    if (shs is None) == (colors_precomp is None):
        raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
    if cov3D_precomp is None and (scales is None or rotations is None):
        raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    M = 0 if shs is None else shs.shape[1]
    view, proj = _f32c(s.viewmatrix), _f32c(s.projmatrix)
    campos, bg = _f32c(s.campos), _f32c(s.bg)
    stream = _lib.current_stream()

    geom = torch.empty(L.dimo_raster_geom_bytes(N), dtype=torch.uint8, device=dev)
    radii = torch.empty(N, dtype=torch.int32, device=dev)
    r_host = C.c_int64(0)
    exact = capacity is None
    _lib.check(L.dimo_raster_preprocess_forward(
        N, int(s.sh_degree), M, H, W, _lib.ptr(means3D), _lib.ptr(shs), _lib.ptr(colors_precomp),
        _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations), _lib.ptr(cov3D_precomp),
        float(s.scale_modifier), _lib.ptr(view), _lib.ptr(proj), _lib.ptr(campos), float(s.tanfovx),
        float(s.tanfovy), _lib.ptr(radii), _lib.ptr(geom), geom.numel(),
        C.byref(r_host) if exact else None, stream), "dimo_raster_preprocess_forward")
    if exact:
        r_cap = max(int(r_host.value), 1)
    else:
        r_cap = int(capacity.next_capacity())
        capacity.track(_total_view(geom, N))

    bin_ws = torch.empty(L.dimo_raster_bin_bytes(N, r_cap, H, W), dtype=torch.uint8, device=dev)
    img_ws = torch.empty(L.dimo_raster_img_bytes(H, W), dtype=torch.uint8, device=dev)
    if out is None:
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        normal = torch.empty(3, H, W, dtype=torch.float32, device=dev) if with_normal else None
        alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    else:
        color, depth, normal, alpha = out
        if not with_normal:
            normal = None
    _lib.check(L.dimo_raster_render_forward(
        N, H, W, r_cap, _lib.ptr(bg), _lib.ptr(geom), _lib.ptr(bin_ws), bin_ws.numel(), _lib.ptr(img_ws),
        img_ws.numel(), _lib.ptr(color), _lib.ptr(depth), _lib.ptr(normal), _lib.ptr(alpha), stream),
        "dimo_raster_render_forward")
    st = RasterState()
    st.settings, st.N, st.M, st.H, st.W, st.r_cap, st.with_normal = s, N, M, H, W, r_cap, with_normal
    st.means3D, st.shs, st.colors_precomp, st.opacities = means3D, shs, colors_precomp, opacities
    st.scales, st.rotations, st.cov3D_precomp = scales, rotations, cov3D_precomp
    st.view, st.proj, st.campos, st.bg, st.radii = view, proj, campos, bg, radii
    st.geom, st.bin_ws, st.img_ws = geom, bin_ws, img_ws
    return color, depth, normal, alpha, radii, st


def raster_backward(st: RasterState, g_color, g_depth, g_normal, g_alpha, out=None, scratch=None):
    """-> dict(means3D, means2D, shs|colors, opacities, scales, rotations | cov3D).  `out` may provide
    preallocated gradient tensors under the same keys (reused across the renders of a step)."""
    L = _lib.lib()
    s, N, M, H, W, r_cap = st.settings, st.N, st.M, st.H, st.W, st.r_cap
    g_color, g_depth, g_normal, g_alpha = _f32c(g_color), _f32c(g_depth), _f32c(g_normal), _f32c(g_alpha)
    dev = st.means3D.device
    out = {} if out is None else out

    def buf(key, *shape):
        t = out.get(key)
        if t is None:
            t = out[key] = torch.empty(*shape, dtype=torch.float32, device=dev)
        return t

    d_means3D, d_means2D, d_opac = buf("means3D", N, 3), buf("means2D", N, 3), buf("opacities", N, 1)
    d_shs = buf("shs", N, M, 3) if st.shs is not None else None
    d_colors = buf("colors", N, 3) if st.colors_precomp is not None else None
    d_scales = buf("scales", N, 3) if st.cov3D_precomp is None else None
    d_rot = buf("rotations", N, 4) if st.cov3D_precomp is None else None
    d_cov = buf("cov3D", N, 6) if st.cov3D_precomp is not None else None
    need = L.dimo_raster_backward_scratch_bytes(N, r_cap)
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check(L.dimo_raster_backward(
        N, int(s.sh_degree), M, H, W, r_cap, _lib.ptr(st.means3D), _lib.ptr(st.shs), _lib.ptr(st.colors_precomp),
        _lib.ptr(st.opacities), _lib.ptr(st.scales), _lib.ptr(st.rotations), _lib.ptr(st.cov3D_precomp),
        float(s.scale_modifier), _lib.ptr(st.view), _lib.ptr(st.proj), _lib.ptr(st.campos), _lib.ptr(st.bg),
        float(s.tanfovx), float(s.tanfovy), _lib.ptr(st.radii), _lib.ptr(st.geom), _lib.ptr(st.bin_ws),
        _lib.ptr(st.img_ws), _lib.ptr(g_color), _lib.ptr(g_depth), _lib.ptr(g_normal) if st.with_normal else None,
        _lib.ptr(g_alpha), _lib.ptr(d_means3D), _lib.ptr(d_means2D), _lib.ptr(d_shs), _lib.ptr(d_colors),
        _lib.ptr(d_opac), _lib.ptr(d_scales), _lib.ptr(d_rot), _lib.ptr(d_cov), _lib.ptr(scratch), scratch.numel(),
        _lib.current_stream()), "dimo_raster_backward")
    out["_scratch"] = scratch
    return out


def _checked(stage, st, tensors):
    """`raster_settings.debug` (latent_gs_renderer.py:1145): checked mode.  The published extensions snapshot the
    arguments and re-raise when a launch fails; here the call is synchronised (so an asynchronous fault surfaces at
    the stage that caused it, as a RuntimeError), the tile-instance capacity flag is read back, and every output is
    checked to be finite."""
    torch.cuda.synchronize(st.means3D.device)
    tot = _total_view(st.geom, st.N).cpu()
    if int(tot[1]) != 0 or int(tot[0]) > st.r_cap:
        raise RuntimeError(f"rasterizer {stage}: {int(tot[0])} tile instances exceed the workspace capacity "
                           f"{st.r_cap} (the render was truncated)")
    for name, t in tensors.items():
        if t is not None and not bool(torch.isfinite(t).all()):
            raise RuntimeError(f"rasterizer {stage}: non-finite values in `{name}`")


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings,
                with_normal, capacity):
        color, depth, normal, alpha, radii, st = raster_forward(
            means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings, with_normal,
            capacity)
        if settings.debug:
            _checked("forward", st, dict(image=color, depth=depth, normal=normal, alpha=alpha))
        ctx.st = st
        ctx.r_cap = st.r_cap
        ctx.opacity_shape = opacities.shape
        # keep the workspaces visible as saved tensors too (tests inspect them)
        ctx.save_for_backward(st.geom, st.bin_ws, st.img_ws)
        ctx.mark_non_differentiable(radii)
        if with_normal:
            return color, depth, normal, alpha, radii
        return color, depth, alpha, radii

    @staticmethod
    def backward(ctx, *grads):
        st = ctx.st
        if st.with_normal:
            g_color, g_depth, g_normal, g_alpha, _ = grads
        else:
            g_color, g_depth, g_alpha, _ = grads
            g_normal = None
        g = raster_backward(st, g_color, g_depth, g_normal, g_alpha)
        if st.settings.debug:
            _checked("backward", st, {k: v for k, v in g.items() if not k.startswith("_")})
        return (g["means3D"], g["means2D"], g.get("shs"), g.get("colors"), g["opacities"].view(ctx.opacity_shape),
                g.get("scales"), g.get("rotations"), g.get("cov3D"), None, None, None)


def rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                        raster_settings, with_normal=True, capacity: Optional[CapacityPolicy] = None):
    return _Rasterize.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                            raster_settings, with_normal, capacity)


class GaussianRasterizerNormal(torch.nn.Module):
    """diff_gauss flavour: (image, depth, normal, alpha, radii, extra)."""

    def __init__(self, raster_settings, capacity: Optional[CapacityPolicy] = None):
        super().__init__()
        self.raster_settings = raster_settings
        self.capacity = capacity

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3Ds_precomp=None, extra_attrs=None):
        if extra_attrs is not None:
            raise NotImplementedError("extra_attrs is not used on DIMO's path (latent_gs_renderer.py:1265 passes None)")
        image, depth, normal, alpha, radii = rasterize_gaussians(
            means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, self.raster_settings,
            True, self.capacity)
        return image, depth, normal, alpha, radii, None


class GaussianRasterizer(torch.nn.Module):
    """diff_gaussian_rasterization (ashawkey) flavour: (image, radii, depth, alpha)."""

    def __init__(self, raster_settings, capacity: Optional[CapacityPolicy] = None):
        super().__init__()
        self.raster_settings = raster_settings
        self.capacity = capacity

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        image, depth, alpha, radii = rasterize_gaussians(
            means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, self.raster_settings,
            False, self.capacity)
        return image, radii, depth, alpha


def inspect_state(ctx_tensors, N, H, W, r_cap):
    """Test helper: views of the inspectable workspace sub-buffers (see include/dimo_hip.h)."""
    L = _lib.lib()
    geom, bin_ws, img_ws = ctx_tensors
    go, bo, io = (C.c_size_t * 6)(), (C.c_size_t * 3)(), (C.c_size_t * 2)()
    L.dimo_raster_geom_layout(N, go)
    L.dimo_raster_bin_layout(r_cap, H, W, bo)
    L.dimo_raster_img_layout(H, W, io)
    T = ((W + 15) // 16) * ((H + 15) // 16)

    def view(buf, off, nbytes, dtype):
        return buf[off:off + nbytes].view(dtype)

    n = max(N, 1)
    total = view(geom, go[5], 16, torch.int32)
    ranges = view(bin_ws, bo[1], T * 8, torch.int32).view(T, 2)
    # The library stores neither half of an instance's sort key: the key's tile is the list the instance sits in
    # (`ranges`) and its 32 depth bits are its Gaussian's (gathered by dimo_raster_depth_keys), so the published 64-bit
    # (tile << 32 | depth bits) keys are rebuilt here for inspection.
    R = int(min(int(total[0]) & 0xFFFFFFFF, r_cap))
    dkeys = torch.zeros(max(r_cap, 1), dtype=torch.int32, device=geom.device)
    _lib.check(L.dimo_raster_depth_keys(N, H, W, r_cap, geom.data_ptr(), bin_ws.data_ptr(), dkeys.data_ptr(),
                                        torch.cuda.current_stream(geom.device).cuda_stream), "depth_keys")
    torch.cuda.synchronize(geom.device)
    lens = (ranges[:, 1] - ranges[:, 0]).to(torch.int64).clamp_(min=0)
    tile_of = torch.repeat_interleave(torch.arange(T, dtype=torch.int64, device=geom.device), lens)[:R]
    keys = torch.zeros(max(R, 1), dtype=torch.int64, device=geom.device)
    keys[:tile_of.numel()] = (tile_of << 32) | (dkeys[:tile_of.numel()].to(torch.int64) & 0xFFFFFFFF)
    return dict(
        splat=view(geom, go[0], n * 64, torch.float32).view(n, 16)[:N],
        rect=view(geom, go[1], n * 8, torch.int16).view(n, 4)[:N],
        tiles_touched=view(geom, go[2], n * 4, torch.int32)[:N],
        offsets=view(geom, go[3], n * 4, torch.int32)[:N],
        flags=view(geom, go[4], n, torch.uint8)[:N],
        total=total,
        keys_sorted=keys,
        depth_keys_sorted=dkeys,
        vals_sorted=view(bin_ws, bo[0], r_cap * 4, torch.int32),
        ranges=ranges,
        final_T=view(img_ws, io[0], H * W * 4, torch.float32).view(H, W),
        n_contrib=view(img_ws, io[1], H * W * 4, torch.int32).view(H, W),
    )
