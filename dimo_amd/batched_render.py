"""Batched execution behind the reference-shaped call surface.

The reference's step calls `Renderer.render(cam, time=, stage=, latent_index=)` once per (motion, view, frame) triple
and runs ONE `loss.backward()` over all of them (main_train_dimo.py:276-318, 415).  Rendered one by one, a render's
kernels cannot fill an MI355X (a 512^2 blend forward of one render takes 57 us alone, 27 us as one of eight) and every
call pays its own launch chain.  This module keeps the call surface and batches underneath:

  * `Renderer.render` hands its request to a `RenderBatcher` and returns the usual dict whose image / depth / normal /
    alpha / radii entries are `LazyTensor`s -- stand-ins that run the pending batch the first time anything looks at
    them (any torch function, method, operator or attribute).  A loop that renders all its triples and only then forms
    the losses gets ONE launch per rasterizer stage for all of them (the native step executor, dimo_amd/executor.py);
    a loop that consumes each render at once degrades to batches of one, with unchanged results.
  * A batch is ONE autograd node (`_BatchRenderFn`): its backward receives the gradient images of all its renders at
    once and runs the joint rasterizer + skinning backward the trainer's direct pipeline runs.
  * TimeNet for `deform=None` calls is evaluated once per batch for the distinct (latent, time) pairs by the fused HIP
    forward (`timenet_apply`, also used by `Trainer.batched_deform`).

GPU only: the conditions under which `Renderer.render` may batch are in `Renderer._batchable`; everything else takes
the immediate per-render path.  No CPU fallback anywhere (a request that cannot be batched is rendered by the HIP
kernels directly, never by a host implementation).
"""
import ctypes as C
import weakref

import torch

from . import _lib


# ------------------------------------------------------------------------------------------------ lazy tensors
_METAS = {}


def _meta(shape, dtype=torch.float32):
    """Cached meta tensor (shape + dtype, no storage): what a stand-in answers metadata questions from."""
    key = (tuple(shape), dtype)
    m = _METAS.get(key)
    if m is None:
        m = _METAS[key] = torch.empty(key[0], dtype=dtype, device="meta")
    return m


# methods that a PENDING stand-in answers with another stand-in (shape from the meta tensor, the work replayed at
# first use): the reference's loop body applies exactly such ops to every render before it looks at a value --
# out["image"].unsqueeze(0) (main_train_dimo.py:305,310,315,317), torch.cat per motion (:320-325),
# .permute(0, 2, 3, 1) (:364,370), [i] (:333) -- and must not cut the step's batch into batches of one
_DEFERRED_METHODS = ("unsqueeze", "squeeze", "view", "reshape", "permute", "transpose", "flatten", "contiguous",
                     "detach", "clamp", "clamp_min", "clamp_max", "expand", "movedim", "unflatten", "narrow", "select",
                     "float")


def _basic_index(idx):
    """True for an index made of ints, slices, None and Ellipsis only (a view, no tensor operand)."""
    if isinstance(idx, tuple):
        return all(_basic_index(i) for i in idx)
    return idx is None or idx is Ellipsis or isinstance(idx, (int, slice))


class LazyTensor:
    """Stand-in for a tensor that a pending batch will produce.  Not a `torch.Tensor` subclass on purpose: nothing
    about it needs the dispatcher until it is used, and the first use replaces it by the real tensor.

    Metadata (`shape`, `dtype`, `device`, `dim()`, `size()`, `len()`) is answered from the queued request without
    running anything; view-type methods, basic indexing and `torch.cat` / `torch.stack` over stand-ins return new
    stand-ins (`defer=False` switches that off: the stand-in then materialises on any access).  Everything else --
    any other torch function, method, operator or attribute -- runs the pending batch and works on the real tensor.
    A stand-in handed to a custom `autograd.Function.apply` or a C extension is NOT unwrapped by torch: pass
    `materialize(x)` there (the drop-ins of this package do)."""
    __slots__ = ("_fn", "_val", "_meta", "_dev", "_src", "_defer", "unit_range", "_rows", "__weakref__")

    def __init__(self, fn, meta=None, device=None, src=None, defer=True, unit_range=False):
        self._fn, self._val, self._meta, self._dev, self._src = fn, None, meta, device, src
        self._rows = None
        self._defer = bool(defer and meta is not None)
        # values known to lie in [0, 1] (the clamped render and its views / concatenations): lets the fused smoothness
        # drop-ins, which read rgb as clamp(rgb, 0, 1), stand in for src/loss.py:64-106 without changing a value
        self.unit_range = unit_range

    def materialize(self):
        if self._fn is not None:
            self._val, self._fn = self._fn(), None
        return self._val

    @property
    def pending(self):
        return self._fn is not None

    # ---- metadata: no flush
    def _m(self):
        return self._meta if (self._fn is not None and self._meta is not None) else self.materialize()

    shape = property(lambda self: self._m().shape)
    dtype = property(lambda self: self._m().dtype)
    ndim = property(lambda self: self._m().dim())
    device = property(lambda self: self._dev if (self._fn is not None and self._dev is not None)
                      else self.materialize().device)
    is_cuda = property(lambda self: self.device.type == "cuda")

    def dim(self):
        return self._m().dim()

    def size(self, *a):
        return self._m().size(*a)

    def numel(self):
        return self._m().numel()

    def __len__(self):
        return self._m().shape[0]

    # ---- deferred views
    def _derive(self, name, args, kwargs, src=None):
        if self._fn is None:
            val = getattr(self._val, name)(*_unwrap_all(args), **_unwrap_all(kwargs))
            if self.unit_range and _keeps_unit_range(name, args) and isinstance(val, torch.Tensor):
                out = LazyTensor(None, unit_range=True)  # (already resolved: only carries the [0, 1] tag on)
                out._val = val
                return out
            return val
        if self._defer and not any(type(a) is LazyTensor for a in args) and not kwargs:
            try:
                meta = getattr(self._meta, name)(*args)
            except Exception:
                meta = None  # (let the real tensor raise the real error)
            if meta is not None:
                parent = self
                return LazyTensor(lambda: getattr(parent.materialize(), name)(*args), meta, self._dev, src,
                                  unit_range=self.unit_range and _keeps_unit_range(name, args))
        return getattr(self.materialize(), name)(*_unwrap_all(args), **_unwrap_all(kwargs))

    def unsqueeze(self, *args, **kwargs):
        s = self._src
        src = (s[0], s[1], s[2], "unsq0") if (s is not None and s[3] == "plain" and args == (0,)) else None
        return self._derive("unsqueeze", args, kwargs, src)

    def contiguous(self, *args, **kwargs):
        return self._derive("contiguous", args, kwargs, self._src)

    def _row(self, i):
        """`self[i]` for an integer i through ONE `unbind` of the tensor, shared by all rows: the reference's loss loop
        takes `batch[...][k]` image by image (main_train_dimo.py:331-337), and a `select` per image is a backward node
        per image that zero-fills and copies a WHOLE batch-sized gradient, the engine then adding them up one by one;
        `unbind` is one node whose backward stacks the rows' gradients once."""
        if self._rows is None:
            self._rows = self.materialize().unbind(0)
        return self._rows[i]

    def __getitem__(self, idx):
        if self._fn is not None and self._defer and type(idx) is int and self._meta.dim() > 1:
            n = self._meta.shape[0]
            if -n <= idx < n:
                parent = self
                return LazyTensor(lambda: parent._row(idx), _meta(tuple(self._meta.shape[1:]), self._meta.dtype),
                                  self._dev, None, unit_range=self.unit_range)
        if self._fn is not None and self._defer and _basic_index(idx):
            s = self._src
            lead = idx == (None, Ellipsis) or idx is None or idx == (None,)
            src = (s[0], s[1], s[2], "unsq0") if (s is not None and s[3] == "plain" and lead) else None
            return self._derive("__getitem__", (idx,), {}, src)
        return self.materialize()[_unwrap_all(idx)]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func in _CAT_FUNCS:
            out = _lazy_cat(func, args, kwargs or {})
            if out is not None:
                return out
        return func(*_unwrap_all(args), **(_unwrap_all(kwargs) if kwargs else {}))

    def __getattr__(self, name):  # (only reached for names the class does not define)
        return getattr(self.materialize(), name)

    def __repr__(self):
        if self._fn is not None:
            return "LazyTensor(pending%s)" % ("" if self._meta is None else f", shape={tuple(self._meta.shape)}")
        return f"LazyTensor({self._val!r})"


def _keeps_unit_range(name, args):
    """Does method `name(*args)` map values in [0, 1] to values in [0, 1]?  (Views do; a clamp does when its lower
    bound is <= 1 and its upper bound >= 0.)"""
    if not name.startswith("clamp"):
        return True
    try:
        lo, hi = {"clamp": lambda a: (a[0] if len(a) > 0 else None, a[1] if len(a) > 1 else None),
                  "clamp_min": lambda a: (a[0], None), "clamp_max": lambda a: (None, a[0])}[name](args)
        return (lo is None or float(lo) <= 1.0) and (hi is None or float(hi) >= 0.0)
    except Exception:
        return False


def _deferred_method(name):
    def method(self, *args, **kwargs):
        return self._derive(name, args, kwargs)
    method.__name__ = name
    return method


for _n in _DEFERRED_METHODS:
    if _n not in LazyTensor.__dict__:
        setattr(LazyTensor, _n, _deferred_method(_n))

_CAT_FUNCS = {torch.cat: "cat", torch.stack: "stack"}
for _n in ("concat", "concatenate"):
    if hasattr(torch, _n):
        _CAT_FUNCS[getattr(torch, _n)] = "cat"


def _lazy_cat(func, args, kwargs):
    """torch.cat / torch.stack over stand-ins only, at least one of them pending: a stand-in for the result.  When the
    parts turn out to be the consecutive renders of ONE batch -- `out[name].unsqueeze(0)` for cat, `out[name]` for
    stack, dim 0: the reference's per-motion collection -- the result is a zero-copy slice of the batch's contiguous
    [n, C, H, W] output (and its gradient reaches the batch's autograd node as one tensor)."""
    if "out" in kwargs:
        return None
    parts = args[0] if args else kwargs.get("tensors")
    dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
    if not isinstance(parts, (list, tuple)) or not parts or not isinstance(dim, int):
        return None
    if not all(type(p) is LazyTensor and p._defer for p in parts) or not any(p._fn is not None for p in parts):
        return None
    # the result's shape by hand: torch.cat / torch.stack on META tensors run a Python decomposition (~140 us per call,
    # 1.2 ms per step of the reference's loop -- more than all the renders' own host work)
    kind = _CAT_FUNCS[func]
    s0, dt = tuple(parts[0]._meta.shape), parts[0]._meta.dtype
    nd = len(s0) + (1 if kind == "stack" else 0)
    if not -nd <= dim < nd or nd == 0:
        return None
    d = dim % nd
    if kind == "stack":
        if any(tuple(p._meta.shape) != s0 or p._meta.dtype != dt for p in parts):
            return None
        shape = s0[:d] + (len(parts),) + s0[d:]
    else:
        tot = 0
        for p in parts:
            sp = tuple(p._meta.shape)
            if len(sp) != nd or sp[:d] != s0[:d] or sp[d + 1:] != s0[d + 1:] or p._meta.dtype != dt:
                return None
            tot += sp[d]
        shape = s0[:d] + (tot,) + s0[d + 1:]
    meta = _meta(shape, dt)
    parts = tuple(parts)

    def run():
        view = _batch_view(parts, kind, dim)
        return view if view is not None else func([p.materialize() for p in parts], dim)

    return LazyTensor(run, meta, parts[0]._dev, unit_range=all(p.unit_range for p in parts))


def _batch_view(parts, kind, dim):
    want = "unsq0" if kind == "cat" else "plain"
    s0 = parts[0]._src
    if dim != 0 or s0 is None:
        return None
    pend, i0, name, _ = s0
    for k, p in enumerate(parts):
        s = p._src
        if s is None or s[0] is not pend or s[1] != i0 + k or s[2] != name or s[3] != want:
            return None
    full = pend["batcher"].full_output(pend, name)
    if full is None:
        return None
    return full if (i0 == 0 and len(parts) == full.shape[0]) else full[i0:i0 + len(parts)]


def _unwrap(x):
    return x.materialize() if isinstance(x, LazyTensor) else x


def _unwrap_all(x):
    """`x` with every LazyTensor inside lists / tuples / dicts replaced by its tensor (a pytree map costs 40 us per
    call: the reference-shaped step makes sixteen such calls and is host bound)."""
    t = type(x)
    if t is LazyTensor:
        return x.materialize()
    if t is list or t is tuple:
        return t(_unwrap_all(y) for y in x)
    if t is dict:
        return {k: _unwrap_all(v) for k, v in x.items()}
    return x


def _forward_dunder(name):
    def method(self, *args, **kwargs):
        return getattr(self.materialize(), name)(*_unwrap_all(args), **_unwrap_all(kwargs))
    method.__name__ = name
    return method


for _n in ("add", "sub", "mul", "truediv", "floordiv", "pow", "matmul", "mod", "and", "or", "xor", "lshift", "rshift"):
    setattr(LazyTensor, f"__{_n}__", _forward_dunder(f"__{_n}__"))
    setattr(LazyTensor, f"__r{_n}__", _forward_dunder(f"__r{_n}__"))
for _n in ("neg", "pos", "abs", "invert", "lt", "le", "gt", "ge", "eq", "ne", "setitem", "iter",
           "bool", "float", "int", "index", "contains", "format", "array"):
    setattr(LazyTensor, f"__{_n}__", _forward_dunder(f"__{_n}__"))
LazyTensor.__hash__ = lambda self: id(self)


def materialize(x):
    """The real tensor behind `x` (identity for anything that is not a LazyTensor)."""
    return _unwrap(x)


# ------------------------------------------------------------------------------------------------ fused TimeNet
_TN_POOL = weakref.WeakKeyDictionary()  # net -> FusedTimeNet wrappers not in flight (one wraps twenty ctypes pointers)


class _TimeNetFn(torch.autograd.Function):
    """TimeNet on the HIP library as one autograd node.  The gradients of the network's OWN parameters are added to
    their `.grad` by the backward kernel (FusedTimeNet.backward); control points and latents get theirs returned."""

    @staticmethod
    def forward(ctx, net, pts, latent_table, times, rows):
        from .fused_timenet import FusedTimeNet
        free = _TN_POOL.setdefault(net, [])
        # (a wrapper's workspace belongs to ONE forward until its backward has run: several forwards may precede one
        # backward, each then takes its own wrapper)
        fused = free.pop() if free else FusedTimeNet(net)
        d_xyz, d_rot = fused.forward(pts.detach().contiguous(), times, latent_table.detach().contiguous(), rows)
        ctx.fused, ctx.shapes, ctx.net = fused, (pts.shape, latent_table.shape), net
        return d_xyz, d_rot

    @staticmethod
    def backward(ctx, g_xyz, g_rot):
        pshape, lshape = ctx.shapes
        dev = g_xyz.device
        g_pts = torch.zeros(pshape, dtype=torch.float32, device=dev)
        g_lat = torch.zeros(lshape, dtype=torch.float32, device=dev)
        ctx.fused.backward(g_xyz.contiguous(), g_rot.contiguous(), g_pts, g_lat)
        free = _TN_POOL.setdefault(ctx.net, [])
        if len(free) < 8:
            free.append(ctx.fused)
        ctx.fused = None
        return None, g_pts, g_lat, None, None


def timenet_fusable(net, pts):
    return pts.is_cuda and len(getattr(net, "skips", [0, 0])) <= 1 and pts.dtype == torch.float32


def timenet_apply(net, pts, latent_table, times, rows=None):
    """d_xyz [P,M,3], d_rot [P,M,4] of `net` for P (latent row, time) pairs in ONE fused launch chain, differentiable
    w.r.t. pts, the latent table and (by side effect on .grad) the network's parameters."""
    if net.deformnet[0].weight.grad is None or net.rot_layers[-1].bias.grad is None:
        for p in net.parameters():  # the backward kernel ADDS to .grad: every parameter needs one
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p)
    return _TimeNetFn.apply(net, pts, latent_table, [float(t) for t in times], None if rows is None else list(rows))


# ------------------------------------------------------------------------------------------------ the batch node
class _Ticket:
    """Holds a batch's render slots; gives them back when the autograd graph (or a no-grad caller) lets go."""

    def __init__(self, batcher, first, count):
        self.batcher, self.first, self.count = weakref.ref(batcher), first, count

    def release(self):
        b = self.batcher()
        if b is not None and self.count:
            b._release(self.first, self.count)
        self.count = 0

    def __del__(self):
        self.release()


def _aligned_views(dev, shapes):
    """One zero-filled allocation carved into fp32 tensors of `shapes`, every start 16-byte aligned."""
    offs, o = [], 0
    for s in shapes:
        n = 1
        for d in s:
            n *= d
        offs.append((o, n))
        o += (n + 3) // 4 * 4
    buf = torch.zeros(max(o, 4), dtype=torch.float32, device=dev)
    return [buf[a:a + n].view(s) for (a, n), s in zip(offs, shapes)]


class _BatchRenderFn(torch.autograd.Function):
    """All renders of a pending batch: skinning -> projection -> binning -> blend, one launch per stage for the batch
    (native step executor, fully batched on the caller's stream).  Inputs: the canonical Gaussians' raw parameters,
    control points, then per DISTINCT deformation its (d_xyz, d_rot), then one screen-space gradient sink per render.
    Outputs, each for the whole batch: clamped image [n,3,H,W], depth [n,1,H,W], normal [n,3,H,W] (or an empty
    tensor), alpha [n,1,H,W], radii [n,N]."""

    @staticmethod
    def forward(ctx, job, xyz, rotation, scaling, opacity, f_dc, c_xyz, c_radius, *rest):
        b, ex, n = job.batcher, job.batcher.ex, len(job.reqs)
        nd = len(job.deforms)
        deform_t, sinks = rest[:2 * nd], rest[2 * nd:]
        dev = xyz.device
        H, W = ex.H, ex.W
        f32 = dict(dtype=torch.float32, device=dev)
        raw = torch.empty(n, 3, H, W, **f32)
        depth, alpha = torch.empty(n, 1, H, W, **f32), torch.empty(n, 1, H, W, **f32)
        normal = torch.empty(n, 3, H, W, **f32) if b.with_normal else None
        radii = torch.empty(n, ex.N, dtype=torch.int32, device=dev)
        c = ex.common
        p = _lib.ptr
        c.stage1, c.log_r, c.g_log_r = 0, None, None
        c.N, c.M, c.H, c.W = ex.N, ex.M, H, W
        c.with_normal, c.local_frame, c.R_cap = int(b.with_normal), int(job.local_frame), ex.r_cap
        keep = [t.detach().contiguous() for t in (xyz, rotation, scaling, opacity, f_dc, c_xyz, c_radius)]
        c.xyz, c.rotation, c.scaling, c.opacity, c.f_dc, c.c_xyz, c.c_log_radius = [p(t) for t in keep]
        c.nn_dist, c.nn_idx, c.bg = p(job.nn_dist), p(job.nn_idx), p(job.bg)
        c.scale_modifier = float(job.scale_modifier)
        c.lbs_scratch, c.lbs_scratch_bytes = p(ex.lbs_scratch), ex.lbs_scratch.numel()
        c.geom_bytes, c.bin_bytes, c.img_bytes, c.bwd_scratch_bytes = ex.geom_bytes, ex.bin_bytes, ex.img_bytes, ex.bwd_bytes
        dkeep = [t.detach().contiguous() for t in deform_t]
        HW4 = H * W * 4
        for i, rq in enumerate(job.reqs):
            d = ex.descs[job.first + i]
            cam = rq.cam
            d.view, d.proj, d.campos = p(cam.world_view_transform), p(cam.full_proj_transform), p(cam.camera_center)
            d.tanfovx, d.tanfovy = rq.tanfovx, rq.tanfovy
            d.d_xyz, d.d_rot = p(dkeep[2 * rq.deform]), p(dkeep[2 * rq.deform + 1])
            d.g_d_xyz, d.g_d_rot = None, None  # (set by the backward; equal within a deformation group either way)
            d.out_color, d.out_depth = raw.data_ptr() + i * 3 * HW4, depth.data_ptr() + i * HW4
            d.out_normal = (normal.data_ptr() + i * 3 * HW4) if normal is not None else None
            d.out_alpha = alpha.data_ptr() + i * HW4
            d.radii = radii.data_ptr() + i * ex.N * 4
            d.g_color = d.g_depth = d.g_normal = d.g_alpha = d.g_dot = None
        # deformation groups are found by pointer equality of (d_xyz, d_rot, g_d_xyz, g_d_rot): give the renders of one
        # deformation the same (still unset) gradient rows NOW so that the forward groups exactly like the backward
        job.grad_rows = _aligned_views(dev, [s for t in dkeep for s in (tuple(t.shape),)])
        for i, rq in enumerate(job.reqs):
            d = ex.descs[job.first + i]
            d.g_d_xyz, d.g_d_rot = p(job.grad_rows[2 * rq.deform]), p(job.grad_rows[2 * rq.deform + 1])
        ex.forward_range(job.first, n)
        if b.capacity is not None:
            # a SNAPSHOT: the binning's fill kernel rewrites these words whenever the slots are re-used, and a renderer
            # nobody polls looks at them only every 64 renders (8n bytes, same stream)
            b.capacity.track(ex.total_words_range(job.first, n).clone())
        image = raw.clamp(0.0, 1.0)
        # where the clamp changed nothing (bounds included) the gradient passes.  (The mask, not `image`, is kept: an
        # OUTPUT held by its own node's ctx is a reference cycle, and the render slots would wait for the garbage
        # collector instead of coming back when the graph is dropped.)
        ctx.job, ctx.keep, ctx.n = job, (keep, dkeep, raw, radii, image == raw), n
        ctx.ticket = job.ticket
        ctx.mark_non_differentiable(radii)
        # the batch's outputs stay whole: a render's image is a select of [n, 3, H, W] made outside this node, and the
        # reference's per-motion torch.cat (main_train_dimo.py:320-325) is a slice of it (`_batch_view`)
        return image, depth, normal if normal is not None else raw.new_empty(0), alpha, radii

    @staticmethod
    def backward(ctx, *grads):
        job, n = ctx.job, ctx.n
        if job.backward_done:
            # (retain_graph=True / two losses backpropagated separately: the render slots went back to the batcher with
            # the first pass and `grad_rows` still hold its sums -- a second pass would be silently wrong)
            raise RuntimeError("a render batch's backward runs once: its render slots were released by the first pass; "
                               "sum the losses and call backward() once (main_train_dimo.py:415)")
        job.backward_done = True
        b, ex = job.batcher, job.batcher.ex
        keep, dkeep, raw, _radii, passes = ctx.keep
        dev = raw.device
        H, W = ex.H, ex.W
        N, M = ex.N, ex.M
        f32 = dict(dtype=torch.float32, device=dev)

        g_img, g_depth, g_normal, g_alpha = [None if g is None else g.contiguous() for g in grads[:4]]
        if not b.with_normal:
            g_normal = None
        if g_img is None:
            g_img = torch.zeros(n, 3, H, W, **f32)
        else:  # through the clamp of the returned image
            g_img = g_img * passes
        if g_alpha is None:
            g_alpha = torch.zeros(n, 1, H, W, **f32)
        (g_xyz, g_rot, g_scaling, g_opacity, g_fdc, g_cxyz, g_crad) = _aligned_views(
            dev, [(N, 3), (N, 4), (N, 3), (N, 1), (N, 1, 3), (M, 3), (M, 1)])
        c = ex.common  # (job.grad_rows were zero-filled when they were made and nothing has written them since)
        p = _lib.ptr
        c.xyz, c.rotation, c.scaling, c.opacity, c.f_dc, c.c_xyz, c.c_log_radius = [p(t) for t in keep]
        c.nn_dist, c.nn_idx, c.bg = p(job.nn_dist), p(job.nn_idx), p(job.bg)
        c.with_normal, c.local_frame, c.scale_modifier = int(b.with_normal), int(job.local_frame), float(job.scale_modifier)
        c.g_xyz, c.g_rotation, c.g_scaling, c.g_opacity, c.g_f_dc = p(g_xyz), p(g_rot), p(g_scaling), p(g_opacity), p(g_fdc)
        c.g_c_xyz, c.g_c_log_radius = p(g_cxyz), p(g_crad)
        HW4 = H * W * 4
        # the screen-space gradients go to a tensor of this backward (not to the slots' buffers, which the next batch
        # overwrites): the sinks' .grad may keep views of it
        gm2d = torch.empty(n, N, 3, **f32)
        for i in range(n):
            d = ex.descs[job.first + i]
            d.g_means2D = gm2d.data_ptr() + i * N * 12
            d.g_color, d.g_alpha = g_img.data_ptr() + i * 3 * HW4, g_alpha.data_ptr() + i * HW4
            d.g_depth = (g_depth.data_ptr() + i * HW4) if g_depth is not None else None
            d.g_normal = (g_normal.data_ptr() + i * 3 * HW4) if g_normal is not None else None
            d.g_dot = None
        ex.backward_launch(job.first, n)
        ex.backward_accumulate(job.first, n)
        # screen-space gradients (densification statistics read them)
        sink_grads = [gm2d[i] if ctx.needs_input_grad[8 + 2 * len(job.deforms) + i] else None for i in range(n)]
        ctx.ticket.release()
        return (None, g_xyz, g_rot, g_scaling, g_opacity, g_fdc, g_cxyz, g_crad, *job.grad_rows, *sink_grads)


class _Request:
    __slots__ = ("cam", "tanfovx", "tanfovy", "deform", "sink", "index")


class _Job:
    """What a flushed batch's autograd node needs (NOT its outputs: those reference the node)."""
    __slots__ = ("batcher", "reqs", "deforms", "first", "ticket", "local_frame", "scale_modifier", "nn_dist", "nn_idx",
                 "bg", "grad_rows", "backward_done")


class RenderBatcher:
    """Collects `Renderer.render` requests and runs them as batches on a `StepExecutor` (all on the caller's stream).
    A request takes its render slot when it is queued (the slots of a batch are consecutive); a slot is free again
    when its batch's backward has run or its autograd graph is gone."""

    def __init__(self, renderer, slots=16):
        self.renderer = weakref.proxy(renderer)
        self.n_slots = slots
        self.ex = None
        self.in_use = []
        self.pending = None
        self._sink_zeros = None
        self.flushes = 0   # (diagnostics: batches run, renders in them)
        self.rendered = 0

    # ---- slots
    def _ensure_executor(self, N, M, H, W):
        """True if the executor fits (N, M, H, W); re-created only while none of its slots is taken."""
        from .executor import StepExecutor
        cap = self.capacity.next_capacity()
        ex = self.ex
        busy = any(self.in_use) if ex is not None else False
        if ex is not None and (ex.N, ex.M, ex.H, ex.W) == (N, M, H, W):
            if ex.r_cap != cap and not busy:
                ex.resize_capacity(cap)
            return True
        if busy:
            return False  # renders whose backward is still to come live in the old slots: the caller renders directly
        if ex is not None:
            torch.cuda.synchronize()
            self.ex = None
            del ex
        self.ex = StepExecutor(N, M, H, W, self.n_slots, cap, self.renderer.device, n_streams=0)
        self.in_use = [False] * self.n_slots
        return True

    def _release(self, first, count):
        for j in range(first, first + count):
            if j < len(self.in_use):
                self.in_use[j] = False

    @property
    def capacity(self):
        return self.renderer.capacity_policy()

    @property
    def with_normal(self):
        return bool(self.renderer.add_normal)

    # ---- requests
    def add(self, cam, tanfovx, tanfovy, key, deform, time, latent_index):
        """Queues one render.  key = what a batch shares: (H, W, scale_modifier, local_frame), the autograd mode of the
        caller and the identity of the model (row counts, parameter storage) -- a change of any of them runs the
        pending batch first.  Returns (batch record, index in it), or None if no render slot is free (the caller then
        renders directly)."""
        g = self.renderer.gaussians
        key = (*key, torch.is_grad_enabled(), g._xyz.shape[0], g._c_xyz.shape[0], g._xyz.data_ptr())
        pend = self.pending
        if pend is not None and pend["key"] != key:
            self.flush()
            pend = None
        if pend is None and not self._ensure_executor(g._xyz.shape[0], g._c_xyz.shape[0], key[0], key[1]):
            return None
        if pend is not None:
            nxt = pend["first"] + len(pend["reqs"])
            if nxt >= len(self.in_use) or self.in_use[nxt]:
                self.flush()
                pend = None
                if not self._ensure_executor(g._xyz.shape[0], g._c_xyz.shape[0], key[0], key[1]):
                    return None
        if pend is None:
            free = [i for i, used in enumerate(self.in_use) if not used]
            if not free:
                return None
            # the start of the longest free run: the batch may grow there
            best, best_len, i = free[0], 0, 0
            while i < len(self.in_use):
                if self.in_use[i]:
                    i += 1
                    continue
                j = i
                while j < len(self.in_use) and not self.in_use[j]:
                    j += 1
                if j - i > best_len:
                    best, best_len = i, j - i
                i = j
            pend = self.pending = dict(key=key, first=best, reqs=[], deforms=[], deform_ids={}, lazy_pairs=[],
                                       full=None, selects={}, error=None, batcher=self, leaders=None)
            nxt = best
        self.in_use[nxt] = True
        rq = _Request()
        # gradient sink for the screen-space means (latent_gs_renderer.py:1114-1126: zeros with requires_grad; the
        # densification statistics read its .grad): a leaf that SHARES a persistent block of zeros -- nobody writes
        # a sink's values -- instead of a fresh 1.2 MB memset per render
        z = self._sink_zeros
        if z is None or z.shape[1] != g._xyz.shape[0] or z.shape[0] != len(self.in_use):
            z = self._sink_zeros = torch.zeros(len(self.in_use), g._xyz.shape[0], 3, dtype=torch.float32,
                                               device=g._xyz.device)
        sink = z[nxt].detach().requires_grad_(True)
        rq.cam, rq.tanfovx, rq.tanfovy, rq.sink, rq.index = cam, float(tanfovx), float(tanfovy), sink, len(pend["reqs"])
        if deform is not None:
            dx, dq = deform
            k = (dx.data_ptr(), dq.data_ptr(), tuple(dx.shape))
            if k not in pend["deform_ids"]:
                pend["deform_ids"][k] = len(pend["deforms"])
                pend["deforms"].append((dx, dq))
        else:  # TimeNet is evaluated when first needed, once for the distinct (latent, time) pairs queued by then
            k = ("lazy", latent_index, float(time))
            if k not in pend["deform_ids"]:
                pend["deform_ids"][k] = len(pend["deforms"])
                pend["deforms"].append(None)
                pend["lazy_pairs"].append((len(pend["deforms"]) - 1, latent_index, float(time)))
        rq.deform = pend["deform_ids"][k]
        pend["reqs"].append(rq)
        return pend, rq.index

    def _ran(self, pend):
        """Runs `pend` if it is still the pending batch; raises if it failed."""
        if pend["full"] is None and pend["error"] is None:
            self.flush(pend)
        if pend["error"] is not None:
            raise RuntimeError("the batch this output belongs to failed: %r" % (pend["error"],)) from pend["error"]

    def full_output(self, pend, name):
        """Output `name` of the whole batch: [n, C, H, W] (radii: [n, N])."""
        self._ran(pend)
        return pend["full"][name]

    def output(self, pend, index, name):
        """Output `name` of render `index` of the batch (a select of the batch's tensor, made once)."""
        self._ran(pend)
        key = (index, name)
        t = pend["selects"].get(key)
        if t is None:
            full = pend["full"][name]
            t = pend["selects"][key] = None if full is None else full[index]
        return t

    def pts_slot(self, pend, index):
        self._ran(pend)
        return pend["leaders"][index]

    def deform_of(self, pend, slot):
        """(d_xyz [M,3], d_rot [M,4]) of the batch's deformation `slot`.  Lazily evaluated pairs run through ONE fused
        TimeNet forward for all the pairs queued so far and not yet evaluated -- WITHOUT running the batch's renders:
        the reference reads out["cpts_t"] right after every render() (geometry-anchor term, main_train_dimo.py:295-303)."""
        d = pend["deforms"][slot]
        if d is None:
            self._eval_lazy_pairs(pend)
            d = pend["deforms"][slot]
        return d

    def _eval_lazy_pairs(self, pend):
        pairs = [q for q in pend["lazy_pairs"] if pend["deforms"][q[0]] is None]
        if not pairs:
            return
        g = self.renderer.gaussians
        with torch.set_grad_enabled(pend["key"][4]):
            if g.vae_latent:
                table, rows = torch.stack([g.latent_code(li) for (_, li, _) in pairs]), None
            else:
                table, rows = g._latent_codes, [li for (_, li, _) in pairs]
            dx, dq = timenet_apply(g._timenet, g._c_xyz, table, [t for (_, _, t) in pairs], rows)
            for j, (slot, _, _) in enumerate(pairs):
                pend["deforms"][slot] = (dx[j], dq[j])

    def flush(self, pend=None):
        """Runs the pending batch (no-op if there is none or `pend` is not the pending one)."""
        if self.pending is None or (pend is not None and pend is not self.pending):
            return
        pend, self.pending = self.pending, None
        n = len(pend["reqs"])
        try:
            with torch.set_grad_enabled(pend["key"][4]):  # (the mode the renders were queued in)
                self._run(pend)
        except BaseException as e:  # the stand-ins of this batch re-raise it; its slots are free again
            pend["error"] = e
            self._release(pend["first"], n)
            raise
        finally:
            pend["reqs"] = None

    def _run(self, pend):
        r = self.renderer
        g = r.gaussians
        reqs = pend["reqs"]
        n = len(reqs)
        _H, _W, scale_modifier, local_frame = pend["key"][:4]
        self._eval_lazy_pairs(pend)
        deforms = pend["deforms"]
        job = _Job()
        job.batcher, job.reqs, job.deforms = self, reqs, deforms
        job.local_frame, job.scale_modifier = bool(local_frame), float(scale_modifier)
        job.nn_dist, job.nn_idx, job.bg = g.neighbor_dists, g.neighbor_indices, r.bg_color
        job.first, job.ticket = pend["first"], _Ticket(self, pend["first"], n)
        job.backward_done = False
        flat = [t for dq in deforms for t in dq]
        sinks = [rq.sink for rq in reqs]
        image, depth, normal, alpha, radii = _BatchRenderFn.apply(
            job, g._xyz, g._rotation, g._scaling, g._opacity, g._features_dc, g._c_xyz, g._c_radius, *flat, *sinks)
        # (the skinned Gaussians of a deformation group live in the slot of its first render of the launch chunk)
        lead, leaders = {}, []
        for i, rq in enumerate(reqs):
            leaders.append(pend["first"] + lead.setdefault((i // 8, rq.deform), i))
        pend["leaders"] = leaders
        pend["full"] = dict(image=image, depth=depth, normal=normal if self.with_normal else None, alpha=alpha,
                            radii=radii)
        self.flushes += 1
        self.rendered += n
        if not (image.requires_grad or depth.requires_grad or alpha.requires_grad):
            job.ticket.release()  # nothing will come back for these renders: their slots are free again
        job.ticket = None  # (the autograd node holds the only other reference)
