"""The native step executor (dimo_amd/csrc/executor.hip) over the BATCHED kernels the benchmark times -- skinning,
projection, tile binning, blend forward; blend backward, projection backward, skinning backward and the fold into the
shared gradient views: deform.hip, preprocess.hip, binning.hip, blend.hip as hipcc compiles them -- run on the CPU SIMT
emulation (tests/simt/) in each of the executor's three modes and each of its backward call sequences, per render
against the C oracle like tests/test_gpu_executor.py (integer stages bit for bit, images and per-Gaussian gradients
within 1e-4), and the ACCUMULATED gradients of the step (canonical Gaussians, control points, the TimeNet rows of every
(motion, frame) pair) against autograd through oracle/deform_ref.py fed the oracle rasterizer's gradients.

Streams and events are no-ops here (a launch has run when the call returns): what is tested is what the executor
launches, over which renders, in which groups and chunks -- not its cross-stream schedule, which stays with
tests/test_gpu_deform.py's C3-size schedule test on the GPU."""
import ctypes as C

import numpy as np
import pytest
import torch

from dimo_amd.executor import RenderDesc, StepCommon
from oracle import raster_oracle as ro
from oracle.deform_ref import skinning_ref
from tests.scenes import camera_np, random_scene
from tests.simt import build as simt_build
from tests.simt import harness as hz

L1_TOL = 1e-4
_LIBS = {}
_CURRENT = "step"  # which build of the executor S_() hands out (the broken-schedule variants: see the last test)


def S_():
    if _CURRENT not in _LIBS:
        lib = C.CDLL(simt_build.build(target=_CURRENT))
        p, i, q = C.c_void_p, C.c_int, C.c_int64
        lib.dimo_executor_create.argtypes = [i]
        lib.dimo_executor_create.restype = p
        lib.dimo_executor_destroy.argtypes = [p]
        for name in ("forward_range", "backward_launch", "backward_launch_in_order", "backward_skinning_in_order",
                     "backward_launch_joint", "backward_accumulate"):
            getattr(lib, "dimo_executor_" + name).argtypes = [p, p, i, i, p, p]
        lib.dimo_executor_forward.argtypes = [p, p, i, p, p]
        lib.dimo_executor_join.argtypes = [p, i, i, p]
        lib.dimo_executor_join_ranges.argtypes = [p, i, i, p]
        lib.simt_step_layout.argtypes = [i, i, i, i, q, C.POINTER(C.c_size_t)]
        lib.simt_synchronize.argtypes = []
        lib.simt_enqueue_copy.argtypes = [p, p, C.c_size_t, p]
        lib.dimo_executor_range_stream.argtypes = [p, i]
        lib.dimo_executor_range_stream.restype = p
        lib.dimo_executor_private_stream.argtypes = [p, i]
        lib.dimo_executor_private_stream.restype = p
        lib.dimo_executor_side_done.argtypes = [p, i]
        lib.dimo_executor_wait_side.argtypes = [p, i, p]
        _LIBS[_CURRENT] = lib
    return _LIBS[_CURRENT]


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class Step:
    """A model, n renders (pair_of[i] = the (motion, frame) pair render i shows) and every buffer of
    dimo_amd/executor.py's StepExecutor, in host memory."""

    def __init__(self, N, M, H, W, pair_of, seed=0, stage1=False, r_cap=None, scale=0.03, opacity=(0.2, 0.95)):
        self.N, self.M, self.H, self.W, self.pair_of, self.n = N, M, H, W, pair_of, len(pair_of)
        self.stage1 = stage1  # stage s1: the TimeNet moves every Gaussian itself, one shared log-radius
        n, P = self.n, max(pair_of) + 1
        rng = np.random.default_rng(seed)
        sc = random_scene(N, seed=seed, scale=scale, opacity=opacity)
        self.xyz = _f32(sc["means3D"])
        self.rotation = _f32(sc["rotations"] * rng.uniform(0.5, 2.0, (N, 1)))  # (raw: not unit length)
        self.scaling = _f32(np.log(sc["scales"]))
        op = sc["opacities"]
        self.opacity = _f32(np.log(op / (1 - op)))
        self.f_dc = _f32(sc["shs"][:, :1])
        self.c_xyz = _f32(rng.uniform(-0.5, 0.5, (M, 3)))
        self.c_log_radius = _f32(np.log(rng.uniform(0.05, 0.25, (M, 1))))
        d = torch.cdist(torch.from_numpy(self.xyz), torch.from_numpy(self.c_xyz))
        nd, ni = torch.topk(d, 4, dim=1, largest=False)
        self.nn_dist, self.nn_idx = _f32(nd.numpy()), np.ascontiguousarray(ni.numpy(), np.int64)
        self.d_xyz = _f32(0.02 * rng.standard_normal((P, N if stage1 else M, 3)))
        self.log_r = _f32([[np.log(0.03)]])
        self.d_rot = _f32(np.array([1.0, 0, 0, 0]) + 0.1 * rng.standard_normal((P, M, 4)))
        self.bg = _f32([0.2, 0.5, 0.8])
        self.cams = [camera_np(37.0 * i + 11.0 * pair_of[i], elevation=5.0 * (i % 3) - 5.0, W=W, H=H) for i in range(n)]
        self.cam_arrays = [[_f32(c[k]).reshape(-1) for k in ("view", "proj", "campos")] for c in self.cams]
        self.gw = [_f32(rng.standard_normal((n, c, H, W))) for c in (3, 1, 3, 1)]
        self.r_cap = 64 * N if r_cap is None else r_cap
        lay = (C.c_size_t * 8)()
        S_().simt_step_layout(N, M, H, W, self.r_cap, lay)
        self.lay = dict(zip(("geom", "bin", "img", "bwd", "lbs", "vals", "ranges", "total"), (int(x) for x in lay)))
        self.fresh()

    def fresh(self, fill=0x5A, keep_bin=False):
        """New outputs, workspaces and zeroed gradient accumulators; the descriptors over them.  The gradient images
        the descriptors point to start as NaN: `losses` puts the real ones there.  keep_bin: the bin workspaces keep
        what the previous step left in them, like the trainer's persistent render slots."""
        N, M, H, W, n, L = self.N, self.M, self.H, self.W, self.n, self.lay
        P = max(self.pair_of) + 1
        nan = lambda *s: np.full(s, np.nan, np.float32)
        self.out = dict(color=nan(n, 3, H, W), depth=nan(n, 1, H, W), normal=nan(n, 3, H, W), alpha=nan(n, 1, H, W))
        self.seen = {k: nan(*v.shape) for k, v in self.out.items()}  # the images as the "loss kernels" saw them
        self.gw_live = [nan(*g.shape) for g in self.gw]
        # (the workspaces are allocated once and re-filled: a few hundred MB per step otherwise go through mmap / munmap)
        if not hasattr(self, "_ws"):
            self._ws = [dict(geom=hz.workspace(L["geom"], fill), bin=hz.workspace(L["bin"], fill),
                             img=hz.workspace(L["img"], fill), bwd_scratch=hz.workspace(L["bwd"], fill)) for _ in range(n)]
        for w in self._ws:
            for k_, a_ in w.items():
                if not (keep_bin and k_ == "bin"):
                    a_[...] = fill
        self.slots = [dict(pts=nan(N, 3), rot=nan(N, 4), scales=nan(N, 3), opac=nan(N, 1), radii=np.full(N, -1, np.int32),
                           **self._ws[i],
                           g_means3D=nan(N, 3), g_means2D=nan(N, 3), g_shs=nan(N, 1, 3), g_opac=nan(N, 1),
                           g_scales=nan(N, 3), g_rot=nan(N, 4)) for i in range(n)]
        self.acc = dict(xyz=np.zeros((N, 3), np.float32), rotation=np.zeros((N, 4), np.float32),
                        scaling=np.zeros((N, 3), np.float32), opacity=np.zeros((N, 1), np.float32),
                        f_dc=np.zeros((N, 1, 3), np.float32), c_xyz=np.zeros((M, 3), np.float32),
                        c_log_radius=np.zeros((M, 1), np.float32), d_xyz=np.zeros(self.d_xyz.shape, np.float32),
                        d_rot=np.zeros((P, M, 4), np.float32), log_r=np.zeros((1, 1), np.float32))
        self.lbs_scratch = hz.workspace(L["lbs"] * n, fill)
        self.totals = np.zeros((n, 2), np.int32)
        p = lambda a: a.ctypes.data
        c = self.common = StepCommon()
        c.N, c.M, c.H, c.W, c.with_normal, c.local_frame, c.R_cap = N, M, H, W, 1, 1, self.r_cap
        c.xyz, c.rotation, c.scaling, c.opacity, c.f_dc = p(self.xyz), p(self.rotation), p(self.scaling), p(self.opacity), p(self.f_dc)
        c.c_xyz, c.c_log_radius, c.nn_dist, c.nn_idx, c.bg = p(self.c_xyz), p(self.c_log_radius), p(self.nn_dist), p(self.nn_idx), p(self.bg)
        c.scale_modifier = 1.0
        a = self.acc
        c.g_xyz, c.g_rotation, c.g_scaling, c.g_opacity = p(a["xyz"]), p(a["rotation"]), p(a["scaling"]), p(a["opacity"])
        c.g_f_dc, c.g_c_xyz, c.g_c_log_radius = p(a["f_dc"]), p(a["c_xyz"]), p(a["c_log_radius"])
        c.lbs_scratch, c.lbs_scratch_bytes = p(self.lbs_scratch), self.lbs_scratch.nbytes
        c.geom_bytes, c.bin_bytes, c.img_bytes, c.bwd_scratch_bytes = L["geom"], L["bin"], L["img"], L["bwd"]
        if self.stage1:
            c.stage1, c.log_r, c.g_log_r, c.nn_dist, c.nn_idx = 1, p(self.log_r), p(a["log_r"]), None, None
        self.descs = (RenderDesc * n)()
        for i, (d, s) in enumerate(zip(self.descs, self.slots)):
            for k in ("pts", "rot", "scales", "opac", "radii", "geom", "img", "bin", "bwd_scratch", "g_means3D",
                      "g_means2D", "g_shs", "g_opac", "g_scales", "g_rot"):
                setattr(d, k, p(s[k]))
            view, proj, campos = self.cam_arrays[i]
            d.view, d.proj, d.campos = p(view), p(proj), p(campos)
            d.tanfovx, d.tanfovy = self.cams[i]["tanfovx"], self.cams[i]["tanfovy"]
            q = self.pair_of[i]
            d.d_xyz, d.d_rot = p(self.d_xyz[q]), p(self.d_rot[q])
            d.g_d_xyz, d.g_d_rot = p(a["d_xyz"][q]), p(a["d_rot"][q])
            for k, name in (("color", "out_color"), ("depth", "out_depth"), ("normal", "out_normal"), ("alpha", "out_alpha")):
                setattr(d, name, p(self.out[k][i]))
            for g, name in zip(self.gw_live, ("g_color", "g_depth", "g_normal", "g_alpha")):
                setattr(d, name, p(g[i]))
            d.totals_out = p(self.totals[i])

    # ---- the reference: per render the C oracle, for the step autograd through the skinning restatement
    def reference(self):
        if hasattr(self, "_ref"):
            return self._ref
        t = lambda a: torch.from_numpy(a).double().requires_grad_(True)
        names = ("xyz", "rotation", "scaling", "opacity", "c_xyz", "c_log_radius")
        leaves = {k: t(getattr(self, k)) for k in names}
        d_xyz, d_rot = t(self.d_xyz), t(self.d_rot)
        log_r = t(self.log_r)
        per_render, f_dc_grad = [], np.zeros((self.N, 1, 3))
        for i in range(self.n):
            q, cam = self.pair_of[i], self.cams[i]
            if self.stage1:  # renderer/latent_gs_renderer.py:1176-1177, 1211-1212, get_scaling :341-351
                rot = leaves["rotation"]
                outs = (leaves["xyz"] + d_xyz[q], rot / rot.norm(dim=1, keepdim=True).clamp_min(1e-12),
                        torch.exp(log_r).expand(self.N, 3), torch.sigmoid(leaves["opacity"]))
            else:
                outs = skinning_ref(**leaves, d_xyz=d_xyz[q], d_rot=d_rot[q], nn_dist=torch.from_numpy(self.nn_dist).double(),
                                    nn_idx=torch.from_numpy(self.nn_idx), local_frame=True)
            pts, rot, scales, opac = (_f32(o.detach().numpy()) for o in outs)
            o = ro.forward(pts, self.f_dc, None, opac, scales, rot, None, 1.0, cam["view"], cam["proj"], cam["campos"],
                           self.bg, cam["tanfovx"], cam["tanfovy"], self.H, self.W, 0, f64=False)
            go = ro.backward(o, *[g[i] for g in self.gw])
            tg = lambda a, ref: torch.from_numpy(np.asarray(a, np.float64)).reshape(ref.shape)
            torch.autograd.backward(list(outs), [tg(go["dL_dmeans3D"], outs[0]), tg(go["dL_drot"], outs[1]),
                                                 tg(go["dL_dscales"], outs[2]), tg(go["dL_dopacity"], outs[3])])
            f_dc_grad += go["dL_dshs"].reshape(self.N, 1, 3)
            per_render.append((o, go, (pts, rot, scales, opac)))
        acc = {k: v.grad.numpy() for k, v in leaves.items() if v.grad is not None}
        acc["f_dc"], acc["d_xyz"] = f_dc_grad, d_xyz.grad.numpy()
        if self.stage1:
            acc["log_r"] = log_r.grad.numpy()
        else:
            acc["d_rot"] = d_rot.grad.numpy()
        self._ref = (per_render, acc)
        return self._ref

    def check_forward(self):
        S_().simt_synchronize()  # (deferred streams: run what is queued)
        per_render, _ = self.reference()
        N, H, W, L = self.N, self.H, self.W, self.lay
        T = ((H + 15) // 16) * ((W + 15) // 16)
        leader = {}
        for i in range(self.n):
            o, _, skinned = per_render[i]
            s = self.slots[leader.setdefault((self.pair_of[i], self._launch_of(i)), i)]  # the group's leader holds them
            for got, want in zip((s["pts"], s["rot"], s["scales"], s["opac"]), skinned):
                assert np.abs(got - want).mean() <= 1e-5 * max(1.0, np.abs(want).mean())
            assert int(self.totals[i, 0]) == o["R"] and int(self.totals[i, 1]) == 0
            assert np.array_equal(self.slots[i]["radii"], o["radii"])
            b = self.slots[i]["bin"]
            assert np.array_equal(b[L["vals"]:L["vals"] + 4 * o["R"]].view(np.uint32), o["vals_sorted"]), "sorted order differs"
            assert np.array_equal(b[L["ranges"]:L["ranges"] + 8 * T].view(np.uint32).reshape(T, 2), o["ranges"].reshape(T, 2))
            for k, ok in (("color", "out_color"), ("depth", "out_depth"), ("alpha", "out_alpha"), ("normal", "out_normal")):
                assert np.isfinite(self.out[k][i]).all()
                err = np.abs(self.out[k][i] - o[ok]).mean()
                assert err <= L1_TOL, (i, k, err)

    def _launch_of(self, i):
        return self.launch_of[i] if hasattr(self, "launch_of") else 0

    def losses(self, first, count, stream):
        """What the trainer enqueues between a range's forward and its backward, on `stream`: something that reads the
        renders' images and leaves their gradient images."""
        for i in range(first, first + count):
            for k in self.out:
                S_().simt_enqueue_copy(self.seen[k][i].ctypes.data, self.out[k][i].ctypes.data, self.out[k][i].nbytes, stream)
            for live, g in zip(self.gw_live, self.gw):
                S_().simt_enqueue_copy(live[i].ctypes.data, g[i].ctypes.data, g[i].nbytes, stream)

    def check_seen(self):
        S_().simt_synchronize()
        for k in self.out:
            assert np.array_equal(self.seen[k], self.out[k]), "a loss kernel ran before its render had finished: " + k

    def hand_over_dot_planes(self):
        """The optional per-pixel plane S = sum over the channels of gradient x rendered value (what the loss kernels
        emit): handed to the second half of the renders; for the first half the blend backward forms S itself."""
        S_().simt_synchronize()
        o, g = self.out, self.gw
        self.dot = _f32((g[0] * o["color"]).sum(1, keepdims=True) + g[1] * o["depth"]
                        + (g[2] * o["normal"]).sum(1, keepdims=True) + g[3] * o["alpha"])
        for i in range(self.n // 2, self.n):
            self.descs[i].g_dot = self.dot[i].ctypes.data

    def check_raster_gradients(self, renders=None):
        S_().simt_synchronize()
        per_render, _ = self.reference()
        rel = lambda a, b: np.abs(a - b).sum() / (np.abs(b).sum() + 1e-12)
        for i in (range(self.n) if renders is None else renders):
            _, go, _ = per_render[i]
            s = self.slots[i]
            assert rel(s["g_means2D"][:, :2].reshape(-1), go["dL_dmean2D"].reshape(-1)) <= L1_TOL
            assert rel(s["g_shs"].reshape(-1), go["dL_dshs"].reshape(-1)) <= L1_TOL

    def check_accumulated(self):
        S_().simt_synchronize()
        _, acc = self.reference()
        for k, want in acc.items():
            got = self.acc[k].astype(np.float64)
            assert np.isfinite(got).all(), k
            err = np.abs(got - want.reshape(got.shape)).sum() / (np.abs(want).sum() + 1e-12)
            assert err <= 2 * L1_TOL, (k, err)


PAIRS_4 = [0, 0, 1, 1]           # two motions x (one pair, two views)
PAIRS_6 = [0, 0, 1, 2, 2, 3]     # two motions x (a pair seen twice + a pair seen once)


@pytest.fixture(scope="module")
def step4():
    return Step(1200, 24, 64, 48, PAIRS_4, seed=1)


@pytest.fixture(scope="module")
def step6():
    return Step(900, 17, 48, 64, PAIRS_6, seed=2)


def _ranges(st):
    if getattr(st, "ranges", None):
        return st.ranges
    half = st.n // 2
    return [(0, half), (half, st.n - half)]


# How the streams' operations are ordered (tests/simt/runtime.cpp): run at enqueue, or QUEUED per stream and run at the
# next synchronisation one at a time from any stream whose head is ready -- by a seeded draw, youngest stream first
# (the private streams run as far as they can before the caller's), or oldest first.  An edge the executor forgot to
# enqueue lets a consumer run ahead of its producer in at least one of these.
DEFERRED = ["deferred:1", "deferred:2", "deferred:lifo", "deferred:fifo"]
STREAM_ORDERS = ["immediate"] + DEFERRED
SEQUENCES = ["in_order", "in_order_skinned", "in_order_skinned_side", "joint", "launch"]


def run_ranged_step(st, sequence, deferred):
    """One step in the batched-ranges mode with the call sequence of Trainer._forward_backward_direct
    (dimo_amd/trainer.py) named by `sequence`.  In the deferred orders nothing synchronises between the first enqueue
    and the last (the trainer's host never waits for the device inside a step either)."""
    ex = S_().dimo_executor_create(-2)
    assert ex
    c, d = C.addressof(st.common), C.addressof(st.descs)
    # (deformation groups are formed per LAUNCH: a range longer than 8 renders is cut from its start)
    st.launch_of = [next((k, (i - f) // 8) for k, (f, cnt) in enumerate(_ranges(st)) if f <= i < f + cnt) for i in range(st.n)]
    E = S_()
    try:
        for first, count in _ranges(st):
            assert E.dimo_executor_forward_range(ex, c, first, count, d, None) == 0
        if not deferred:
            st.check_forward()
            st.hand_over_dot_planes()
        own = sequence != "launch"  # a motion's losses on the stream its chain runs on, or on the caller's stream
        if not own:
            assert E.dimo_executor_join(ex, 0, st.n, None) == 0
        for first, count in _ranges(st):
            st.losses(first, count, E.dimo_executor_range_stream(ex, first) if own else None)
            if sequence == "launch":
                assert E.dimo_executor_backward_launch(ex, c, first, count, d, None) == 0
            elif sequence != "joint":
                assert E.dimo_executor_backward_launch_in_order(ex, c, first, count, d, None) == 0
                if sequence.startswith("in_order_skinned"):  # the default: ... and its skinning backward behind it
                    assert E.dimo_executor_backward_skinning_in_order(ex, c, first, count, d, None) == 0
        if sequence == "joint":  # ONE blend backward over all the step's renders, then skinning backward + fold
            assert E.dimo_executor_backward_launch_joint(ex, c, 0, st.n, d, None) == 0
            assert E.dimo_executor_backward_accumulate(ex, c, 0, st.n, d, None) == 0
        elif sequence == "in_order_skinned":
            # the caller's stream joins the ranges' streams (round 6: the skinning backward adds its sums to the shared
            # gradients itself, the call has no kernel of its own any more) ...
            assert E.dimo_executor_backward_accumulate(ex, c, 0, st.n, d, None) == 0
            # ... and what the optimizer is to the executor follows: a reader of the gradient bucket on that stream
            st.bucket_seen = np.full_like(st.acc["xyz"], np.nan)
            E.simt_enqueue_copy(st.bucket_seen.ctypes.data, st.acc["xyz"].ctypes.data, st.acc["xyz"].nbytes, None)
        elif sequence == "in_order_skinned_side":
            # ... on private stream 0 (followed there by the optimizer's early launch, next to the TimeNet backward on
            # the caller's stream, which needs the ranges' TimeNet-row gradients: join_ranges)
            side = E.dimo_executor_private_stream(ex, 0)
            assert side
            assert E.dimo_executor_join_ranges(ex, 0, st.n, None) == 0
            # (what the TimeNet backward is to the executor: a reader of the rows' gradients on the caller's stream)
            rows = st.acc["d_xyz"]
            st.rows_seen = np.full_like(rows, np.nan)
            E.simt_enqueue_copy(st.rows_seen.ctypes.data, rows.ctypes.data, rows.nbytes, None)
            assert E.dimo_executor_backward_accumulate(ex, c, 0, st.n, d, side) == 0
            assert E.dimo_executor_side_done(ex, 0) == 0
            assert E.dimo_executor_wait_side(ex, 0, None) == 0
        else:  # the skinning backward per motion on the caller's stream
            for first, count in _ranges(st):
                assert E.dimo_executor_backward_accumulate(ex, c, first, count, d, None) == 0
        st.check_seen()
        st.check_forward()
        st.check_raster_gradients()
        st.check_accumulated()
        if sequence == "in_order_skinned_side":
            assert np.array_equal(st.rows_seen, st.acc["d_xyz"]), "the caller's stream read the TimeNet rows' gradients early"
        if sequence == "in_order_skinned":
            assert np.array_equal(st.bucket_seen, st.acc["xyz"]), "the caller's stream read the gradient bucket early"
    finally:
        E.simt_synchronize()
        E.dimo_executor_destroy(ex)
        del st.launch_of


def test_emulated_executor_adaptive_chains():
    """The batched forward picks the backward's chain length (buckets of 64 list entries per work item) per render slot
    from what the slot's PREVIOUS forward left in its bin workspace (blend.hip: adaptive chains).  Deep lists -- every
    Gaussian over most of a small image at opacity 0.01-0.05, ~11 buckets per tile, SURVEY 8d's "init" regime in
    miniature -- ask for four: the first step over fresh workspaces runs one bucket per item and leaves 4, the second
    and third run four; every step must match the oracle, and shallow lists take the length back to one."""
    st = Step(700, 12, 48, 32, PAIRS_4, seed=5, scale=0.25, opacity=(0.01, 0.05))
    meta = hz._layouts(st.N, st.H, st.W, st.r_cap)[1]["meta"]
    word = lambda i: int(st._ws[i]["bin"][meta + 16:meta + 20].view(np.uint32)[0])
    for rnd, sequence in enumerate(("in_order_skinned", "joint", "in_order")):
        st.fresh(keep_bin=rnd > 0)
        run_ranged_step(st, sequence, False)
        assert [word(i) for i in range(st.n)] == [4] * st.n
    # the same slots then serve a model with short lists: the first step still runs chains of four (correctly), and
    # leaves 1
    shallow = Step(700, 12, 48, 32, PAIRS_4, seed=7, scale=0.006)
    shallow._ws = st._ws
    shallow.fresh(keep_bin=True)
    run_ranged_step(shallow, "in_order_skinned", False)
    assert [word(i) for i in range(shallow.n)] == [1] * shallow.n


@pytest.mark.parametrize("streams", STREAM_ORDERS)
@pytest.mark.parametrize("sequence", SEQUENCES)
@pytest.mark.parametrize("which", ["step4", "step6"])
def test_emulated_executor_batched_ranges(which, sequence, streams, request, monkeypatch):
    """n_streams < 0 (the benchmark's mode): a range per motion on a private stream; the trainer's backward call
    sequences, each under every ordering of the streams' operations; cross-stream dependencies through stream
    write / wait values (the default) and, for `launch`, through events."""
    st = request.getfixturevalue(which)
    if streams != "immediate" and which == "step6" and (sequence != "joint" or streams in ("deferred:2", "deferred:fifo")):
        pytest.skip("covered by step4")
    st.fresh()
    monkeypatch.setenv("SIMT_STREAMS", streams)
    if sequence == "launch":
        monkeypatch.setenv("DIMO_XSTREAM", "event")
    run_ranged_step(st, sequence, streams != "immediate")


@pytest.mark.parametrize("lengths,sequence,streams", [
    ((10,), "joint", "immediate"),              # a range longer than a launch: cut 8 + 2 like its forward
    ((10,), "in_order_skinned", "deferred:lifo"),
    ((9, 2), "joint", "deferred:2"),            # ... followed by a short one
    ((3, 3, 3), "joint", "deferred:fifo"),      # three motions on two streams: 3 + 3 merge into a launch, the third alone
    ((3, 3, 3), "in_order_skinned_side", "deferred:1"),
    ((1, 1), "in_order", "deferred:lifo"),      # one render per motion
    ((6, 6), "joint", "immediate"),             # two ranges that do not fit one launch together
    ((5, 1, 4), "launch", "deferred:3"),
])
def test_emulated_executor_range_shapes(lengths, sequence, streams, monkeypatch):
    """How the executor cuts ranges into launches (plan_chunks: whole ranges merged while they fit eight renders, a
    longer range cut from its start exactly like its forward, deformation groups formed per launch) -- range shapes
    beyond the benchmark's two motions of four."""
    pair_of, ranges, q = [], [], 0
    for n in lengths:
        ranges.append((len(pair_of), n))
        for j in range(n):
            pair_of.append(q + j // 2)  # two views per (motion, frame) pair, the last pair of an odd range alone
        q += (n + 1) // 2
    st = Step(300, 9, 32, 48, pair_of, seed=sum(lengths))
    st.ranges = ranges
    monkeypatch.setenv("SIMT_STREAMS", streams)
    if sequence == "launch":
        monkeypatch.setenv("DIMO_XSTREAM", "event")
    run_ranged_step(st, sequence, streams != "immediate")


def test_emulated_executor_instance_capacity_overflow():
    """Workspaces sized for fewer tile instances than the renders produce (the capacity policy runs a step ahead of
    the device's counts): every render flags the overflow in its (R, overflow) words -- the words that turn the step's
    Adam update into a no-op --, nothing is written out of bounds (a stray write into host memory here is a crash or a
    corrupted neighbour, not something a GPU forgives) and every accumulated gradient stays finite."""
    st = Step(1200, 24, 64, 48, PAIRS_4, seed=1, r_cap=3000)
    st.fresh()
    ex = S_().dimo_executor_create(-2)
    c, d = C.addressof(st.common), C.addressof(st.descs)
    for first, count in _ranges(st):
        assert S_().dimo_executor_forward_range(ex, c, first, count, d, None) == 0
    st.losses(0, st.n, None)
    assert S_().dimo_executor_backward_launch_joint(ex, c, 0, st.n, d, None) == 0
    assert S_().dimo_executor_backward_accumulate(ex, c, 0, st.n, d, None) == 0
    S_().simt_synchronize()
    assert (st.totals[:, 0] > 3000).all() and (st.totals[:, 1] == 1).all()
    for k, v in st.out.items():
        assert np.isfinite(v).all(), k
    for k, v in st.acc.items():
        assert np.isfinite(v).all(), k
    S_().dimo_executor_destroy(ex)


@pytest.fixture(scope="module")
def step_s1():
    return Step(400, 8, 48, 48, PAIRS_6, seed=6, stage1=True)


@pytest.mark.parametrize("sequence,streams", [("in_order_skinned", "immediate"), ("in_order_skinned", "deferred:lifo"),
                                              ("joint", "deferred:5"), ("joint", "deferred:fifo")])
def test_emulated_executor_stage_s1(step_s1, sequence, streams, monkeypatch):
    """Stage s1 on the batched executor: the TimeNet's rows move the Gaussians themselves, scales = exp of the one
    shared log-radius (renderer/latent_gs_renderer.py:1176-1177, 1211-1212, 341-351): s1_fwd / s1_bwd_batched_kernel,
    the radius gradient included."""
    step_s1.fresh()
    monkeypatch.setenv("SIMT_STREAMS", streams)
    run_ranged_step(step_s1, sequence, streams != "immediate")


@pytest.mark.parametrize("which", ["step4", "step6"])
def test_emulated_executor_one_launch_per_stage(which, request):
    """n_streams = 0: every stage ONE launch over all the renders of the call, on the caller's stream."""
    st = request.getfixturevalue(which)
    st.fresh()
    ex = S_().dimo_executor_create(0)
    c, d = C.addressof(st.common), C.addressof(st.descs)
    assert S_().dimo_executor_forward(ex, c, st.n, d, None) == 0
    st.check_forward()
    st.hand_over_dot_planes()
    st.losses(0, st.n, None)
    assert S_().dimo_executor_backward_launch(ex, c, 0, st.n, d, None) == 0
    st.check_raster_gradients()
    assert S_().dimo_executor_backward_accumulate(ex, c, 0, st.n, d, None) == 0
    st.check_accumulated()
    S_().dimo_executor_destroy(ex)


@pytest.mark.parametrize("streams", ["immediate", "deferred:3", "deferred:lifo", "deferred:fifo"])
def test_emulated_executor_per_render_chains(step4, streams, monkeypatch):
    """n_streams > 0: every render its own chain of the single-render entry points, renders round-robin over two
    private streams, events between them and the caller's stream."""
    st = step4
    st.fresh()
    monkeypatch.setenv("SIMT_STREAMS", streams)
    st.launch_of = list(range(st.n))  # no deformation groups: every slot holds its own skinned Gaussians
    ex = S_().dimo_executor_create(2)
    c, d = C.addressof(st.common), C.addressof(st.descs)
    try:
        assert S_().dimo_executor_forward(ex, c, st.n, d, None) == 0
        assert S_().dimo_executor_join(ex, 0, st.n, None) == 0
        st.losses(0, st.n, None)
        assert S_().dimo_executor_backward_launch(ex, c, 0, st.n, d, None) == 0
        assert S_().dimo_executor_backward_accumulate(ex, c, 0, st.n, d, None) == 0
        S_().simt_synchronize()
        st.totals[:, 0] = [r[0]["R"] for r in st.reference()[0]]  # (this mode does not write totals_out)
        st.check_seen()
        st.check_forward()
        st.check_raster_gradients()
        st.check_accumulated()
    finally:
        S_().simt_synchronize()
        S_().dimo_executor_destroy(ex)
        del st.launch_of


def test_emulated_executor_refuses_what_the_product_refuses(step4):
    st = step4
    st.fresh()
    ex = S_().dimo_executor_create(-2)
    c, d = C.addressof(st.common), C.addressof(st.descs)
    assert S_().dimo_executor_create(17) is None
    assert S_().dimo_executor_forward_range(ex, c, -1, 2, d, None) != 0
    assert S_().dimo_executor_forward_range(ex, None, 0, 2, d, None) != 0
    assert S_().dimo_executor_forward_range(ex, c, 0, 0, d, None) == 0
    assert S_().dimo_executor_forward_range(ex, c, 0, 2, d, None) == 0
    # a backward over a range that was never started as one
    assert S_().dimo_executor_backward_launch_in_order(ex, c, 1, 1, d, None) != 0
    assert S_().dimo_executor_backward_skinning_in_order(ex, c, 0, 1, d, None) != 0  # not the range's length
    S_().dimo_executor_destroy(ex)


@pytest.mark.parametrize("variant,sequence", [("step_no_accumulate_wait", "in_order_skinned"),
                                              ("step_no_joint_wait", "joint"), ("step_no_backward_fork", "launch")])
def test_the_deferred_stream_orders_catch_a_missing_dependency(variant, sequence, monkeypatch):
    """The detector's own test.  tests/simt/build.py builds the executor three more times, each with ONE cross-stream
    dependency taken out of executor.hip -- the fold's wait for the ranges' backward, the joint backward's wait for the
    ranges' forward (and losses), the fork of a range's backward from the caller's stream (the losses there) -- and
    every one of them must come out WRONG under at least one deferred order, while run at enqueue (as a launch-order
    emulation would) each still passes: that is what the deferred orders add."""
    global _CURRENT
    st = Step(500, 12, 48, 48, PAIRS_4, seed=4)
    st.reference()
    monkeypatch.setenv("DIMO_XSTREAM", "event" if sequence == "launch" else "value")
    _CURRENT = variant
    try:
        monkeypatch.setenv("SIMT_STREAMS", "immediate")
        st.fresh(fill=0)
        run_ranged_step(st, sequence, True)  # (in enqueue order the missing edge goes unnoticed)
        caught = []
        for order in DEFERRED:
            monkeypatch.setenv("SIMT_STREAMS", order)
            st.fresh(fill=0)  # (zeroed workspaces: a consumer that runs too early sees empty lists, not wild indices)
            try:
                run_ranged_step(st, sequence, True)
            except AssertionError:
                caught.append(order)
        assert caught, "no deferred order exposed the missing dependency"
    finally:
        _CURRENT = "step"
