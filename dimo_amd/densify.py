"""Densification, pruning and optimizer surgery of the canonical Gaussians -- the reference's
`GaussianModel.densify_and_prune / densify_and_clone / densify_and_split / prune / prune_points / reset_opacity /
add_densification_stats` (renderer/latent_gs_renderer.py:571-574,652-924), with the same semantics including their
quirks (documented where they matter), on the flat parameter bucket.

MI355X-first difference.  The reference rebuilds six `nn.Parameter`s and their Adam state tensor by tensor, three
times per `densify_and_prune` (clone, split, prune).  Here every operation only edits a small *plan* -- for each
surviving row of the new model the old row it comes from, whether its Adam moments start at zero, and the few
values that are overwritten -- and the flat parameter / gradient / moment buckets are rebuilt ONCE at the end by
gathers (`GaussianModel.rebuild`).  The Adam step counter carries over, as `cat_tensors_to_optimizer` keeps it.

Replica consistency: the only random draw (`densify_and_split`) uses the global torch generator of the model's
device, exactly like the reference's `torch.normal`; under data parallelism `Trainer` sets `split_generator`, seeded
from (seed, step), so that every rank draws the same samples whatever else consumed its global generator.
"""
import torch

from .deform import build_rotation

PER_GAUSSIAN = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def morton_order(xyz, bits=10):
    """Permutation that sorts points along the 3D Morton (Z-order) curve of their bounding box; stable, so equal
    codes keep their current order and replicas holding equal positions get equal permutations."""
    lo, hi = xyz.min(dim=0).values, xyz.max(dim=0).values
    q = ((xyz - lo) / (hi - lo).clamp_min(1e-12) * float((1 << bits) - 1)).to(torch.int64).clamp_(0, (1 << bits) - 1)
    code = torch.zeros(xyz.shape[0], dtype=torch.int64, device=xyz.device)
    for b in range(bits):
        for d in range(3):
            code |= ((q[:, d] >> b) & 1) << (3 * b + d)
    return torch.sort(code, stable=True).indices


class _Plan:
    """Rows of the model being built: `src` = source row in the CURRENT model, `fresh` = Adam moments start at
    zero, `values` = name -> tensor [rows, ...] of the parameter values (gathered lazily)."""

    def __init__(self, model):
        self.m = model
        n = model._xyz.shape[0]
        dev = model._xyz.device
        self.src = torch.arange(n, device=dev)
        self.fresh = torch.zeros(n, dtype=torch.bool, device=dev)
        self.values = {k: p.detach() for k, p in model.per_gaussian().items()}

    @property
    def n(self):
        return self.src.shape[0]

    def append(self, src_rows, new_values):
        """Appends rows copied from the plan's rows `src_rows` (moments zero) with the given values."""
        self.src = torch.cat([self.src, self.src[src_rows]])
        self.fresh = torch.cat([self.fresh, torch.ones(src_rows.shape[0], dtype=torch.bool, device=self.src.device)])
        for k in self.values:
            self.values[k] = torch.cat([self.values[k], new_values[k]], dim=0)

    def keep(self, mask):
        self.src, self.fresh = self.src[mask], self.fresh[mask]
        for k in self.values:
            self.values[k] = self.values[k][mask]

    def scaling_act(self):
        """`GaussianModel.get_scaling` on the rows of the plan (latent_gs_renderer.py:341-351): exp(_scaling) once
        `_r` is empty (stage s2), otherwise exp of the radius `_r` -- shared (1, 1) in stage s1, or one row per
        Gaussian -- broadcast to three axes."""
        r = self.m._r
        if len(r) == 0:
            return torch.exp(self.values["scaling"])
        if "r" in self.values:
            rv = self.values["r"]
            return torch.exp(rv.repeat(1, 3) if rv.shape[1] == 1 else rv)
        return torch.exp(r.detach().reshape(1, -1)[:, :1].repeat(self.n, 3))

    def opacity_act(self):
        return torch.sigmoid(self.values["opacity"])


class DensifyMixin:
    """Mixed into `GaussianModel`."""

    # ------------------------------------------------------------------ statistics
    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """latent_gs_renderer.py:922-924.  `viewspace_point_tensor`: a tensor with `.grad` [N, 3] (the autograd
        path) or the gradient tensor itself (the direct pipeline's `g_means2D`)."""
        g = getattr(viewspace_point_tensor, "grad", None)
        if g is None:
            g = viewspace_point_tensor
        self.xyz_gradient_accum[update_filter] += torch.norm(g[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    def update_max_radii(self, radii, visibility_filter):
        """main_train_dimo.py:431."""
        self.max_radii2D[visibility_filter] = torch.max(self.max_radii2D[visibility_filter],
                                                        radii[visibility_filter].to(self.max_radii2D.dtype))

    # ------------------------------------------------------------------ plan steps (no parameter is touched)
    def _plan_clone(self, plan, grads, grad_threshold, scene_extent):
        """latent_gs_renderer.py:856-874."""
        sel = torch.norm(grads, dim=-1) >= grad_threshold
        sel = torch.logical_and(sel, torch.max(plan.scaling_act(), dim=1).values <= self.percent_dense * scene_extent)
        rows = sel.nonzero(as_tuple=True)[0]
        plan.append(rows, {k: v[rows] for k, v in plan.values.items()})  # (a per-point `_r` is cloned like the rest)

    def _plan_split(self, plan, grads, grad_threshold, scene_extent, N=2):
        """latent_gs_renderer.py:826-854 (grads are those of BEFORE the clone, zero-padded for the clones)."""
        n_init = plan.n
        padded = torch.zeros(n_init, device=grads.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = padded >= grad_threshold
        sel = torch.logical_and(sel, torch.max(plan.scaling_act(), dim=1).values > self.percent_dense * scene_extent)
        rows = sel.nonzero(as_tuple=True)[0]
        sc = plan.scaling_act()[rows]
        stds = sc.repeat(N, 1)
        means = torch.zeros((stds.size(0), 3), device=stds.device)
        gen = getattr(self, "split_generator", None)  # data-parallel ranks: Trainer supplies a rank-identical one
        samples = torch.normal(mean=means, std=stds) if gen is None else torch.normal(mean=means, std=stds, generator=gen)
        rots = build_rotation(plan.values["rotation"][rows]).repeat(N, 1, 1)
        new = {
            "xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + plan.values["xyz"][rows].repeat(N, 1),
            "scaling": torch.log(sc.repeat(N, 1) / (0.8 * N)),
            "rotation": plan.values["rotation"][rows].repeat(N, 1),
            "f_dc": plan.values["f_dc"][rows].repeat(N, 1, 1),
            "f_rest": plan.values["f_rest"][rows].repeat(N, 1, 1),
            "opacity": plan.values["opacity"][rows].repeat(N, 1),
        }
        if "r" in plan.values:  # new_r = new_scaling[:, :r.shape[1]] (latent_gs_renderer.py:846-848)
            new["r"] = new["scaling"][:, :plan.values["r"].shape[1]]
        plan.append(rows.repeat(N), new)
        keep = torch.cat([~sel, torch.ones(N * rows.shape[0], dtype=torch.bool, device=sel.device)])
        plan.keep(keep)

    def _plan_prune(self, plan, min_opacity, extent, max_screen_size, max_radii2D):
        """latent_gs_renderer.py:883-889 / 892-900."""
        mask = (plan.opacity_act() < min_opacity).squeeze(-1)
        if max_screen_size:
            big_vs = max_radii2D > max_screen_size
            big_ws = plan.scaling_act().max(dim=1).values > 0.1 * extent
            mask = torch.logical_or(torch.logical_or(mask, big_vs), big_ws)
        plan.keep(~mask)
        return mask

    # ------------------------------------------------------------------ the reference's entry points
    @torch.no_grad()
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        """latent_gs_renderer.py:876-890.  Quirk kept: `densification_postfix` zeroes `max_radii2D` (and the
        gradient statistics) when the clones are appended, so the screen-size criterion of the final prune never
        fires here -- only the opacity and world-size criteria do."""
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        plan = _Plan(self)
        self._plan_clone(plan, grads, max_grad, extent)
        self._plan_split(plan, grads, max_grad, extent)
        self._plan_prune(plan, min_opacity, extent, max_screen_size, torch.zeros(plan.n, device=plan.src.device))
        self.rebuild(plan)
        n = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        self.max_radii2D = torch.zeros(n, device=self.device)

    @torch.no_grad()
    def prune(self, min_opacity, extent, max_screen_size=None):
        """latent_gs_renderer.py:892-902 (stage s2, main_train_dimo.py:439-443)."""
        plan = _Plan(self)
        self._plan_prune(plan, min_opacity, extent, max_screen_size, self.max_radii2D)
        self._finish_prune(plan)

    @torch.no_grad()
    def prune_points(self, mask):
        """latent_gs_renderer.py:719-733: removes the rows where `mask` is True."""
        plan = _Plan(self)
        plan.keep(~mask)
        self._finish_prune(plan)

    @torch.no_grad()
    def prune_s1_end(self, min_opacity, extent=None, max_screen_size=None):
        """latent_gs_renderer.py:914-919 / 748-766 (end of stage s1, main_train_dimo.py:200): removes the Gaussians
        below `min_opacity` AND the same rows of the control points -- at that point both have `num_cpts` rows (the
        last FPS ran after the density window closed) and stage s2 is about to copy the Gaussians' positions into
        the control points (`prepare_train_s2`).  Like the reference, only the opacity criterion is applied."""
        if self._c_xyz.shape[0] != self._xyz.shape[0]:
            raise ValueError("prune_s1_end: the control points must have one row per Gaussian "
                             f"({self._c_xyz.shape[0]} vs {self._xyz.shape[0]})")
        plan = _Plan(self)
        mask = (plan.opacity_act() < min_opacity).squeeze(-1)
        plan.keep(~mask)
        self._finish_prune(plan, ctrl_keep=~mask)

    def _finish_prune(self, plan, ctrl_keep=None):
        src = plan.src
        self.rebuild(plan, ctrl_keep=ctrl_keep)
        self.xyz_gradient_accum = self.xyz_gradient_accum[src]
        self.denom = self.denom[src]
        self.max_radii2D = self.max_radii2D[src]

    @torch.no_grad()
    def sort_spatially(self):
        """MI355X layout step with no counterpart in the reference (a Gaussian model is a set): stores the Gaussians
        in Morton order of their canonical positions, so that the 64 Gaussians of a wavefront share control points
        (the skinning backward's LDS atomics combine inside the wave), reach the same screen tiles (binning) and sit
        in neighbouring cache lines when the blend kernels gather them.  Parameters, Adam moments and densification
        statistics move together; returns the permutation (new row -> old row)."""
        perm = morton_order(self._xyz.detach())
        if self.optimizer is None:
            self.flush_pending_renders()
            for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation") + \
                    (("_r",) if self.r_is_per_point() else ()):
                p = getattr(self, name)
                setattr(self, name, torch.nn.Parameter(p.detach()[perm].contiguous().requires_grad_(True)))
            for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
                t = getattr(self, name, None)
                if t is not None and t.shape[0] == perm.shape[0]:
                    setattr(self, name, t[perm])
            self.neighbor_dists = self.neighbor_indices = None
            return perm
        plan = _Plan(self)
        plan.keep(perm)
        self._finish_prune(plan)
        return perm

    @torch.no_grad()
    def reset_opacity(self):
        """latent_gs_renderer.py:571-574: opacity = min(opacity, 0.01), its Adam moments zeroed."""
        plan = _Plan(self)
        op = plan.opacity_act()
        plan.values["opacity"] = self.inverse_opacity_activation(torch.min(op, torch.ones_like(op) * 0.01))
        self.rebuild(plan, zero_moments=("opacity",))
