"""Run-to-run reproducibility of one training step's gradient bucket ("deterministic-reduction
mode for gradient accumulation so parity runs are reproducible").  What the HIP pipeline guarantees, kernel by kernel:

  * bit-reproducible: projection, binning (the tile lists are a pure function of the keys), blend forward / backward
    (one wave owns a record: no atomics), the per-instance gradient records and their per-Gaussian sums (fixed order),
    the skinning backward's per-Gaussian outputs and the fold into the gradient views (fixed render order) -- i.e. the
    whole per-Gaussian HEAD of the flat bucket (xyz, colour, opacity, scaling, rotation): asserted bit for bit;
  * order-dependent (fp32 atomics): the skinning backward's control-point scatter (LDS float atomics inside a
    workgroup; since round 6 the workgroups' sums are added with global fp32 atomics too) and the TimeNet weight gradients
    (split-K partial products added with hardware fp32 atomics) -- and everything downstream of the first: control
    points, radii, the TimeNet's output-row gradients, latents.  These differ run to run by rounding only: asserted
    within 1e-5 of each group's largest magnitude.  Data-parallel replicas are unaffected: every rank applies the
    same all-reduced bucket (tests/test_gpu_trains.py holds two replicas bit-identical through a whole schedule).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_same_step_twice_head_bit_identical_tail_to_rounding():
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=20000, num_cpts=128, num_motions=6, resolution=128, motions_per_step=2,
                      views_per_step=2, frames_per_step=2)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                  capacity=CapacityPolicy(initial=1 << 21))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=2, num_latent=cfg.num_motions)
    tr = Trainer(cfg, rd)
    assert tr.direct
    tr.step = 1000
    for _ in range(3):
        tr.train_step()
    g = rd.gaussians
    triples = tr.sample()
    real_step = tr.optimizer.step
    tr.optimizer.step = lambda *a, **k: None  # same parameters every time; the gradients stay in the bucket
    grads, losses = [], []
    for _ in range(4):
        g.zero_grad()
        tr.step = 1000
        tr.train_step(triples)
        torch.cuda.synchronize()
        grads.append(g.flat_grads.detach().clone())
        losses.append(float(tr.last_loss))
    tr.optimizer.step = real_step
    split = g.flat_split
    assert split > 0 and float(grads[0][:split].abs().max()) > 0
    for other in grads[1:]:
        assert torch.equal(grads[0][:split], other[:split]), "the per-Gaussian head of the bucket must be reproducible"
    base = g.flat_params.data_ptr()
    for grp in tr.optimizer.param_groups:
        lo = min((p.data_ptr() - base) // 4 for p in grp["params"])
        hi = max((p.data_ptr() - base) // 4 + p.numel() for p in grp["params"])
        if hi <= split:
            continue
        ref = grads[0][lo:hi]
        scale = float(ref.abs().max())
        for other in grads[1:]:
            assert float((other[lo:hi] - ref).abs().max()) <= 1e-5 * max(scale, 1e-12), grp["name"]
    assert max(losses) - min(losses) <= 1e-6 * abs(losses[0])


def test_side_stream_fold_against_single_stream_at_large_n():
    """the per-Gaussian fold + Adam head run on a private stream NEXT TO the TimeNet backward,
    whose embedding backward adds its input gradient to `_c_xyz.grad` atomically -- the fold's control-point sums must
    not be a plain read-modify-write of the same words.  Many Gaussians (a long fold), small images (short renders):
    the two writers overlap; the control-point gradients must equal the serial schedule's (`Trainer._split_adam = False`:
    fold and ONE Adam launch on the caller's stream) up to the rounding of their atomics, step after step."""
    from dimo_amd.rasterizer import CapacityPolicy
    from dimo_amd.renderer import Renderer
    from dimo_amd.synth import init_synthetic_model
    from dimo_amd.trainer import TrainConfig, Trainer
    cfg = TrainConfig(num_pts=400000, num_cpts=512, num_motions=6, resolution=64, motions_per_step=2,
                      views_per_step=2, frames_per_step=2)
    rd = Renderer(sh_degree=0, num_latent_code=cfg.num_motions, add_normal=True, device="cuda",
                  capacity=CapacityPolicy(initial=1 << 22))
    init_synthetic_model(rd, cfg.num_pts, cfg.num_cpts, seed=5, num_latent=cfg.num_motions)
    tr = Trainer(cfg, rd)
    assert tr.direct and tr._split_adam
    tr.step = 1000
    tr.train_step()
    g = rd.gaussians
    triples = tr.sample()
    real_step = tr.optimizer.step
    tr.optimizer.step = lambda *a, **k: None

    def grads(split_adam):
        tr._split_adam = split_adam
        g.zero_grad()
        tr.step = 1000
        tr.train_step(triples)
        torch.cuda.synchronize()
        return g._c_xyz.grad.detach().clone(), g._c_radius.grad.detach().clone(), g.flat_grads.detach().clone()

    ref_c, ref_r, ref_all = grads(False)
    assert float(ref_c.abs().max()) > 0
    for _ in range(6):
        got_c, got_r, got_all = grads(True)
        assert float((got_c - ref_c).abs().max()) <= 1e-5 * float(ref_c.abs().max())
        assert float((got_r - ref_r).abs().max()) <= 1e-5 * float(ref_r.abs().max())
        assert torch.equal(got_all[:g.flat_split], ref_all[:g.flat_split])
    tr.optimizer.step = real_step
    tr._split_adam = True
