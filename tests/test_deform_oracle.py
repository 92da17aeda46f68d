"""Pins oracle/deform_ref.py against the reference-generated fixture (CPU)."""
import os

import numpy as np
import torch

from oracle.deform_ref import skinning_ref
from tests.scenes import timenet_weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def fixture_inputs(name, dtype=torch.float32):
    """Inputs of the skinning block as the reference computed them: parameters + its TimeNet outputs."""
    from dimo_amd.deform import TimeNet
    z = np.load(os.path.join(GOLD, name))
    T = lambda k: torch.from_numpy(z[k]).to(dtype)
    net = TimeNet(device="cpu")
    net.load_state_dict({k: torch.from_numpy(v)
                         for k, v in timenet_weights(int(z["weight_seed"]), float(z["head_std"])).items()})
    lat = T("param.latent_codes")[int(z["latent_index"])]
    d_xyz, d_rot = net(T("param.c_xyz").float(), float(z["time"]), lat.float())
    return z, dict(xyz=T("param.xyz"), rotation=T("param.rotation"), scaling=T("param.scaling"),
                   opacity=T("param.opacity"), c_xyz=T("param.c_xyz"), c_log_radius=T("param.c_radius"),
                   d_xyz=d_xyz.detach().to(dtype), d_rot=d_rot.detach().to(dtype), nn_dist=T("knn_dist"),
                   nn_idx=torch.from_numpy(z["knn_idx"]))


def test_skinning_ref_matches_reference_render():
    z, a = fixture_inputs("deform_latent.npz")
    pts, rot, scales, opac = skinning_ref(**a)
    np.testing.assert_allclose(pts.numpy(), z["s2.in.means3D"], atol=2e-6)
    np.testing.assert_allclose(rot.numpy(), z["s2.in.rotations"], atol=2e-6)
    np.testing.assert_allclose(scales.numpy(), z["s2.in.scales"], atol=1e-7)
    np.testing.assert_allclose(opac.numpy(), z["s2.in.opacities"], atol=1e-7)


def test_skinning_ref_gradients_match_reference():
    z, a = fixture_inputs("deform_latent.npz")
    fixed = ("nn_dist", "nn_idx", "d_xyz", "d_rot")
    leaves = {k: v.clone().requires_grad_(True) for k, v in a.items() if k not in fixed}
    outs = skinning_ref(**leaves, **{k: a[k] for k in fixed})
    loss = sum((o * torch.from_numpy(z[f"s2.w.{k}"])).sum()
               for o, k in zip(outs, ("means3D", "rotations", "scales", "opacities")))
    loss.backward()
    # parameters whose only path to the loss is the skinning block (c_xyz / latents also feed TimeNet)
    for k, gk in (("xyz", "xyz"), ("rotation", "rotation"), ("scaling", "scaling"), ("opacity", "opacity"),
                  ("c_log_radius", "c_radius")):
        ref = z[f"s2.grad.{gk}"]
        np.testing.assert_allclose(leaves[k].grad.numpy(), ref, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)
