"""GPU parity of the IRN random-walk propagation (src/indexing.py::propagate_to_edge) against the
fixture produced by the reference's own functions and, at a larger size, against the oracle.  fp32; the walk is
exp_times matrix squarings, so the stated tolerance is 2e-4 of the output maximum."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_propagate_to_edge_golden():
    from muscle_amd import indexing
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "irn_rw.npz"))
    for tag in ("a", "b", "c"):
        radius, beta, times = (int(v) for v in z[f"{tag}_params"])
        x, edge, ref = (torch.from_numpy(z[f"{tag}_{k}"]) for k in ("x", "edge", "rw"))
        rw = indexing.propagate_to_edge(x.to(DEV), edge.to(DEV), radius=radius, beta=beta, exp_times=times).cpu()
        assert tuple(rw.shape) == tuple(ref.shape)
        err = float((rw - ref).abs().max()) / float(ref.abs().max())
        assert err <= 2e-4, (tag, err)


def test_propagate_to_edge_vs_oracle_larger():
    from oracle import mcl_oracle as O
    from muscle_amd import indexing, synth
    h, w = 31, 45                                   # n = 1395 (not a multiple of 4 -> padded operands)
    x = torch.from_numpy(synth.uniform(7, "irn_x", (1, 20, h, w)).astype(np.float32))
    edge = torch.from_numpy(synth.uniform(7, "irn_e", (1, h, w)).astype(np.float32)) ** 3
    ref = O.irn_propagate_to_edge(x, edge, 5, 10, 8)
    rw = indexing.propagate_to_edge(x.to(DEV), edge.to(DEV), radius=5, beta=10, exp_times=8).cpu()
    err = float((rw - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-4, err
    # a transition matrix is column-stochastic: propagating an all-ones map with edge = 0 returns all ones
    one = indexing.propagate_to_edge(torch.ones(1, 1, h, w, device=DEV), torch.zeros(1, h, w, device=DEV), exp_times=3)
    assert float((one - 1).abs().max()) <= 1e-4


def test_finish_semseg_matches_oracle():
    from oracle import mcl_oracle as O
    from muscle_amd import indexing, synth
    h, w, H, W = 19, 23, 74, 90                       # (H, W) crops the 4x upsampled 76 x 92 maps, as orig_img_size does
    rw = torch.from_numpy(synth.uniform(9, "rw", (20, 1, h, w)).astype(np.float32)) ** 2
    rw[3] *= 0.0                                       # an absent class
    lab_ref, soft_ref = O.irn_finish(rw, H, W, 0.25)
    lab, soft = indexing.finish_semseg(rw.to(DEV), H, W, 0.25, soft_output=True)
    lab, soft = lab.cpu().numpy(), soft.cpu().numpy()
    assert lab.dtype == np.uint8 and lab.shape == (H, W) and soft.dtype == np.float16 and soft.shape == (H, W, 21)
    assert float(np.abs(soft.astype(np.float32) - soft_ref.astype(np.float32)).max()) <= 1e-3      # one fp16 ulp near 1.0
    # labels: identical except where the two best values are within fp32 round-off of each other
    diff = lab != lab_ref
    assert diff.mean() <= 2e-3, float(diff.mean())
