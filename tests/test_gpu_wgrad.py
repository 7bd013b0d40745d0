"""Small-output weight gradient (csrc/wgrad.hip) against an fp64 reference on the B7 first-stage shapes (scaled-down row
counts would not take the kernel: it needs R >= 65536), with the operand prologue, and its run-to-run determinism."""
import numpy as np
import pytest
import torch

from muscle_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("R,Co,Ci,mode", [(100352, 288, 48, "plain"), (100352, 48, 288, "bnact"), (131072, 32, 32, "bnact"),
                                          (90000, 64, 28, "plain"), (80000, 192, 32, "plain"), (66000 + 7, 48, 192, "bnact"),
                                          (99999, 32, 64, "bnact"), (131072, 32, 32, "plain")])
def test_wgrad_small_matches_fp64_and_is_deterministic(R, Co, Ci, mode):
    from muscle_amd import ops
    from muscle_amd._lib import lib
    assert lib().mx_pw_wgrad_small_ws(R, Co, Ci, 1 if mode == "bnact" else 0) > 0
    g = torch.Generator(device=DEV).manual_seed(R + Co)
    G = torch.randn(R, Co, device=DEV, generator=g)
    X = torch.randn(R, Ci, device=DEV, generator=g)
    kw = {}
    Xr = X.double()
    if mode == "bnact":
        rps = 1000                                        # rows per sample: the gate changes inside slabs
        sc = torch.rand(Ci, device=DEV, generator=g) + 0.5
        sh = torch.randn(Ci, device=DEV, generator=g) * 0.3
        gate = torch.rand((R + rps - 1) // rps, Ci, device=DEV, generator=g)
        kw = dict(x_mode=ops.BNACT, x_scale=sc, x_shift=sh, x_gate=gate, rows_per_sample=rps)
        z = Xr * sc.double() + sh.double()
        Xr = z * torch.sigmoid(z) * gate.double().repeat_interleave(rps, dim=0)[:R]
    ref = G.double().t() @ Xr
    base = torch.randn(Co, Ci, device=DEV, generator=g)           # dW += : a running sum must be kept
    outs = []
    for _ in range(2):
        dW = base.clone()
        ops.pw_wgrad(G, X, dW, **kw)
        outs.append(dW)
    assert torch.equal(outs[0], outs[1])                          # fixed summation order
    err = (outs[0].double() - base.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-5 * scale, (err, scale)
    # and the tiled (atomic) kernel agrees
    ops.WGRAD_SMALL = False
    try:
        dW2 = base.clone()
        ops.pw_wgrad(G, X, dW2, **kw)
    finally:
        ops.WGRAD_SMALL = True
    assert (dW2 - outs[0]).abs().max().item() <= 1e-4 * scale
