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


@pytest.mark.parametrize("R,Co,Ci,mode", [(25088, 384, 2304, "plain"), (25088, 2304, 384, "bnact"), (12544, 960, 160, "plain"),
                                          (12544, 224, 1344, "bnact"), (25088, 640, 3840, "plain"), (1111, 200, 136, "bnact"),
                                          (2048, 128, 128, "plain"), (5000 + 3, 480, 80, "bnact"), (4096, 1344, 224, "affine")])
def test_wgrad_tile_matches_fp64_and_is_deterministic(R, Co, Ci, mode):
    """Large-output weight gradient (wgrad_tile_kernel): every tile shape (128/64 x 128/64), ragged rows and edges, the
    operand prologues, a running sum in dW, bit-identical repeat runs, agreement with the atomic TN GEMM it replaces."""
    from muscle_amd import ops
    from muscle_amd._lib import lib
    assert lib().mx_pw_wgrad_small_ws(R, Co, Ci, 1 if mode == "bnact" else 0) == 0
    assert lib().mx_pw_wgrad_tile_ws(R, Co, Ci, {'plain': 0, 'bnact': 1, 'affine': 2}[mode]) > 0
    g = torch.Generator(device=DEV).manual_seed(R + Co)
    G = torch.randn(R, Co, device=DEV, generator=g)
    X = torch.randn(R, Ci, device=DEV, generator=g)
    kw = {}
    Xr = X.double()
    if mode != "plain":
        rps = 49
        sc = torch.rand(Ci, device=DEV, generator=g) + 0.5
        sh = torch.randn(Ci, device=DEV, generator=g) * 0.3
        gate = torch.rand((R + rps - 1) // rps, Ci, device=DEV, generator=g)
        Xr = Xr * sc.double() + sh.double()
        if mode == "bnact":
            kw = dict(x_mode=ops.BNACT, x_scale=sc, x_shift=sh, x_gate=gate, rows_per_sample=rps)
            Xr = Xr * torch.sigmoid(Xr) * gate.double().repeat_interleave(rps, dim=0)[:R]
        else:
            kw = dict(x_mode=ops.AFFINE, x_scale=sc, x_shift=sh, rows_per_sample=rps)
    ref = G.double().t() @ Xr
    base = torch.randn(Co, Ci, device=DEV, generator=g)
    outs = []
    for _ in range(2):
        dW = base.clone()
        ops.pw_wgrad(G, X, dW, **kw)
        outs.append(dW)
    assert torch.equal(outs[0], outs[1])
    scale = ref.abs().max().item()
    err = (outs[0].double() - base.double() - ref).abs().max().item()
    assert err <= 2e-5 * scale, (err, scale)
    ops.WGRAD_TILE = False
    try:
        dW2 = base.clone()
        ops.pw_wgrad(G, X, dW2, **kw)
    finally:
        ops.WGRAD_TILE = True
    assert (dW2 - outs[0]).abs().max().item() <= 1e-4 * scale


def test_dgrad_with_folded_bn_backward_apply():
    """mx_pw_dgrad_bnbwd (the BN backward apply inside the data-gradient GEMM's operand load; an option, off by default)
    against the two separate steps it replaces: same dX, same materialised dZ."""
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(9)
    for (M, K, N) in ((6272, 2304, 384), (3000, 288, 48), (777, 160, 960)):
        G = torch.randn(M, K, device=DEV, generator=g)
        X = torch.randn(M, K, device=DEV, generator=g)
        c = torch.randn(3, K, device=DEV, generator=g)
        W = torch.randn(K, N, device=DEV, generator=g) * K ** -0.5          # [Cexp, Cin]
        res = torch.randn(M, N, device=DEV, generator=g)
        dz_ref = ops.bn_bwd_apply_plain(G, X, c, torch.empty_like(G))
        dx_ref = ops.pw_dgrad(dz_ref, W, N, residual=res)
        dx, dz = ops.pw_dgrad_bnbwd(G, X, c, W, N, residual=res)
        assert torch.allclose(dz, dz_ref, rtol=0, atol=2e-6 * float(dz_ref.abs().max()))
        assert (dx - dx_ref).abs().max().item() <= 2e-5 * dx_ref.abs().max().item()


@pytest.mark.parametrize("R,Co,Ci,pad", [(25088, 2304, 384, 0), (25107, 640, 2304, 0), (12551, 1344, 256, 0), (1024, 128, 128, 0), (2048, 256, 384, 64),
                                         (50176, 384, 640, 0), (4700, 512, 128, 32), (3136, 1280, 640, 0)])
def test_exact_fp32_wgrad_on_specialised_waves_vs_fp64(R, Co, Ci, pad):
    """wgrad_f32_ws_kernel (mx_set_gemm_mode(0), plain operands, outputs tiled 128 x 128): rows moved by range-checked LDS-DMA, one float per
    operand and lane into v_mfma_f32_32x32x2_f32.  Ragged row counts, a last tile of 64 columns, operands that are column slices of wider
    tensors (leading dimension > width), one / two / three 1568-row chains per group, a running sum in dW, the same bits on a second run -
    against fp64, held to the error of an fp32 dot product of that length."""
    from muscle_amd import ops
    from muscle_amd._lib import lib
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(0)
    try:
        assert lib().mx_pw_wgrad_small_ws(R, Co, Ci, 0) == 0 and lib().mx_pw_wgrad_tile_ws(R, Co, Ci, 0) > 0
        g = torch.Generator(device=DEV).manual_seed(R * 3 + Co + Ci)
        Gw = torch.randn(R, Co + pad, device=DEV, generator=g)
        Xw = torch.randn(R, Ci + pad, device=DEV, generator=g)
        G, X = Gw[:, :Co], Xw[:, pad:] if pad else Xw          # (the X slice starts `pad` columns in: a base pointer off the row start)
        ref = G.double().t() @ X.double()
        base = torch.randn(Co, Ci, device=DEV, generator=g)
        outs = []
        from muscle_amd._lib import call, ptr, stream
        need = lib().mx_pw_wgrad_tile_ws(R, Co, Ci, 0)
        ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
        for _ in range(2):
            dW = base.clone()
            if pad:      # the C entry point takes the leading dimensions; the Python wrapper only hands over whole tensors
                call("mx_pw_wgrad_tile", G.data_ptr(), X.data_ptr(), 0, None, None, None, 1, ptr(dW), R, Co, Ci, G.stride(0), X.stride(0),
                     ws.data_ptr(), ws.numel(), stream())
            else:
                ops.pw_wgrad(G, X, dW)
            outs.append(dW)
        assert torch.equal(outs[0], outs[1])
        scale = ref.abs().max().item()
        err = (outs[0].double() - base.double() - ref).abs().max().item()
        assert err <= 2e-6 * scale + 4e-7 * float(R) ** 0.5 * 4.0, (err, scale)
    finally:
        ops.set_gemm_mode(prev)
