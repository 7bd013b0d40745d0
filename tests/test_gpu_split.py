"""The opt-in "split" arithmetic of the forward / data-gradient GEMMs (include/muscle_hip.h, mx_set_gemm_mode): fp32
operands split exactly into three bf16 terms, six products on the bf16 matrix pipe, fp32 accumulation.  Claim under
test: the results are fp32 results - the error against fp64 is no larger than the exact-fp32 MFMA kernel's (up to a
small factor for the different summation order), on every operand prologue / epilogue option and on ragged shapes, and
the model-level parity tests hold in this mode with the SAME tolerances."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def split_everywhere():
    import muscle_amd
    muscle_amd.set_gemm_mode(2)
    yield
    muscle_amd.set_gemm_mode(0)


def _ref(A, W, scale, shift, gate, rps, bias, res, relu, mode):
    A64 = A.double()
    if mode != 0:                       # ops.BNACT = 1: affine + SiLU + per-sample gate; ops.AFFINE = 2: affine only
        A64 = A64 * scale.double() + shift.double()
        if mode == 1:
            A64 = A64 * torch.sigmoid(A64)
            A64 = A64 * gate.double().repeat_interleave(rps, dim=0)[: A.shape[0]]
    C = A64 @ W.double().t()
    if bias is not None:
        C = C + bias.double()
    if res is not None:
        C = C + res.double()
    if relu:
        C = C.clamp_min(0)
    return C


@pytest.mark.parametrize("M,K,N,mode,opts", [
    (25088, 384, 2304, 0, "stats"), (6272, 2304, 384, 1, "stats"), (777, 640, 21, 0, "bias+relu"), (3001, 36, 48, 2, "res"),
    (12544, 224, 1344, 1, "stats"), (4096, 1344, 224, 0, "res+stats"), (130, 132, 200, 0, ""), (128, 16, 128, 0, "stats")])
def test_split_gemm_matches_fp64_as_well_as_fp32_mfma(M, K, N, mode, opts):
    import muscle_amd
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * (K ** -0.5)
    rps = 49 if M % 49 == 0 else 1
    scale = torch.rand(K, device=DEV, generator=g) + 0.5
    shift = torch.randn(K, device=DEV, generator=g) * 0.3
    gate = torch.rand((M + rps - 1) // rps, K, device=DEV, generator=g)
    bias = torch.randn(N, device=DEV, generator=g) if "bias" in opts else None
    res = torch.randn(M, N, device=DEV, generator=g) if "res" in opts else None
    kw = dict(a_mode=mode, a_scale=scale if mode else None, a_shift=shift if mode else None, a_gate=gate if mode == 1 else None,
              rows_per_sample=rps, bias=bias, residual=res, relu="relu" in opts, want_stats="stats" in opts)
    want = _ref(A, W, scale, shift, gate, rps, bias, res, "relu" in opts, mode)
    errs, stats = [], []
    for gm in (0, 2):
        muscle_amd.set_gemm_mode(gm)
        try:
            out = ops.pw_fwd(A, W, N, **kw)
        finally:
            muscle_amd.set_gemm_mode(0)
        if "stats" in opts:
            out, st = out
            s = st.double().sum(0)                     # [2, N]
            stats.append((float((s[0] - want.sum(0)).abs().max() / want.sum(0).abs().max()),
                          float((s[1] - (want * want).sum(0)).abs().max() / (want * want).sum(0).abs().max())))
        errs.append(float((out.double() - want).abs().max()))
    scale_ = float(want.abs().max())
    assert errs[0] <= 2e-6 * scale_ + 1e-6, errs                 # the exact-fp32 kernel itself
    assert errs[1] <= 1.5 * errs[0] + 2e-7 * scale_, errs        # the split kernel is as close to fp64
    for a, b in stats:
        assert a <= 1e-5 and b <= 1e-5


@pytest.mark.parametrize("M,K,N,opts", [
    (25088, 384, 2304, "stats"), (6272, 2304, 384, "stats+res"), (777, 640, 21, "bias+relu"), (3001, 64, 48, "res"),
    (12544, 224, 1344, "stats"), (4096, 1344, 224, "res+stats"), (130, 160, 200, ""), (128, 32, 128, "stats"), (1, 96, 130, "bias"),
    (50176, 48, 288, "stats"), (25088 + 5, 80, 480, "stats+res"), (1000, 48, 96, "bias"), (401408, 48, 288, "")])     # K % 32 == 16: half a K step of zeros
def test_planes_gemm_matches_fp64_as_well_as_fp32_mfma(M, K, N, opts):
    """Second-generation split kernel (mx_pw_fwd_planes: weight planes split once, LDS-DMA, activations straight to registers,
    16x16x32 MFMA) against fp64, beside the exact-fp32 kernel and the first-generation split kernel on the same inputs; ragged
    rows / columns, every epilogue option, bit-identical from run to run."""
    import muscle_amd
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * (K ** -0.5)
    bias = torch.randn(N, device=DEV, generator=g) if "bias" in opts else None
    res = torch.randn(M, N, device=DEV, generator=g) if "res" in opts else None
    kw = dict(bias=bias, residual=res, relu="relu" in opts, want_stats="stats" in opts)
    want = _ref(A, W, None, None, None, 1, bias, res, "relu" in opts, 0)
    plan = ops.PlanesPlan([W])
    (image,) = plan.run()
    assert image is not None
    outs = []
    for gm, planes in ((0, None), (2, None), (2, image), (2, image)):
        muscle_amd.set_gemm_mode(gm)
        try:
            assert planes is None or ops._planes_take(M, K, N)
            out = ops.pw_fwd(A, W, N, planes=planes, **kw)
        finally:
            muscle_amd.set_gemm_mode(0)
        outs.append(out)
    if "stats" in opts:
        assert torch.equal(outs[2][0], outs[3][0]) and torch.equal(outs[2][1], outs[3][1])
        for _, st in outs:
            s = st.double().sum(0)
            assert float((s[0] - want.sum(0)).abs().max() / want.sum(0).abs().max()) <= 1e-5
            assert float((s[1] - (want * want).sum(0)).abs().max() / (want * want).sum(0).abs().max()) <= 1e-5
        outs = [o for o, _ in outs]
    else:
        assert torch.equal(outs[2], outs[3])
    errs = [float((o.double() - want).abs().max()) for o in outs[:3]]
    scale_ = float(want.abs().max())
    assert errs[0] <= 2e-6 * scale_ + 1e-6, errs                 # the exact-fp32 kernel itself
    assert errs[2] <= 1.5 * errs[0] + 2e-7 * scale_, errs        # the planes kernel is as close to fp64
    assert errs[2] <= 1.5 * errs[1] + 2e-7 * scale_, errs        # ... and as close as the first-generation split kernel


@pytest.mark.parametrize("M,K,N,rps", [(100352, 288, 48, 3136), (65536 + 77, 480, 80, 784), (131072, 192, 48, 12544), (70000, 288, 96, 1)])
def test_planes_gemm_with_activated_input_matches_fp64(M, K, N, rps):
    """mx_pw_fwd_planes_act: the project convolution's operand prologue (BN1 scale / shift, SiLU, the SE gate of the row's sample,
    model.py:83-86) in the second-generation split kernel's register loads, against float64 and beside the exact-fp32 kernel with the
    same prologue; ragged row counts, the statistics, bit-identical from run to run."""
    import muscle_amd
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * (K ** -0.5)
    scale = torch.rand(K, device=DEV, generator=g) + 0.5
    shift = torch.randn(K, device=DEV, generator=g) * 0.3
    gate = torch.rand((M + rps - 1) // rps, K, device=DEV, generator=g)
    want = _ref(A, W, scale, shift, gate, rps, None, None, False, 1)
    plan = ops.PlanesPlan([W])
    (image,) = plan.run()
    kw = dict(a_mode=ops.BNACT, a_scale=scale, a_shift=shift, a_gate=gate, rows_per_sample=rps, want_stats=True)
    outs = []
    for gm, planes in ((0, None), (1, image), (1, image)):
        muscle_amd.set_gemm_mode(gm)
        try:
            assert planes is None or ops._planes_act_take(M, K, N)
            outs.append(ops.pw_fwd(A, W, N, planes=planes, **kw))
        finally:
            muscle_amd.set_gemm_mode(0)
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])
    for _, st in outs:
        ssum = st.double().sum(0)
        assert float((ssum[0] - want.sum(0)).abs().max() / want.sum(0).abs().max()) <= 1e-5
        assert float((ssum[1] - (want * want).sum(0)).abs().max() / (want * want).sum(0).abs().max()) <= 1e-5
    errs = [float((o.double() - want).abs().max()) for o, _ in outs[:2]]
    scale_ = float(want.abs().max())
    assert errs[0] <= 2e-6 * scale_ + 1e-6, errs                 # the exact-fp32 kernel itself
    assert errs[1] <= 1.5 * errs[0] + 2e-7 * scale_, errs        # the planes kernel is as close to fp64


@pytest.mark.parametrize("M,K,N,res", [(25088, 960, 160, True), (6272, 2304, 384, False), (3001, 384, 130, True), (700, 1344, 224, False),
                                       (401408, 288, 48, True), (200704, 192, 32, False), (100352 + 37, 288, 48, False)])   # stages 1-2: small-output kernel
def test_bnbwd_fold_in_both_consumers_matches_fp64(M, K, N, res):
    """The BatchNorm backward apply dZ = c1*G + c2*X + c3 folded into BOTH consumers of dZ (mx_pw_dgrad_bnbwd_planes: the data gradient's
    operand load; mx_pw_wgrad_tile_bnbwd: the weight gradient's G operand) against float64, beside the unfused pair (bn_bwd_apply, then
    the same two GEMMs on the materialised dZ); bit-identical from run to run."""
    import muscle_amd
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + K)
    G = torch.randn(M, K, device=DEV, generator=g)
    X = torch.randn(M, K, device=DEV, generator=g)
    coef = torch.stack([torch.rand(K, device=DEV, generator=g) + 0.5, torch.randn(K, device=DEV, generator=g) * 0.3,
                        torch.randn(K, device=DEV, generator=g) * 0.1]).contiguous()
    Wt = torch.randn(N, K, device=DEV, generator=g) * (K ** -0.5)           # W^T of the expand conv [Cin, Cexp]
    Xin = torch.randn(M, N, device=DEV, generator=g)                         # the block input x
    R = torch.randn(M, N, device=DEV, generator=g) if res else None
    dz64 = coef[0].double() * G.double() + coef[1].double() * X.double() + coef[2].double()
    want_dx = dz64 @ Wt.double().t() + (R.double() if res else 0)
    want_dw = dz64.t() @ Xin.double()
    plan = ops.PlanesPlan([Wt])               # (kept alive: the image lives in the plan's buffer)
    (image,) = plan.run()
    muscle_amd.set_gemm_mode(1)
    try:
        if ops.bnbwd_fold_takes(M, K, N) is None:
            pytest.skip("shape not folded in mode 1")
        dz = ops.bn_bwd_apply_plain(G, X, coef, torch.empty_like(G))
        dx_ref = ops.pw_dgrad(dz, None, N, wt=Wt, planes=image, residual=R)
        dw_ref = torch.zeros(K, N, device=DEV)
        ops.pw_wgrad(dz, Xin, dw_ref)
        outs = []
        for _ in range(2):
            dx = ops.pw_dgrad_bnbwd_planes(G, X, coef, image, N, residual=R)
            dw = torch.zeros(K, N, device=DEV)
            ops.pw_wgrad_bnbwd(G, X, coef, Xin, dw)
            outs.append((dx, dw))
    finally:
        muscle_amd.set_gemm_mode(0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for got, ref, want in ((outs[0][0], dx_ref, want_dx), (outs[0][1], dw_ref, want_dw)):
        scale = float(want.abs().max())
        e_got, e_ref = float((got.double() - want).abs().max()), float((ref.double() - want).abs().max())
        assert e_got <= 1.5 * e_ref + 3e-7 * scale, (e_got, e_ref, scale)


def test_planes_image_is_rebuilt_from_the_current_weight():
    """The image is a function of the weight at the time of PlanesPlan.run(): after an in-place update, run() again gives the new
    product; shapes with K % 32 != 0 have no image and stay on the first-generation kernels."""
    import muscle_amd
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    A = torch.randn(1024, 256, device=DEV, generator=g)
    W = torch.randn(192, 256, device=DEV, generator=g)
    W2 = torch.randn(192, 40, device=DEV, generator=g)
    plan = ops.PlanesPlan([W, W2])
    im, none = plan.run()
    assert none is None and im is not None
    muscle_amd.set_gemm_mode(1)
    try:
        a = ops.pw_fwd(A, W, 192, planes=im)
        W.mul_(2.0)
        plan.run()
        b = ops.pw_fwd(A, W, 192, planes=im)
    finally:
        muscle_amd.set_gemm_mode(0)
    assert torch.equal(b, 2.0 * a)                                # scaling by two is exact in every term


def test_split_mode_leaves_small_shapes_on_fp32_mfma_in_mode_1():
    """mode 1 only switches the MFMA-bound shapes: a K = 48 GEMM must give bit-identical results in modes 0 and 1."""
    import muscle_amd
    from muscle_amd import ops
    A = torch.randn(4096, 48, device=DEV)
    W = torch.randn(288, 48, device=DEV)
    a = ops.pw_fwd(A, W, 288)
    muscle_amd.set_gemm_mode(1)
    try:
        b = ops.pw_fwd(A, W, 288)
        assert muscle_amd.get_gemm_mode() == 1
    finally:
        muscle_amd.set_gemm_mode(0)
    assert torch.equal(a, b)


@pytest.mark.parametrize("name,n,size,mode,training", [("efficientnet-b0", 3, 64, "cam", True), ("efficientnet-b7", 2, 64, "cam", True)])
def test_model_parity_holds_in_split_mode(split_everywhere, name, n, size, mode, training):
    import test_gpu_model as tm
    tm.test_model_forward_backward(name, n, size, mode, training)


@pytest.mark.parametrize("fname", ["step_b0_ep4.npz", "step_b7_448_ep4.npz"])
def test_mcl_step_golden_in_split_mode(split_everywhere, fname):
    """The reference's own step fixtures (B0, and B7 at 448x448 where the big layers take the split kernel)."""
    import test_gpu_model as tm
    tm.test_mcl_step_phase1_golden(fname, False)


@pytest.mark.parametrize("R,Co,Ci,mode", [(25088, 384, 2304, "plain"), (12544, 2304, 384, "bnact"), (1111, 200, 136, "bnact"),
                                          (50176, 384, 640, "plain"), (62720 + 5, 256, 384, "bnact"),      # longer than the bench shape: the chain bound, not the group count
                                          (4096, 1344, 224, "affine"), (2048, 128, 128, "plain"), (5000 + 3, 640, 384, "bnact")])
def test_split_wgrad_matches_fp64_as_well_as_fp32_mfma(R, Co, Ci, mode):
    """The tiled weight gradient in split arithmetic (wgrad_split_kernel: fragments along the pixel rows, transposed in
    registers) against fp64, beside the fp32-MFMA kernel on the same inputs; deterministic; keeps a running sum."""
    import muscle_amd
    from muscle_amd import ops
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
    errs = []
    for gm in (0, 2):
        muscle_amd.set_gemm_mode(gm)
        try:
            outs = []
            for _ in range(2):
                dW = base.clone()
                ops.pw_wgrad(G, X, dW, **kw)
                outs.append(dW)
        finally:
            muscle_amd.set_gemm_mode(0)
        assert torch.equal(outs[0], outs[1])
        errs.append((outs[0].double() - base.double() - ref).abs().max().item())
    scale = ref.abs().max().item()
    assert errs[0] <= 2e-5 * scale, (errs, scale)
    assert errs[1] <= 1.5 * errs[0] + 2e-7 * scale, (errs, scale)


@pytest.mark.parametrize("R,Co,Ci,groups", [(25088, 384, 2304, 16), (12544 + 7, 1344, 224, 9), (5003, 640, 384, 4), (3136, 256, 132, 2), (1536, 256, 256, 1),
                                              (6272, 2304, 384, 5)])
def test_wgrad_kernels_of_round_5_equal_the_first_split_kernel_bit_for_bit(R, Co, Ci, groups):
    """wgrad_split_pipe_kernel (one software-pipelined stream) and wgrad_split_ws_kernel (4 MFMA waves + 4 loader waves, persistent)
    keep the first split kernel's tiles, MFMA order and fixed-order sum of the row groups: with the group count held equal the three
    give the same bits - on ragged row counts (the last slab and the last group are short) and on outputs that pad their tiles.  The
    fp64 bound of the planner's own group count is test_split_wgrad_matches_fp64_as_well_as_fp32_mfma's."""
    import muscle_amd
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(R + Ci)
    G = torch.randn(R, Co, device=DEV, generator=g)
    X = torch.randn(R, Ci, device=DEV, generator=g)
    base = torch.randn(Co, Ci, device=DEV, generator=g)
    muscle_amd.set_gemm_mode(2)
    was = ops.get_wgrad_kernel()
    outs = []
    try:
        for kern in (0, 1, 2):
            ops.set_wgrad_kernel(kern, groups)
            dW = base.clone()
            ops.pw_wgrad(G, X, dW)
            outs.append(dW)
    finally:
        ops.set_wgrad_kernel(was, 0)
        muscle_amd.set_gemm_mode(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = base.double() + G.double().t() @ X.double()
    assert (outs[2].double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()


@pytest.mark.parametrize("M,K,N", [(25088, 960, 160), (12544 + 5, 2304, 384), (6272, 1344, 224), (3136 * 3, 3840, 640)])
def test_wgrad_side_fold_leaves_dz_for_the_data_gradient(M, K, N):
    """Round 5's form of the BatchNorm-backward fold: mx_pw_wgrad_tile_bnbwd_dz forms dZ = c1*G + c2*X + c3 in the loader waves of the
    weight-gradient kernel, uses it and STORES it; the data gradient then reads the stored dZ.  Held to the unfused pair (bn_bwd_apply,
    then the plain weight gradient) and to float64: dZ itself to one rounding of its three terms, dW like the unfused kernel; the rows
    past a ragged end and the padded tile columns leave nothing behind; bit-identical from run to run."""
    import muscle_amd
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + N)
    G = torch.randn(M, K, device=DEV, generator=g)
    X = torch.randn(M, K, device=DEV, generator=g)
    coef = torch.stack([torch.rand(K, device=DEV, generator=g) + 0.5, torch.randn(K, device=DEV, generator=g) * 0.3,
                        torch.randn(K, device=DEV, generator=g) * 0.1]).contiguous()
    Xin = torch.randn(M, N, device=DEV, generator=g)
    dz64 = coef[0].double() * G.double() + coef[1].double() * X.double() + coef[2].double()
    want_dw = dz64.t() @ Xin.double()
    muscle_amd.set_gemm_mode(1)
    try:
        if not ops.wgrad_bnbwd_dz_takes(M, K, N):
            pytest.skip("shape not taken by the wave-specialised kernel in mode 1")
        dz_ref = ops.bn_bwd_apply_plain(G, X, coef, torch.empty_like(G))
        dw_ref = torch.zeros(K, N, device=DEV)
        ops.pw_wgrad(dz_ref, Xin, dw_ref)
        outs = []
        for _ in range(2):
            dw = torch.zeros(K, N, device=DEV)
            dz = torch.full_like(G, float("nan"))                      # every element must be written
            ops.pw_wgrad_bnbwd_dz(G, X, coef, Xin, dw, dz)
            outs.append((dz, dw))
    finally:
        muscle_amd.set_gemm_mode(0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dz, dw = outs[0]
    assert torch.isfinite(dz).all()
    mag = (coef[0].abs() * G.abs() + coef[1].abs() * X.abs() + coef[2].abs()).double()
    assert float(((dz.double() - dz64).abs() / mag).max()) <= 2.0 * 2 ** -24            # two fused roundings at most
    assert float((dz - dz_ref).abs().max()) <= 4e-7 * float(dz64.abs().max())
    scale = float(want_dw.abs().max())
    e_got, e_ref = float((dw.double() - want_dw).abs().max()), float((dw_ref.double() - want_dw).abs().max())
    assert e_got <= 1.5 * e_ref + 3e-7 * scale, (e_got, e_ref, scale)
