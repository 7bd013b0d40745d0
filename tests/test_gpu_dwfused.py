"""The fused stride-1 depthwise backward (mx_dwconv_bwd_fused: BN1 data gradient formed while the tile is staged, weight + data
gradients from the one staged tile, swish'(bn0) and the BN0 backward sums in the epilogue; reference model.py:76-90 backward)
against float64 autograd of the same chain, on every tile shape the kernel has (14 x 28 and 8 x 16, 3x3 and 5x5), ragged images,
channel counts that are not a multiple of the 32-channel chunk, with and without BN0 / residual."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _swish_grad(z):
    s = torch.sigmoid(z)
    return s * (1 + z * (1 - s))


def _reference(dA, D, gate, add, st1, c1, X, st0, W, K, residual):
    d = lambda t: t.double()
    dA, D, gate, add, X, W, c1 = map(d, (dA, D, gate, add, X, W, c1))
    a1, b1 = d(st1.scale), d(st1.shift)
    dd = c1[0] * ((dA * gate[:, None, None, :] + add[:, None, None, :]) * _swish_grad(a1 * D + b1)) + c1[1] * D + c1[2]
    x = X.clone().requires_grad_()
    w = W.clone().requires_grad_()
    act = F.silu(d(st0.scale) * x + d(st0.shift)) if st0 is not None else x
    y = F.conv2d(act.permute(0, 3, 1, 2), w, padding=(K - 1) // 2, groups=X.shape[3])
    (y * dd.permute(0, 3, 1, 2)).sum().backward()
    gX = x.grad
    part = None
    if st0 is not None:
        gX = gX / d(st0.scale)       # the kernel returns dL/d(a0*x + b0): the BatchNorm-0 backward that follows carries the scale
        part = torch.stack([gX.sum((0, 1, 2)), (gX * X).sum((0, 1, 2))])
    elif residual is not None:
        gX = gX + residual.double()
    return gX, w.grad, part


@pytest.mark.parametrize("N,H,W,C,K,bn0,res", [
    (2, 28, 28, 64, 5, True, False), (2, 28, 28, 40, 3, True, False),       # 14 x 28 tiles (B7's 28 x 28 stages), ragged channels
    (3, 56, 56, 36, 5, True, False), (2, 56, 56, 32, 3, False, True),       # several tiles per plane; the plain-input (stage 1) form
    (2, 14, 28, 32, 5, True, False), (1, 42, 84, 16, 3, True, False),
    (2, 20, 37, 48, 5, True, False), (2, 20, 37, 20, 3, True, False),       # 8 x 16 tiles, image not a multiple of the tile
    (3, 9, 9, 32, 5, False, True), (2, 16, 16, 64, 3, False, False),
    (4, 112, 112, 32, 3, True, False), (2, 112, 112, 24, 5, True, False)])
def test_fused_depthwise_backward_matches_float64(N, H, W, C, K, bn0, res):
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(N * 1000 + H * 10 + C + K)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    dA, D, X = rn(N, H, W, C), rn(N, H, W, C), rn(N, H, W, C)
    gate, add = torch.sigmoid(rn(N, C)), rn(N, C) * 0.1
    mk = lambda: ops.BNState(torch.rand(C, device=DEV, generator=g) + 0.5, rn(C) * 0.1, rn(C) * 0.1, torch.rand(C, device=DEV, generator=g) + 0.5)
    st1, st0 = mk(), (mk() if bn0 else None)
    c1 = rn(3, C) * 0.3
    Wt = rn(C, 1, K, K) * 0.3
    residual = rn(N, H, W, C) if res else None
    want_gx, want_dw, want_part = _reference(dA, D, gate, add, st1, c1, X, st0, Wt, K, residual)
    outs = []
    for _ in range(2):
        dW = torch.zeros_like(Wt)
        gX, part = ops.dwconv_bwd_fused(dA, D, gate, add, st1, c1, X, st0, Wt, dW, K, (K - 1) // 2, residual=residual)
        outs.append((gX, dW, part))
    (gX, dW, part), (gX2, dW2, part2) = outs
    assert torch.equal(gX, gX2) and torch.equal(dW, dW2)                    # same bits every run (ordered partial rows, no atomics)
    assert float((gX.double() - want_gx).abs().max()) <= 2e-5 * float(want_gx.abs().max()) + 1e-6
    assert float((dW.double() - want_dw).abs().max()) <= 5e-5 * float(want_dw.abs().max()) + 1e-5
    if bn0:
        assert torch.equal(part, part2)
        got = part.double().sum(0)
        assert float((got - want_part).abs().max()) <= 5e-5 * float(want_part.abs().max()) + 1e-5
        # the same kernel finishing the BatchNorm-0 backward statistics itself (mx_dwconv_bwd_fused_bn0): what the separate
        # mx_bn_bwd_finalize launch leaves - bit for bit while both add the partial rows in one pass (<= 1024 rows), run twice
        bn = torch.nn.BatchNorm2d(C).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, device=DEV, generator=g) + 0.5)
        rows = N * H * W
        dg_a, db_a = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        c_sep = ops.bn_bwd_coeffs(part, rows, bn, st0, dg_a, db_a, True)
        for _ in range(2):
            dg_b, db_b = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            dW3 = torch.zeros_like(Wt)
            gX3, part3, c_fin = ops.dwconv_bwd_fused(dA, D, gate, add, st1, c1, X, st0, Wt, dW3, K, (K - 1) // 2, bn0=(bn, dg_b, db_b, True))
            assert torch.equal(gX3, gX) and torch.equal(part3, part) and torch.equal(dW3, dW)
            if part.shape[0] <= 1024:
                assert torch.equal(c_fin, c_sep) and torch.equal(dg_b, dg_a) and torch.equal(db_b, db_a)
            else:
                assert torch.allclose(c_fin, c_sep, rtol=1e-5, atol=1e-6) and torch.allclose(dg_b, dg_a, rtol=1e-5, atol=1e-5)
    else:
        assert part is None


@pytest.mark.parametrize("N,H,W,C,K,S,act", [
    (2, 28, 28, 64, 5, 1, True), (2, 28, 28, 40, 3, 1, True), (3, 56, 56, 36, 5, 1, True), (2, 112, 112, 32, 3, 1, False),     # 8 x 28 row-staged tiles
    (1, 30, 84, 16, 5, 1, True), (2, 12, 28, 32, 3, 1, True),                                                                  # ... ragged rows
    (2, 20, 37, 48, 5, 1, True), (2, 16, 16, 20, 3, 1, False),                                                                 # 8 x 16 tiles
    (2, 56, 56, 32, 3, 2, True), (2, 28, 28, 24, 5, 2, True)])                                                                 # stride 2 (TF 'same' padding)
def test_depthwise_forward_matches_float64(N, H, W, C, K, S, act):
    """mx_dwconv_fwd (model.py:76-79: BN0 + SiLU of the expand output applied while the tile is staged, depthwise convolution with
    TensorFlow 'same' padding, the BatchNorm-1 statistics as partial rows; eval mode: the SE squeeze sums instead) against float64, on
    both stride-1 tile shapes, stride 2, ragged images and channel counts."""
    from muscle_amd import ops
    g = torch.Generator(device=DEV).manual_seed(N * 1000 + H * 10 + C + K + S)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    X = rn(N, H, W, C)
    Wt = rn(C, 1, K, K) * 0.3
    st = ops.BNState(torch.rand(C, device=DEV, generator=g) + 0.5, rn(C) * 0.1, rn(C) * 0.1, torch.rand(C, device=DEV, generator=g) + 0.5) if act else None
    Ho, Wo = (H + S - 1) // S, (W + S - 1) // S
    pt_h, pt_w = max((Ho - 1) * S + K - H, 0), max((Wo - 1) * S + K - W, 0)
    x64 = X.double()
    if act:
        x64 = F.silu(st.scale.double() * x64 + st.shift.double())
    xp = F.pad(x64.permute(0, 3, 1, 2), (pt_w // 2, pt_w - pt_w // 2, pt_h // 2, pt_h - pt_h // 2))
    want = F.conv2d(xp, Wt.double(), stride=S, groups=C).permute(0, 2, 3, 1)
    assert pt_h // 2 == pt_w // 2
    outs = [ops.dwconv_fwd(X, Wt, K, S, pt_h // 2, Ho, Wo, st=st, want_stats=True) for _ in range(2)]
    (Y, stats), (Y2, stats2) = outs
    assert torch.equal(Y, Y2) and torch.equal(stats, stats2)
    assert float((Y.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-6
    ssum = stats.double().sum(0)
    assert float((ssum[0] - want.sum((0, 1, 2))).abs().max()) <= 2e-5 * float(want.abs().sum((0, 1, 2)).max()) + 1e-5
    assert float((ssum[1] - (want * want).sum((0, 1, 2))).abs().max()) <= 2e-5 * float((want * want).sum((0, 1, 2)).max()) + 1e-5
    # inference form: no statistics, the squeeze sums of swish(ps * y + pb) per sample instead
    ps, pb = torch.rand(C, device=DEV, generator=g) + 0.5, rn(C) * 0.1
    Yp, pooled = ops.dwconv_fwd(X, Wt, K, S, pt_h // 2, Ho, Wo, st=st, pool=(ps, pb))
    assert float((Yp.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-6
    want_pool = F.silu(ps.double() * want + pb.double()).sum((1, 2))
    assert float((pooled.double() - want_pool).abs().max()) <= 2e-5 * float(want_pool.abs().max()) + 1e-5
