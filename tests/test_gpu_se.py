"""The SE excitation path (mx_se_fwd, mx_se_bwd, mx_se_bwd_gh + mx_se_bwd_params) against an fp64 autograd restatement of
src/efficientnet_pytorch/model.py:81-84 (squeeze -> se_reduce -> swish -> se_expand -> sigmoid gate) over widths whose squeeze
size is / is not a multiple of 4 or 16 (the round-5 backward kernel covers 16 squeeze units per workgroup with 1024 threads; other
EfficientNet widths take the one-unit-per-wave form) and batch sizes beside 32.  The whole-network parity tests cover the same
kernels through the blocks; this one pins them alone."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [(32, 2304, 96), (32, 3840, 160), (3, 48, 12), (5, 32, 8), (2, 240, 10), (16, 672, 28), (64, 960, 40), (1, 1344, 56),
         (7, 144, 6), (4, 192, 48), (33, 480, 20)]


def _ref(pooled, inv_hw, W1, b1, W2, b2, ggate):
    p = [t.double().detach().requires_grad_(t is not pooled) for t in (pooled, W1, b1, W2, b2)]
    pooled64, W1_, b1_, W2_, b2_ = p
    s = (pooled64 * inv_hw).requires_grad_(True)
    h = s @ W1_.t() + b1_
    r = h * torch.sigmoid(h)
    gate = torch.sigmoid(r @ W2_.t() + b2_)
    h.retain_grad()
    (gate * ggate.double()).sum().backward()
    return s.detach(), h.detach(), gate.detach(), h.grad, W1_.grad, b1_.grad, W2_.grad, b2_.grad, s.grad


@pytest.mark.parametrize("N,C,SQ", CASES)
def test_se_excitation_forward_backward_vs_fp64(N, C, SQ):
    from muscle_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(N * 100003 + C * 17 + SQ)
    hw = 784
    pooled = (torch.randn(N, C, generator=g) * 0.7 * hw).to(dev)
    W1 = (torch.randn(SQ, C, generator=g) / C ** 0.5).to(dev)
    b1 = (torch.randn(SQ, generator=g) * 0.1).to(dev)
    W2 = (torch.randn(C, SQ, generator=g) / SQ ** 0.5).to(dev)
    b2 = (torch.randn(C, generator=g) * 0.1).to(dev)
    ggate = torch.randn(N, C, generator=g).to(dev)
    s64, h64, gate64, gh64, dW1_64, db1_64, dW2_64, db2_64, gs64 = _ref(pooled, 1.0 / hw, W1, b1, W2, b2, ggate)

    s, h, gate = ops.se_fwd(pooled, 1.0 / hw, W1, b1, W2, b2)
    tol = dict(rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(s.double(), s64, **tol)
    torch.testing.assert_close(h.double(), h64, rtol=2e-5, atol=1e-5)
    torch.testing.assert_close(gate.double(), gate64, **tol)

    def grads():
        return [torch.randn(SQ, C, generator=g).to(dev), torch.randn(SQ, generator=g).to(dev),
                torch.randn(C, SQ, generator=g).to(dev), torch.randn(C, generator=g).to(dev)]
    # the one-call form and the two halves the engine uses (gh on the chain, parameter gradients aside): "+=" into what is handed in
    for split in (False, True):
        base = grads()
        acc = [b.clone() for b in base]
        if split:
            gh = ops.se_bwd_gh(ggate, gate, h, W2)
            ops.se_bwd_params(ggate, gate, s, h, gh, *acc)
        else:
            gh = ops.se_bwd(ggate, gate, s, h, W2, *acc)
        scale = float(gh64.abs().max()) + 1e-6
        torch.testing.assert_close(gh.double(), gh64, rtol=3e-5, atol=3e-6 * scale)
        for got, b, ref in zip(acc, base, (dW1_64, db1_64, dW2_64, db2_64)):
            sc = float(ref.abs().max()) + 1e-6
            torch.testing.assert_close(got.double() - b.double(), ref, rtol=5e-5, atol=2e-5 * sc)


def test_se_backward_is_the_same_bits_every_run():
    from muscle_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    N, C, SQ = 32, 2304, 96
    gate = torch.rand(N, C, generator=g).to(dev)
    ggate = torch.randn(N, C, generator=g).to(dev)
    h = torch.randn(N, SQ, generator=g).to(dev)
    W2 = torch.randn(C, SQ, generator=g).to(dev)
    a = ops.se_bwd_gh(ggate, gate, h, W2).cpu()
    for _ in range(3):
        assert torch.equal(a, ops.se_bwd_gh(ggate, gate, h, W2).cpu())


@pytest.mark.parametrize("N,C,SQ", [(32, 2304, 96), (32, 960, 40), (3, 48, 12), (16, 672, 28), (2, 240, 10), (33, 1344, 56), (64, 3840, 160), (5, 32, 8)])
def test_bn1_sums_finalize_vs_fp64(N, C, SQ):
    """mx_bn1_sums_finalize: the pooled-path gradient add[n,c] = inv_hw * sum_j gh[n,j] W1[j,c] (model.py:82-83 backward) and, from the five
    per-sample sums of mx_se_bn1_pool, the BatchNorm-1 backward sums sum_n gate*S1 + add*S2 and sum_n gate*S3 + add*S4 with their
    finalisation (dgamma / dbeta +=, coefficients of dX = c1*g + c2*x + c3) - against numpy fp64."""
    import numpy as np
    from muscle_amd import ops
    from muscle_amd.ops import BNState
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(N * 7919 + C + SQ)
    hw = 784
    rows = float(N * hw)
    pooled5 = rng.normal(0, 30.0, size=(5, N, C)).astype(np.float32)
    gate = rng.uniform(0.05, 0.95, size=(N, C)).astype(np.float32)
    gh = rng.normal(0, 1.0, size=(N, SQ)).astype(np.float32)
    W1 = (rng.normal(size=(SQ, C)) / np.sqrt(C)).astype(np.float32)
    bn = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, size=C).astype(np.float32)))
    mean = rng.normal(0, 0.5, size=C).astype(np.float32)
    rstd = rng.uniform(0.5, 2.0, size=C).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    st = BNState(T(rstd), T(mean), T(mean), T(rstd))
    dg0, db0 = rng.normal(size=C).astype(np.float32), rng.normal(size=C).astype(np.float32)
    dgamma, dbeta = T(dg0.copy()), T(db0.copy())
    c, add = ops.bn1_coeffs(T(pooled5), T(gate), T(gh), T(W1), 1.0 / hw, rows, bn, st, dgamma, dbeta, True)
    torch.cuda.synchronize()
    add64 = (gh.astype(np.float64) @ W1.astype(np.float64)) / hw
    np.testing.assert_allclose(add.cpu().numpy(), add64, rtol=2e-5, atol=2e-6 * np.abs(add64).max())
    a32 = add.cpu().numpy().astype(np.float64)            # the sums are formed from the fp32 `add` the kernel stores
    p = pooled5.astype(np.float64)
    g64 = gate.astype(np.float64)
    s0 = (g64 * p[1] + a32 * p[2]).sum(0)
    s1 = (g64 * p[3] + a32 * p[4]).sum(0)
    gam, m, r = bn.weight.detach().cpu().numpy().astype(np.float64), mean.astype(np.float64), rstd.astype(np.float64)
    dgam = r * (s1 - m * s0)
    k = gam * r * r * (dgam / rows)
    sc = np.abs(s1).max() + np.abs(s0).max() + 1.0
    np.testing.assert_allclose(dgamma.cpu().numpy(), dg0 + dgam, rtol=2e-5, atol=3e-6 * sc)
    np.testing.assert_allclose(dbeta.cpu().numpy(), db0 + s0, rtol=2e-5, atol=3e-6 * sc)
    got = c.cpu().numpy()
    np.testing.assert_allclose(got[0], gam * r, rtol=3e-6)
    np.testing.assert_allclose(got[1], -k, rtol=2e-5, atol=3e-6 * sc / rows)
    np.testing.assert_allclose(got[2], -gam * r * (s0 / rows) + k * m, rtol=2e-5, atol=6e-6 * sc / rows)
