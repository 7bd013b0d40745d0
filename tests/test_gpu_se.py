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
