"""The BatchNorm finalisation entry points (mx_bn_finalize, mx_bn_bwd_finalize) against numpy fp64 over every partial-row count the
producers can hand them: fewer rows than row lanes, every residue of the 4-row batches, the one-launch form's limit (1024 rows) and the
two-level form above it.  Round 5 changed how the one-launch form REQUESTS its rows (all of a lane's rows before the first addition);
the sums and their order are the same, and these cases pin the walk over the rows.

Reference semantics: torch.nn.functional.batch_norm in training mode as the reference's MBConv blocks call it
(src/efficientnet_pytorch/model.py:73,79,88) - biased variance for the normalisation, unbiased for running_var - and its backward."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROWS = [1, 2, 3, 5, 7, 8, 9, 15, 31, 32, 33, 40, 63, 64, 98, 191, 192, 196, 200, 223, 392, 784, 1023, 1024, 1025, 3136]


def _bn(C, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    bn = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    return bn


@pytest.mark.parametrize("C", [32, 48, 2304])
@pytest.mark.parametrize("P", ROWS)
def test_bn_finalize_over_partial_row_counts(P, C):
    from muscle_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1000 * P + C)
    rows_per_part = 128
    count = float(P * rows_per_part)
    # partial rows as a producer leaves them: per workgroup the sums of x and x^2 over its 128 rows
    x = rng.normal(0.3, 1.2, size=(P, rows_per_part, C)).astype(np.float32) if P * C <= 40000 else None
    if x is not None:
        part = np.stack([x.sum(1, dtype=np.float64), (x.astype(np.float64) ** 2).sum(1)], 1).astype(np.float32)
    else:
        s0 = rng.normal(0.3 * rows_per_part, 3.0, size=(P, C))
        part = np.stack([s0, s0 * s0 / rows_per_part + rows_per_part * rng.uniform(0.8, 1.6, size=(P, C))], 1).astype(np.float32)
    bn = _bn(C, dev, 7)
    rm0, rv0 = bn.running_mean.cpu().numpy().astype(np.float64), bn.running_var.cpu().numpy().astype(np.float64)
    st = ops.bn_finalize(torch.from_numpy(part).to(dev), count, bn, True)
    torch.cuda.synchronize()
    s = part.astype(np.float64).sum(0)
    mean = s[0] / count
    var = np.maximum(s[1] / count - mean * mean, 0.0)
    rstd = 1.0 / np.sqrt(var + bn.eps)
    gamma, beta = bn.weight.detach().cpu().numpy().astype(np.float64), bn.bias.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(st.mean.cpu().numpy(), mean, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(st.rstd.cpu().numpy(), rstd, rtol=3e-6)
    np.testing.assert_allclose(st.scale.cpu().numpy(), gamma * rstd, rtol=4e-6)
    np.testing.assert_allclose(st.shift.cpu().numpy(), beta - mean * gamma * rstd, rtol=1e-5, atol=2e-6)
    mom = bn.momentum
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), (1 - mom) * rm0 + mom * mean, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), (1 - mom) * rv0 + mom * var * count / (count - 1), rtol=4e-6)

    # backward: partial rows of (sum g, sum g*x); dgamma / dbeta accumulate, coefficients of dX = c1*g + c2*x + c3
    gp = rng.normal(0.0, 1.0, size=(P, 2, C)).astype(np.float32)
    dg0, db0 = rng.normal(size=C).astype(np.float32), rng.normal(size=C).astype(np.float32)
    dgamma, dbeta = torch.from_numpy(dg0.copy()).to(dev), torch.from_numpy(db0.copy()).to(dev)
    c = ops.bn_bwd_coeffs(torch.from_numpy(gp).to(dev), count, bn, st, dgamma, dbeta, True)
    torch.cuda.synchronize()
    sg = gp.astype(np.float64).sum(0)
    m32, r32 = st.mean.cpu().numpy().astype(np.float64), st.rstd.cpu().numpy().astype(np.float64)
    dgam = r32 * (sg[1] - m32 * sg[0])
    k = gamma * r32 * r32 * (dgam / count)
    scale = np.abs(sg).max() + 1.0
    np.testing.assert_allclose(dgamma.cpu().numpy(), dg0 + dgam, rtol=1e-5, atol=2e-6 * scale)
    np.testing.assert_allclose(dbeta.cpu().numpy(), db0 + sg[0], rtol=1e-5, atol=2e-6 * scale)
    got = c.cpu().numpy()
    np.testing.assert_allclose(got[0], gamma * r32, rtol=3e-6)
    np.testing.assert_allclose(got[1], -k, rtol=1e-5, atol=2e-6 * scale / count)
    np.testing.assert_allclose(got[2], -gamma * r32 * (sg[0] / count) + k * m32, rtol=1e-5, atol=4e-6 * scale / count)


def test_bn_finalize_is_the_same_bits_every_run():
    from muscle_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    part = (torch.randn(196, 2, 2304, generator=g) * 50).to(dev)
    outs = []
    for _ in range(3):
        bn = _bn(2304, dev, 9)
        st = ops.bn_finalize(part, 25088.0, bn, True)
        outs.append(torch.cat([st.scale, st.shift, st.mean, st.rstd, bn.running_mean, bn.running_var]).cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
