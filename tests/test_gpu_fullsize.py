"""Parity properties at the headline size (EfficientNet-B7, 448x448, batch 32 = BASELINE.json configs[2]), where the CPU
oracle is too slow to be the comparator.  The property is exact and size independent:

  a convolution followed by a train-mode BatchNorm sees a loss that is invariant to scaling its weight, up to eps:
      <dW[c], W[c]>  =  eps * rstd[c]^2 * gamma[c] * dgamma[c]          for every output channel c

(d/d alpha of gamma*alpha*(z-mu)/sqrt(alpha^2 var + eps) at alpha = 1).  It ties together, per channel and for all
166 conv -> BN pairs of the network, the weight-gradient GEMMs / depthwise weight gradient, the BatchNorm backward, the
statistics and everything upstream of them; a wrong tile, a dropped partial row or a mis-scaled gradient breaks it.
Checked on one full mcl_step (epoch-4 gates) with lr = 0, together with finiteness and run-to-run agreement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pairs(model, cfg, N, size):
    """(conv weight as [Cout, -1], bn module, elements per channel) for every conv -> train-mode BN pair of the backbone."""
    bb = model.backbone
    h = cfg.stem_out_size(size)
    out = [(bb._conv_stem.weight, bb._bn0, N * h * h)]
    for b, m in zip(cfg.blocks, bb._blocks):
        ho = b.out_size(h)
        if b.expand:
            out.append((m._expand_conv.weight, m._bn0, N * h * h))
        out.append((m._depthwise_conv.weight, m._bn1, N * ho * ho))
        out.append((m._project_conv.weight, m._bn2, N * ho * ho))
        h = ho
        if b.index == cfg.taps[6]:
            break                                      # blocks after the last tap get no gradient (dead tail, as the reference)
    return out


def test_b7_448_bs32_scale_invariance_identity():
    import bench
    import muscle_amd
    from muscle_amd import arch
    name, N, size = "efficientnet-b7", 32, 448
    cfg = arch.net_cfg(name, False)
    torch.manual_seed(0)
    model = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False).to(DEV)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0                          # running_var <- this batch's (unbiased) variance
    opt = muscle_amd.FusedAdam(model.parameters(), lr=0.0, weight_decay=0.0)
    batch = bench.make_batch(N, size, size // 2, 1234, torch.device(DEV))
    vc = int(batch["label"].sum().item())
    torch.manual_seed(1)
    out = muscle_amd.mcl_step(model, opt, batch, 4, valid_channel=vc)
    losses = {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in out.items()}
    assert all(np.isfinite(v) for v in losses.values()), losses
    worst = 0.0
    checked = 0
    for w, bn, n in _pairs(model, cfg, N, size):
        assert w.grad is not None and bn.weight.grad is not None
        assert bool(torch.isfinite(w.grad).all()) and bool(torch.isfinite(bn.weight.grad).all())
        W = w.detach().double().flatten(1)
        G = w.grad.double().flatten(1)
        lhs = (W * G).sum(1)
        var_b = bn.running_var.double() * (n - 1) / n
        rstd2 = 1.0 / (var_b + bn.eps)
        rhs = bn.eps * rstd2 * bn.weight.detach().double() * bn.weight.grad.double()
        scale = W.norm(dim=1) * G.norm(dim=1) + 1e-30
        err = ((lhs - rhs).abs() / scale).max().item()
        # measured on MI355X: the two sides are up to 0.44 (median 3e-3) of |W||dW| and agree to 3.3e-6 at worst
        # (median 1.8e-7) over all 166 layers
        assert err <= 5e-5, (tuple(w.shape), err, float((lhs.abs() / scale).max()))
        worst = max(worst, err)
        checked += W.shape[0]
    assert checked > 100000                              # every output channel of 160+ convolutions
    # run to run: same batch, same drop_connect draws -> same losses up to the order of fp32 atomics
    torch.manual_seed(1)
    out2 = muscle_amd.mcl_step(model, opt, batch, 4, valid_channel=vc)
    for k, v in out2.items():
        v = float(v.detach()) if torch.is_tensor(v) else float(v)
        assert abs(v - losses[k]) <= 1e-4 * max(1.0, abs(losses[k])), (k, v, losses[k])
