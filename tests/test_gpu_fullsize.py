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

pytestmark = [pytest.mark.gpu, pytest.mark.both_arith]
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


def scale_invariance_identity(name, N, size, min_channels):
    """One full mcl_step (epoch-4 gates, lr 0) of `name` at batch N, size x size; the identity over every conv -> BN pair."""
    import bench
    import muscle_amd
    from muscle_amd import arch
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
        # measured on MI355X (B7): the two sides are up to 0.44 (median 3e-3) of |W||dW| and agree to 3.3e-6 at worst
        # (median 1.8e-7) over all 166 layers
        assert err <= 5e-5, (tuple(w.shape), err, float((lhs.abs() / scale).max()))
        worst = max(worst, err)
        checked += W.shape[0]
    assert checked > min_channels                        # every output channel of every convolution
    # run to run: same batch, same drop_connect draws -> the same bits (no result is joined through fp atomics)
    grads = torch.cat([w.grad.reshape(-1) for w, _, _ in _pairs(model, cfg, N, size)]).clone()
    model.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    out2 = muscle_amd.mcl_step(model, opt, batch, 4, valid_channel=vc)
    for k, v in out2.items():
        v = float(v.detach()) if torch.is_tensor(v) else float(v)
        assert abs(v - losses[k]) <= 1e-4 * max(1.0, abs(losses[k])), (k, v, losses[k])
    return model, cfg, losses, out2, grads


def test_b7_448_bs32_scale_invariance_identity():
    scale_invariance_identity("efficientnet-b7", 32, 448, 100000)


def test_er_loss_fused_vs_materialised_448_bs32():
    """Two independent HIP implementations of train_mcl.py:175-188 at the headline size: the fused low-resolution kernels
    (upsample + softmax-norm recomputed per pixel, band backward) against the kernels that work on materialised
    [N,21,448,448] maps produced by the public upsample op.  Loss and the gradient w.r.t. the 1/16 SGC must agree."""
    from muscle_amd import ops, synth
    from muscle_amd.train_step import er_loss, er_loss_lowres
    N, K, h, w, H, W = 32, 21, 28, 28, 448, 448
    g = torch.Generator(device="cpu").manual_seed(3)
    cam = torch.zeros(N, h, w, 24)
    sgc = torch.zeros(N, h, w, 24)
    cam[..., :K] = torch.rand(N, h, w, K, generator=g) * 3
    sgc[..., :K] = torch.rand(N, h, w, K, generator=g) * 3
    cam, sgc = cam.to(DEV), sgc.to(DEV).requires_grad_(True)
    label = torch.from_numpy(synth.synth_labels(N, 5)).to(DEV)
    lwb = torch.cat([torch.ones(N, 1, device=DEV), label], dim=1)
    vc = int(label.sum().item())
    l1 = er_loss_lowres(cam, sgc, lwb, vc, H, W)
    l1.backward()
    g1 = sgc.grad.clone()
    sgc2 = sgc.detach().clone().requires_grad_(True)

    class _Up(torch.autograd.Function):               # public upsample op with its adjoint
        @staticmethod
        def forward(ctx, x):
            return ops.upsample_to_nchw(x, K, H, W)

        @staticmethod
        def backward(ctx, gy):
            gx = torch.zeros(N, h, w, 24, device=DEV)
            ops.upsample_to_nchw_bwd(gy.contiguous(), gx, accumulate=False)
            return gx

    l2 = er_loss(_Up.apply(cam), _Up.apply(sgc2), lwb, vc)
    l2.backward()
    assert abs(float(l1.detach()) - float(l2.detach())) <= 2e-5 * abs(float(l2.detach())), (float(l1.detach()), float(l2.detach()))
    a, b = g1[..., :K].double().flatten(), sgc2.grad[..., :K].double().flatten()
    # every low-res gradient entry is a signed sum over ~1100 full-resolution pixels, accumulated in a different order
    # by the two implementations (register/LDS/atomic bands vs a separable LDS scatter): compare in norm
    rel = float((a - b).norm() / b.norm())
    cos = float(a @ b / (a.norm() * b.norm()))
    assert rel <= 2e-3 and cos >= 0.999995, (rel, cos)


def test_fused_adam_matches_torch_on_the_b7_arena():
    """FusedAdam (one streaming kernel over the flat arenas) against torch.optim.Adam on the full 62 M-parameter set,
    three steps with weight decay."""
    import muscle_amd
    torch.manual_seed(0)
    model = muscle_amd.MuSCLe(21, "efficientnet-b7", layers=3, last_pooling=False).to(DEV)
    ps = [p for p in model.parameters()]
    ref = [p.detach().clone().requires_grad_(True) for p in ps]
    opt = muscle_amd.FusedAdam(ps, lr=1e-3, weight_decay=5e-5)
    opt_ref = torch.optim.Adam(ref, lr=1e-3, weight_decay=5e-5)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for _ in range(3):
        for p, r in zip(ps, ref):
            g = torch.randn(p.shape, device=DEV, generator=gen) * 1e-2
            p.grad = g.clone()
            r.grad = g.clone()
        opt.step()
        opt_ref.step()
    worst = max(float((p.detach() - r.detach()).abs().max()) for p, r in zip(ps, ref))
    assert worst <= 2e-6, worst


def test_b7_decoder_448_bs16_scale_invariance_identity():
    """BASELINE.json configs[3] size (decoder mode, B7, 448x448, batch 16): the same identity over the backbone's conv -> BN
    pairs and the BiFPN's Conv2d(+bias) -> BatchNorm2d pairs (for a biased conv the invariant scaling is of (W, b) jointly:
    <dW[c], W[c]> + db[c]*b[c] = eps * rstd[c]^2 * gamma[c] * dgamma[c]), on one train_muscle.py step (cross entropy)."""
    import muscle_amd
    from muscle_amd import arch, synth
    name, N, size = "efficientnet-b7", 16, 448
    cfg = arch.net_cfg(name, True)
    torch.manual_seed(0)
    model = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=True, mode="dec").to(DEV)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    opt = muscle_amd.FusedAdam(model.parameters(), lr=0.0, weight_decay=0.0)
    label = synth.synth_labels(N, 7)
    batch = {"img": torch.from_numpy(synth.normal(7, "img", (N, 3, size, size)).astype(np.float32)).to(DEV),
             "label": torch.from_numpy(label).to(DEV),
             "mask": torch.from_numpy(synth.synth_soft_mask(label, size, 7)).to(DEV)}
    import random
    random.seed(3)
    out = muscle_amd.muscle_step(model, opt, batch, lamb=0.05, k=8)     # with the BEACON term at 448x448 / batch 16 (a random-init
    # net has few pixels above 0.8*max of its edge map: k = 8 points per side so that classes qualify, edge.py:296)
    assert np.isfinite(float(out["loss_seg"].detach()))
    assert torch.is_tensor(out["loss_beacon"]) and np.isfinite(float(out["loss_beacon"].detach()))
    pairs = [(w, None, bn, n) for w, bn, n in _pairs(model, cfg, N, size)]
    for i in range(3, 8):
        seq = getattr(model.BIFPN, f"inp{i}")
        pairs.append((seq[0].weight, seq[0].bias, seq[1], None))
    for layer in model.BIFPN.BIFPN_Layers:
        for nm in ("out4", "out5", "out6", "out7"):
            seq = getattr(layer, nm)
            pairs.append((seq[0].weight, seq[0].bias, seq[1], None))
    checked = 0
    for w, b, bn, n in pairs:
        if w.grad is None:                               # dead branches of the last BiFPN layer (as in the reference)
            assert bn.weight.grad is None
            continue
        W, G = w.detach().double().flatten(1), w.grad.double().flatten(1)
        lhs = (W * G).sum(1)
        if b is not None:
            lhs = lhs + b.detach().double() * b.grad.double()
        # BiFPN maps (n is None): the unbiased -> biased factor (n-1)/n is within 2e-3 of 1 (n >= 16*7*7) and is absorbed
        # by the looser tolerance below
        var_b = bn.running_var.double() * ((n - 1) / n if n is not None else 1.0)
        rhs = bn.eps / (var_b + bn.eps) * bn.weight.detach().double() * bn.weight.grad.double()
        scale = W.norm(dim=1) * G.norm(dim=1) + (0 if b is None else (b.detach().double() * b.grad.double()).abs()) + 1e-30
        err = ((lhs - rhs).abs() / scale).max().item()
        assert err <= (5e-5 if n is not None else 2e-3), (tuple(w.shape), err)
        checked += W.shape[0]
    assert checked > 60000
