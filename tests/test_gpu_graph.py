"""hipGraph replay of phase 1 of the MCL step (muscle_amd.GraphedStep) against the eager path on the same batches:
same losses, same parameters after several optimizer steps, with the label set (hence ER's top-k count, now taken on
the device) changing from batch to batch; and the device-scalar mode of FusedAdam against the host-scalar mode."""
import numpy as np
import pytest
import torch

import golden_util as gu
from muscle_amd import synth
from muscle_amd.arch import net_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.asarray(a))  # noqa: E731


def build(name, seed):
    import muscle_amd
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    m = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False)
    m.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    return m.to(DEV)


def batches(n, size, count, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(count):
        lab = torch.zeros(n, 20)
        for r in range(n):
            lab[r, torch.randperm(20, generator=g)[: 1 + (i + r) % 3]] = 1.0       # 1..3 classes, varies per batch
        lab[1] = lab[0]                                                                # a positive pair for IMC
        out.append({"img": torch.randn(n, 3, size, size, generator=g).to(DEV), "label": lab.to(DEV)})
    return out


def test_adam_device_scalars_match_host_scalars():
    import muscle_amd
    torch.manual_seed(3)
    ws = [torch.randn(1000, device=DEV), torch.randn(37, 5, device=DEV)]
    pa = [torch.nn.Parameter(w.clone()) for w in ws]
    pb = [torch.nn.Parameter(w.clone()) for w in ws]
    oa = muscle_amd.FusedAdam(pa, lr=1e-3, weight_decay=5e-5)
    ob = muscle_amd.FusedAdam(pb, lr=1e-3, weight_decay=5e-5).use_device_scalars(True)
    for t in range(7):
        if t == 4:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3e-4
            ob.sync_lr()
        for a, b in zip(pa, pb):
            g = torch.randn_like(a)
            a.grad, b.grad = g, g.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        # the bias corrections come from torch.pow on the device instead of Python's pow: equal after rounding to float
        # except for a possible last-place difference of the double result
        assert gu.rel_err(b.detach().cpu(), a.detach().cpu()) <= 1e-6
    assert ob.state_dict()["state"][0]["step"] == 7


@pytest.mark.both_arith            # the replay must equal the eager step in the library's default arithmetic too
@pytest.mark.parametrize("ep", [0, 4])
def test_graphed_step_matches_eager(ep):
    import muscle_amd
    n, size, count = 4, 64, 6
    bs = batches(n, size, count, 11)
    ref, gra = build("efficientnet-b0", 5), build("efficientnet-b0", 5)
    o_ref = muscle_amd.FusedAdam(ref.parameters(), lr=1e-4, weight_decay=5e-5)
    o_gra = muscle_amd.FusedAdam(gra.parameters(), lr=1e-4, weight_decay=5e-5)
    torch.manual_seed(99)
    want = []
    for b in bs:
        out = muscle_amd.mcl_step(ref, o_ref, b, ep)
        want.append({k: float(v) for k, v in out.items()})
    step = muscle_amd.GraphedStep(gra, o_gra, ep, warmup=2)
    torch.manual_seed(99)                       # same drop_connect draws: the graph takes its Philox offset at replay
    got = []
    for b in bs:
        out = step(b)
        got.append({k: float(v) for k, v in out.items()})
    assert step.replays == count - 2
    ers = [w["loss_er"] for w in want]
    assert max(ers) - min(ers) > 1e-4           # the batches do differ (k follows label.sum())
    for i, (w, g) in enumerate(zip(want, got)):
        for k in w:
            assert abs(w[k] - g[k]) <= 2e-4 * max(abs(w[k]), 1e-3), (i, k, w[k], g[k])
    pr, pg = dict(ref.named_parameters()), dict(gra.named_parameters())
    num = sum(float((pr[k].detach() - pg[k].detach()).double().pow(2).sum()) for k in pr)
    den = sum(float(pr[k].detach().double().pow(2).sum()) for k in pr)
    assert (num / den) ** 0.5 <= 1e-4
    for k, v in ref.state_dict().items():       # BatchNorm running statistics and batch counters advance in the replay
        if "num_batches_tracked" in k:
            assert int(v) == int(gra.state_dict()[k]), k           # (the unused conv_head BN stays at 0 in both)
    assert int(gra.state_dict()["backbone._bn0.num_batches_tracked"]) == count
    assert gu.rel_err(gra.state_dict()["backbone._bn0.running_var"].cpu(), ref.state_dict()["backbone._bn0.running_var"].cpu()) <= 1e-4
    sa, sb = o_ref.state_dict()["state"], o_gra.state_dict()["state"]
    assert {k: v["step"] for k, v in sa.items()} == {k: v["step"] for k, v in sb.items()}


def test_graphed_step_refuses_phase2_and_shape_change():
    import muscle_amd
    m = build("efficientnet-b0", 5)
    o = muscle_amd.FusedAdam(m.parameters(), lr=1e-4)
    with pytest.raises(ValueError):
        muscle_amd.GraphedStep(m, o, 8)
    step = muscle_amd.GraphedStep(m, o, 0, warmup=1)
    bs = batches(2, 64, 2, 1)
    step(bs[0]); step(bs[1])
    with pytest.raises(ValueError):
        step(batches(3, 64, 1, 2)[0])


@pytest.mark.parametrize("close_first", [False, True])
def test_graphed_phase1_then_eager_phase2(close_first):
    """The documented flow: GraphedStep for the ep < 8 iterations, mcl_step from ep 8 on with the SAME
    optimizer.  Phase 2 runs in 'pix' mode, which never gives fc.weight a gradient, so from the second ep-8 iteration on
    the live parameters no longer share one step count: an eager step must then take torch.optim.Adam's per-parameter
    scalars on the host instead of raising (round-2 advisor finding), whether or not GraphedStep.close() was called."""
    import muscle_amd
    name, n, size, view, seed = "efficientnet-b0", 4, 64, 32, 5
    full = {k: T(v).to(DEV) for k, v in synth.synth_batch(n, size, view, seed).items()}
    ref, gra = build(name, seed), build(name, seed)
    o_ref = muscle_amd.FusedAdam(ref.parameters(), lr=1e-4, weight_decay=5e-5)
    o_gra = muscle_amd.FusedAdam(gra.parameters(), lr=1e-4, weight_decay=5e-5)
    torch.manual_seed(7)
    for _ in range(3):
        muscle_amd.mcl_step(ref, o_ref, full, 7)
    for _ in range(2):
        muscle_amd.mcl_step(ref, o_ref, full, 8)
    step = muscle_amd.GraphedStep(gra, o_gra, 7, warmup=1)
    torch.manual_seed(7)
    for _ in range(3):
        step(full)
    assert step.replays == 2
    if close_first:
        step.close()
    for _ in range(2):
        out = muscle_amd.mcl_step(gra, o_gra, full, 8)          # second iteration: fc.weight is one step behind
    assert all(np.isfinite(float(v)) for v in out.values())
    sa, sb = o_ref.state_dict()["state"], o_gra.state_dict()["state"]
    assert {k: v["step"] for k, v in sa.items()} == {k: v["step"] for k, v in sb.items()}
    assert len({v["step"] for v in sb.values()}) == 2           # fc.weight: 4 steps, everything else live: 5
    pr, pg = dict(ref.named_parameters()), dict(gra.named_parameters())
    num = sum(float((pr[k].detach() - pg[k].detach()).double().pow(2).sum()) for k in pr)
    den = sum(float(pr[k].detach().double().pow(2).sum()) for k in pr)
    assert (num / den) ** 0.5 <= 1e-4


def test_graphed_step_follows_the_lr_scheduler_in_warmup():
    """A second GraphedStep on an optimizer whose device scalars already exist must run its eager warm-up steps with the
    CURRENT lr (round-2 advisor finding: the device-side lr was only refreshed right before a replay)."""
    import muscle_amd
    m = build("efficientnet-b0", 5)
    o = muscle_amd.FusedAdam(m.parameters(), lr=1e-4)
    bs = batches(2, 64, 3, 1)
    s1 = muscle_amd.GraphedStep(m, o, 0, warmup=1)
    s1(bs[0]); s1(bs[1])
    o.param_groups[0]["lr"] = 0.0                  # the scheduler's product at the epoch boundary
    before = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
    s2 = muscle_amd.GraphedStep(m, o, 4, warmup=1)
    s2(bs[2])                                      # eager warm-up step of the new gate: lr 0 -> parameters must not move
    after = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    assert torch.equal(before, after)


def test_repeated_steps_fit_a_fixed_batch():
    """Sanity of the whole loop (forward, losses, backward, side-stream weight gradients, fused Adam) beyond one step: on a
    fixed batch the classification losses must fall steadily - a wrong-signed or misrouted gradient anywhere shows here."""
    import muscle_amd
    torch.manual_seed(0)
    model = build("efficientnet-b0", 5)
    opt = muscle_amd.FusedAdam(model.parameters(), lr=1e-3, weight_decay=5e-5)
    b = batches(8, 64, 1, 3)[0]
    hist = []
    for it in range(30):
        out = muscle_amd.mcl_step(model, opt, b, 0)
        hist.append(float(out["loss_softmargin"].detach()) + float(out["loss_focal"].detach()) + float(out["loss_pair"].detach()))
    assert all(np.isfinite(hist)), hist
    assert hist[-1] < 0.6 * hist[0], (hist[0], hist[-1])
    assert np.mean(hist[-5:]) < np.mean(hist[:5])
