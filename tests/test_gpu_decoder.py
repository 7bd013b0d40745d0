"""GPU parity of config 4 (decoder mode): BiFPN seg forward/backward, cross entropy, gradient clipping, the BEACON
FieldLoss and the train_muscle loop body — against the oracle and the reference's own fixtures."""
import random

import numpy as np
import pytest
import torch

import golden_util as gu
from muscle_amd import synth
from muscle_amd.arch import net_cfg
from test_gpu_model import close, DEV, T, check_grads
from test_oracle_golden import field_unit_inputs

pytestmark = [pytest.mark.gpu, pytest.mark.both_arith]


def build_dec(name, seed):
    import muscle_amd
    cfg = net_cfg(name, True)
    sd = synth.synth_state_dict(cfg, seed, mode="dec", layers=3)
    m = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=True, mode="dec")
    m.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    return cfg, sd, m.to(DEV)


_PROBES = {}


def test_seg_forward_backward_vs_oracle():
    """Against the stored fp32 + fp64 passes of the CPU oracle (oracle/gen_oracle_runs.py::seg_run)."""
    from test_gpu_model import run_case
    name, n, size, seed = "efficientnet-b3", 2, 96, 31
    cfg, sd, model = build_dec(name, seed)
    x = T(synth.normal(seed, "x", (n, 3, size, size)).astype(np.float32))
    du = gu.drop_draws(cfg, n, 5)
    F = gu.load_run(run_case("seg", name, n, size, seed))
    model.train()
    got = model(x.to(DEV), cam="seg", drop_u={k: v.to(DEV) for k, v in du.items()})
    gu.check_outputs(got, F, "out", 5e-4, _PROBES)
    sum((g * T(synth.normal(seed, f"probe{i}", tuple(g.shape)).astype(np.float32)).to(DEV)).sum() for i, g in enumerate(got)).backward()
    print("worst grad rel err", check_grads(model, F, cache=_PROBES))
    rs = {str(k): F["rs_vals"][F["rs_off"][i]:F["rs_off"][i + 1]] for i, k in enumerate(F["rs_keys"])}
    seen = 0
    for k, v in model.state_dict().items():
        if k.endswith("running_var"):
            assert gu.rel_err(v.cpu().flatten(), rs[k]) <= 1e-4, k
            seen += 1
    assert seen == len(rs)


def test_seg_forward_golden():
    G = gu.load("seg_forward_b3.npz")
    name = str(G["name"]); n, size, seed = (int(v) for v in G["meta"])
    cfg, sd, model = build_dec(name, seed)
    x = T(synth.normal(seed, "fwd.x", (n, 3, size, size)).astype(np.float32)).to(DEV)
    du = {int(i): T(u).to(DEV) for i, u in zip(G["drop_idx"], G["drop_u"])}
    model.train()
    with torch.no_grad():
        seg, ft = model(x, cam="seg", drop_u=du)
    close(seg[:, :, ::4, ::4], G["seg_s4"], 5e-4); close(ft[:, :, ::8, ::8], G["ft_s8"], 5e-4)
    close([float(seg.double().sum()), float(seg.double().pow(2).sum())], G["seg_stats"], 5e-4)


def test_field_loss_golden():
    import muscle_amd as M
    F_, seg, ft, mask, lwb, kk, step = field_unit_inputs()
    crit = M.edge.FieldLoss(sobel_size=5, beta=1e2, k=kk)
    ftg = ft.to(DEV).requires_grad_()
    random.seed(77)
    loss, edge = crit(seg.to(DEV), ftg, mask.to(DEV), lwb.to(DEV), step)
    assert torch.is_tensor(loss)
    close(edge, F_["edge_fg"], 2e-5)
    loss.backward()
    close(loss, F_["loss"], 2e-5); close(ftg.grad, F_["dft"], 5e-5)
    l0, _ = crit(seg.to(DEV), ftg.detach(), mask.to(DEV), torch.cat((torch.ones(2, 1), torch.zeros(2, 20)), 1).to(DEV), step)
    assert l0 is False


def test_ce_and_clip():
    import muscle_amd as M
    seg = T(synth.normal(5, "ce.seg", (2, 21, 17, 19)).astype(np.float32))
    mask = T(synth.uniform(5, "ce.mask", (2, 21, 17, 19)).astype(np.float32))
    sg = seg.to(DEV).requires_grad_()
    l = M.edge.cross_entropy_argmax(sg, mask.to(DEV))
    l.backward()
    sc = seg.clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(sc, mask.argmax(1))
    ref.backward()
    close(l, float(ref), 1e-5); close(sg.grad, sc.grad, 2e-5)

    class Sink:
        pass

    class Mdl:
        pass
    for scale in (0.5, 40.0):                      # below / above the threshold of 9
        g = T(synth.normal(5, "clip.g", (1001,)).astype(np.float32)) * scale / 31.6
        m = Mdl(); m.last_grad_sink = Sink(); m.last_grad_sink.arena = g.to(DEV).clone()
        norm = M.edge.clip_grad_norm_(m, 9)
        tn = float(g.norm())
        close(norm, tn, 1e-5)
        close(m.last_grad_sink.arena, (g * min(1.0, 9 / (tn + 1e-6))).numpy(), 1e-5)


@pytest.mark.parametrize("fname", ["muscle_step_b3_ce.npz", "muscle_step_b3_beacon.npz"])
@pytest.mark.parametrize("fused", [True, False])
def test_muscle_step_golden(fname, fused):
    import muscle_amd as M
    G = gu.load(fname)
    name = str(G["name"]); n, size, seed, tseed, kk, step = (int(v) for v in G["meta"])
    cfg, sd, model = build_dec(name, seed)
    lab = synth.synth_labels(n, seed)
    b = {"img": T(synth.normal(seed, "img", (n, 3, size, size)).astype(np.float32)).to(DEV), "label": T(lab).to(DEV),
         "mask": T(synth.synth_soft_mask(lab, size, seed)).to(DEV)}
    opt = M.FusedAdam(model.parameters(), lr=float(G["lr"]), weight_decay=1e-5)
    du = {int(i): T(u).to(DEV) for i, u in zip(G["drop_idx"], G["drop_u"])}
    random.seed(78)
    crit = None
    if float(G["lamb"]) > 0:
        # the BEACON term on the reference's own boundary points (thresholding softmax(100*seg) at 0.8*max is decided by
        # fp32 round-off, so the sets themselves are replayed; what is compared is everything downstream of them)
        crit = M.edge.FieldLoss(sobel_size=5, beta=1e2, k=kk)
        crit.replay_points = (G["replay_b"], G["replay_out"], G["replay_in"])
    out = M.muscle_step(model, opt, b, lamb=float(G["lamb"]), step=step, k=kk, drop_u=du, fused=fused, criterion2=crit)
    close(out["loss_seg"], G["losses"][0], 1e-4)
    assert torch.is_tensor(out["loss_beacon"]) == bool(G["l2_is_tensor"])
    close(out["loss_beacon"], G["losses"][1], 1e-4)
    close(out["grad_norm"], G["grad_norm"][0], 2e-3)
    keys = [str(k) for k in G["param_keys"]]
    named = dict(model.named_parameters())
    assert keys == list(named.keys())
    ref = G["grad1"]
    g = gu.tensor_summary([(k, named[k].grad) for k in keys])
    assert np.array_equal(np.isnan(g[:, 0]), np.isnan(ref[:, 0]))          # dead branches of the last BiFPN layer: no grad
    if True:                                                              # lamb = 0 and lamb = 0.05 alike
        live = ~np.isnan(ref[:, 0])
        scale = np.maximum(ref[live, :1], 1e-3 * ref[live, 0].max())
        assert np.all(np.abs(g[live] - ref[live]) <= 3e-3 * scale), np.abs((g[live] - ref[live]) / scale).max()
        bn = np.array([[float(v.double().sum()), float(model.state_dict()[k.replace("running_mean", "running_var")].double().sum())]
                       for k, v in model.state_dict().items() if k.endswith("running_mean")])
        close(bn, G["bn_after"], 1e-4)
