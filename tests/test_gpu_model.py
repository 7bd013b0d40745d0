"""GPU parity of the full model (backbone + CAM/PCM head), the phase-1 loss kernels and the loop body, through the
public muscle_amd API, against (a) the CPU oracle on the same seeded inputs and (b) the fixtures produced by the
reference itself (tests/golden).  fp32 tolerances as stated per assert (losses rel <= 1e-4,
CAM/SGC max-abs <= 1e-3 max|ref|, gradients max-abs <= 2e-3 of the tensor scale, cosine >= 0.9999)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from muscle_amd import synth
from muscle_amd.arch import net_cfg

pytestmark = [pytest.mark.gpu, pytest.mark.both_arith]
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.asarray(a))  # noqa: E731


def build(name, seed):
    import muscle_amd
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    m = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False)
    m.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    return cfg, sd, m.to(DEV)


def close(a, b, tol):
    a = a.detach().cpu() if torch.is_tensor(a) else a
    e = gu.rel_err(a, b)
    assert e <= tol, e


def check_grads(model, F, tol=2e-3, cache=None):
    """Parameter gradients against the stored fp64 oracle pass (tests/golden_util.py::check_grads_fixture)."""
    return gu.check_grads_fixture({k: p.grad for k, p in model.named_parameters()}, F, tol, cache)


MODEL_CASES = [("efficientnet-b0", 3, 64, "cam", True), ("efficientnet-b0", 2, 96, "pix", False),
               ("efficientnet-b3", 2, 72, "cam", True), ("efficientnet-b7", 2, 64, "cam", True),
               ("efficientnet-b0", 3, 64, "logits", True)]
_PROBES = {}                                     # projection vectors on the device, shared by the two arithmetics of a case


def run_case(kind, name, *rest):
    return "_".join([kind, name.replace("efficientnet-", "")] + [str(int(r)) if isinstance(r, bool) else str(r) for r in rest])


@pytest.mark.parametrize("name,n,size,mode,training", MODEL_CASES)
def test_model_forward_backward(name, n, size, mode, training):
    """The CPU oracle's fp32 + fp64 passes for these seeded cases are stored (oracle/gen_oracle_runs.py ->
    tests/golden/oracle_runs; the CPU suite re-runs the small ones against the oracle at HEAD): on the GPU box's host
    share they cost 110 s for the B7 case alone."""
    seed = 23
    cfg, sd, model = build(name, seed)
    x = T(synth.normal(seed, "x", (n, 3, size, size)).astype(np.float32))
    du = gu.drop_draws(cfg, n, 5)
    F = gu.load_run(run_case("model", name, n, size, mode, training))
    cache = _PROBES.setdefault((name, n, size, mode, training), {})
    model.train() if training else model.eval()
    got = model(x.to(DEV), cam=mode, drop_u={k: v.to(DEV) for k, v in du.items()})
    gu.check_outputs(got, F, "out", 5e-4, cache)
    loss = sum((g * T(synth.normal(seed, f"probe{i}", tuple(g.shape)).astype(np.float32)).to(DEV)).sum()
               for i, g in enumerate(got))
    loss.backward()
    print("worst grad rel err", check_grads(model, F, cache=cache))


def test_cpu_input_raises():
    import muscle_amd
    _, _, model = build("efficientnet-b0", 1)
    with pytest.raises(muscle_amd._lib.MuscleHipError):
        model(torch.zeros(1, 3, 32, 32))


# ---- loss kernels against the reference's own outputs (tests/golden/units.npz) ---------------------------
U = gu.load("units.npz")
SEED = 3


def test_cls_losses_golden():
    import muscle_amd as M
    lab = T(synth.synth_labels(6, SEED)).to(DEV)
    logit = (T(synth.normal(SEED, "logit", (6, 20)).astype(np.float32)) * 2).to(DEV).requires_grad_()
    p = M.loss_multilabel.sigmoid(logit)
    l1, l2 = M.FocalLoss()(p, lab), M.MultiLabelSoftMarginLoss()(logit, lab)
    l3 = M.Log_Sum_Exp_Pairwise_Loss(p, lab)
    (l1 + l2 + l3.mean()).backward()
    close(torch.stack([l1, l2]), U["cls_losses"], 1e-5); close(l3, U["cls_pair"], 1e-5); close(logit.grad, U["cls_dlogit"], 2e-5)


def test_imc_golden():
    import muscle_amd as M
    emb = T(synth.normal(SEED, "imc.emb", (8, 48)).astype(np.float32)).to(DEV).requires_grad_()
    li = M.image_level_contrast(emb, T(synth.synth_labels(8, SEED + 1)).to(DEV))
    assert torch.is_tensor(li) == bool(U["imc_is_tensor"])
    li.backward()
    close(li, U["imc"], 2e-5); close(emb.grad, U["imc_demb"], 5e-5)
    l0 = M.image_level_contrast(T(synth.normal(SEED, "imc.emb0", (4, 48)).astype(np.float32)).to(DEV), torch.ones(4, 20, device=DEV))
    assert not torch.is_tensor(l0) and l0 == 0.0
    ln, info = M.loss_multilabel.image_level_contrast_nosync(emb, torch.ones(8, 20, device=DEV))
    assert float(ln) == 0.0 and float(info[1]) == 0.0


@pytest.mark.parametrize("n,d", [(32, 640), (37, 320), (64, 1024), (16, 48), (7, 36)])
def test_imc_mfma_vs_oracle(n, d):
    """IMC at the headline size (32 x 640: E E^T and the gradient's second product on v_mfma_f32_16x16x4_f32, one workgroup per
    16 anchor rows) and on ragged / maximal shapes, against the oracle's closed form (loss_multilabel.py:36-66) incl. its
    gradient; (7, 36) takes the one-workgroup kernel (D % 16 != 0).  Bit-identical from run to run."""
    import muscle_amd as M
    from oracle import mcl_oracle as O
    emb = T(synth.normal(SEED + n, "imc.big", (n, d)).astype(np.float32))
    lab = T(synth.synth_labels(n, SEED + d))
    lab[n // 2] = lab[0]                                   # at least one identical-label pair
    e_ref = emb.clone().requires_grad_()
    want = O.image_level_contrast(e_ref, lab)
    assert torch.is_tensor(want)
    want.backward()
    outs = []
    for _ in range(2):
        e = emb.to(DEV).requires_grad_()
        li = M.image_level_contrast(e, lab.to(DEV))
        li.backward()
        outs.append((li.detach().clone(), e.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    close(outs[0][0], want.detach().numpy(), 2e-5); close(outs[0][1], e_ref.grad.numpy(), 5e-5)


def test_softmaxnorm_golden():
    import muscle_amd as M
    cam = T(synth.normal(SEED, "cam", (2, 21, 9, 11)).astype(np.float32)).to(DEV).requires_grad_()
    o = M.cam_softmaxnorm(cam)
    close(o, U["cam_softmaxnorm"], 1e-5)
    # backward against torch autograd of the oracle's restatement
    from oracle import mcl_oracle as O
    g = T(synth.normal(SEED, "cam.g", (2, 21, 9, 11)).astype(np.float32))
    o.backward(g.to(DEV))
    c2 = cam.detach().cpu().requires_grad_()
    O.cam_softmaxnorm(c2).backward(g)
    close(cam.grad, c2.grad, 2e-5)


def test_er_golden():
    # the golden ER case feeds *already normalised* maps through the reference expression; reproduce it through
    # the fused kernel by inverting the softmaxnorm: raw = log(fg) gives softmax(raw[1:]) = fg / sum fg, so instead
    # compare the fused kernel with the oracle on raw maps, and the top-k machinery with torch.topk directly.
    import muscle_amd as M
    from oracle import mcl_oracle as O
    n, h, w = 3, 8, 8
    rc = T(synth.normal(SEED, "er.rc", (n, 21, h, w)).astype(np.float32))
    rs = T(synth.normal(SEED, "er.rs", (n, 21, h, w)).astype(np.float32)).requires_grad_()
    lab = T(synth.synth_labels(n, SEED + 2))
    lwb = torch.cat((torch.ones(n, 1), lab), 1)
    vc = int(lab.sum())
    ref = O.er_loss(O.cam_softmaxnorm(rc).detach(), O.cam_softmaxnorm(rs), lwb, vc)
    ref.backward()
    rsg = rs.detach().to(DEV).requires_grad_()
    got = M.er_loss(rc.to(DEV), rsg, lwb.to(DEV), vc)
    got.backward()
    close(got, float(ref), 1e-5); close(rsg.grad, rs.grad, 2e-5)
    # k larger than the number of non-zero entries (the production regime) and k > row length (must raise)
    big = M.er_loss(rc.to(DEV), rsg, lwb.to(DEV), 40)
    refbig = O.er_loss(O.cam_softmaxnorm(rc), O.cam_softmaxnorm(rs), lwb, 40)
    close(big, float(refbig), 1e-5)
    with pytest.raises(RuntimeError):
        M.er_loss(rc.to(DEV), rsg, lwb.to(DEV), 200)


def test_adam_golden():
    import muscle_amd as M
    w = torch.nn.Parameter(T(synth.normal(SEED, "adam.w", (33,)).astype(np.float32)).to(DEV))
    w2 = torch.nn.Parameter(T(synth.normal(SEED, "adam.w2", (5,)).astype(np.float32)).to(DEV))
    o = M.FusedAdam([w, w2], lr=1e-4, weight_decay=5e-5)
    for stp in range(3):
        w.grad = T(synth.normal(SEED, f"adam.g{stp}", (33,)).astype(np.float32)).to(DEV)
        w2.grad = None if stp == 1 else T(synth.normal(SEED, f"adam.h{stp}", (5,)).astype(np.float32)).to(DEV)
        o.step()
        close(torch.cat([w.detach(), w2.detach()]), U["adam_traj"][stp], 1e-6)


# ---- the loop body against the reference's own step outputs ---------------------------------------------
PHASE1 = ["step_b0_ep0.npz", "step_b0_ep4.npz", "step_b0_ep4_imc0.npz", "step_b7_ep4.npz", "step_b7_448_ep4.npz"]


def test_er_lowres_matches_fullres():
    """The fused low-resolution ER kernels against the materialised path (public forward + full-res ER kernel)."""
    import muscle_amd as M
    from muscle_amd.train_step import er_loss_lowres
    cfg, sd, model = build("efficientnet-b0", 9)
    n, size = 3, 80
    x = T(synth.normal(9, "x", (n, 3, size, size)).astype(np.float32)).to(DEV)
    lab = T(synth.synth_labels(n, 9)).to(DEV)
    lwb = torch.cat((torch.ones(n, 1, device=DEV), lab), 1)
    du = {k: v.to(DEV) for k, v in gu.drop_draws(cfg, n, 5).items()}
    model.train()
    cams, sgc, _, _ = model(x, cam="cam", drop_u=du)
    l_full = M.er_loss(cams.detach(), sgc, lwb, int(lab.sum()))
    model.zero_grad(); l_full.backward()
    g_full = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    cfg, sd, model2 = build("efficientnet-b0", 9)
    model2.train()
    cam_lr, sgc_lr, _, _ = model2(x, cam="cam_lr", drop_u=du)
    l_lr = er_loss_lowres(cam_lr.detach(), sgc_lr, lwb, int(lab.sum()), size, size)
    l_lr.backward()
    close(l_lr, float(l_full), 2e-5)
    for k, p in model2.named_parameters():
        if k in g_full:
            sc = float(g_full[k].abs().max())
            if sc > 1e-7:
                assert float((p.grad - g_full[k]).abs().max()) <= 2e-3 * sc + 1e-9, k


@pytest.mark.parametrize("fname", PHASE1)
@pytest.mark.parametrize("imc_sync", [False, True])
def test_mcl_step_phase1_golden(fname, imc_sync):
    import muscle_amd as M
    G = gu.load(fname)
    name = str(G["name"]); n, size, view, ep, seed, tseed = (int(v) for v in G["meta"])
    cfg, sd, model = build(name, seed)
    opt = M.FusedAdam(model.parameters(), lr=float(G["lr"]), weight_decay=5e-5)
    b = {k: T(v).to(DEV) for k, v in synth.synth_batch(n, size, view, seed).items()}
    du = {int(i): T(u).to(DEV) for i, u in zip(G["drop_idx"], G["drop_u"])}
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    out = M.mcl_step(model, opt, b, ep, drop_u=du, imc_sync=imc_sync)
    names = ("loss_focal", "loss_softmargin", "loss_pair", "loss_er", "loss_imc", "loss_pixpro", "loss_emd")
    got = np.array([float(out[k]) for k in names])
    assert np.all(np.abs(got - G["losses"]) <= 1e-4 * np.maximum(np.abs(G["losses"]), 1e-3)), (got, G["losses"])
    if imc_sync:
        assert torch.is_tensor(out["loss_imc"]) == bool(G["loss_is_tensor"][0]) or ep < 4
    keys = [str(k) for k in G["param_keys"]]
    named = dict(model.named_parameters())
    assert keys == list(named.keys())
    g = gu.tensor_summary([(k, named[k].grad) for k in keys])
    ref = G["grad1"]
    assert np.array_equal(np.isnan(g[:, 0]), np.isnan(ref[:, 0]))          # same set of parameters without gradient
    live = ~np.isnan(ref[:, 0])
    scale = np.maximum(ref[live, :1], 1e-3 * ref[live, 0].max())
    assert np.all(np.abs(g[live] - ref[live]) <= 3e-3 * scale), np.abs((g[live] - ref[live]) / scale).max()
    # Adam: parameters without gradient untouched; the others moved by ~lr (first step = lr * sign(g))
    for k in keys:
        d = (named[k].detach() - before[k]).abs().max().item()
        if np.isnan(ref[keys.index(k), 0]):
            assert d == 0.0, k
        else:
            assert d <= 1.01 * float(G["lr"]) + 1e-9, (k, d)
    dl = gu.tensor_summary([(k, named[k].detach() - before[k]) for k in keys])
    refd = G["delta1"]
    big = live & (ref[:, 0] > 1e-2 * ref[live, 0].max())                    # well-conditioned tensors: same update
    assert np.all(np.abs(dl[big, 0] - refd[big, 0]) <= 2e-2 * refd[big, 0])
    bn = np.array([[float(v.double().sum()), float(model.state_dict()[k.replace("running_mean", "running_var")].double().sum())]
                   for k, v in model.state_dict().items() if k.endswith("running_mean")])
    close(bn, G["bn_after"], 1e-4)


def test_gradient_accumulation_keeps_arena_aliasing():
    """Two backward calls without zero_grad(): p.grad must hold the sum AND stay a view of model.last_grad_sink.arena,
    because the data-parallel average and the clip act on the arena only; a head whose output is unused keeps grad None
    (autograd semantics); a gradient left outside the arena makes GradAverager / clip raise instead of silently skipping it."""
    import muscle_amd
    from muscle_amd.dist import GradAverager
    cfg, sd, model = build("efficientnet-b0", 3)
    n, size = 2, 64
    x = T(synth.normal(3, "x", (n, 3, size, size)).astype(np.float32)).to(DEV)
    du = {k: v.to(DEV) for k, v in gu.drop_draws(cfg, n, 5).items()}
    model.train()

    def one_backward(mode="cam"):
        outs = model(x, cam=mode, drop_u=du)
        sum((o * o).sum() for o in outs).backward()

    one_backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    one_backward()                                            # accumulate
    arena = model.last_grad_sink.arena
    lo, hi = arena.data_ptr(), arena.data_ptr() + 4 * arena.numel()
    gmax = max(float(g.abs().max()) for g in g1.values())
    for k, p in model.named_parameters():
        if p.grad is None:
            assert k not in g1
            continue
        assert lo <= p.grad.data_ptr() < hi, k
        # 2x up to fp32 round-off.  Parameters that a following train-mode BatchNorm cancels (e.g. _bn2.bias) have pure
        # round-off gradients that differ from run to run (atomics reorder sums): judged on the scale of the whole gradient
        sc = max(float(g1[k].abs().max()), 1e-3 * gmax)
        assert float((p.grad - 2 * g1[k]).abs().max()) <= 5e-3 * sc, k
    model.last_grad_sink.check_aliases(model)
    # single-output loss: the untouched heads keep grad None, like autograd
    model.zero_grad(set_to_none=True)
    cams, sgc, emb, logits = model(x, cam="cam", drop_u=du)
    (cams * cams).sum().backward()
    assert model.fc.weight.grad is None and model.fuse.weight.grad is None and model.fuse.bias.grad is None
    assert model.backbone._conv_stem.weight.grad is not None
    # a stale gradient from another forward mode sits outside the new arena -> the hook refuses to average
    model.zero_grad(set_to_none=True)
    one_backward("cam")                                       # fc.weight gets a gradient here ...
    keep = model.fc.weight.grad
    for k, p in model.named_parameters():
        if p is not model.fc.weight:
            p.grad = None
    one_backward("pix")                                       # ... but not here: it stays in the old arena
    assert model.fc.weight.grad is keep
    h = GradAverager()
    h.world = 2
    with pytest.raises(muscle_amd._lib.MuscleHipError, match="zero_grad"):
        h(model, 1)
