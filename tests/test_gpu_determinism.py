"""Run-to-run determinism of the training step (missing #1): the reference CPU path gives the same bits every
run (src/efficientnet_pytorch/model.py:81-82 SE squeeze, src/MuSCLe.py:240 GAP are plain sums); the build joined several
reductions through fp32 atomics until round 3.  Two fresh models fed the same batch must now agree BIT FOR BIT in every loss
term, in the whole gradient arena, in the BatchNorm buffers and in the parameters after the Adam step.  This is a
self-comparison: it runs last (tests/conftest.py)."""
import numpy as np
import pytest
import torch

import muscle_amd
from muscle_amd import synth
from muscle_amd.arch import net_cfg

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None
LOSSES = ("loss_focal", "loss_softmargin", "loss_pair", "loss_er", "loss_imc", "loss_pixpro", "loss_emd")


def _one_step(name, n, size, view, ep, seed, disturb):
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    model = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.to(DEV)
    opt = muscle_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=5e-5)
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synth_batch(n, size, view, seed).items()}
    torch.manual_seed(3)
    du = {b.index: torch.rand(n).to(DEV) for b in cfg.blocks if b.skip and b.drop_rate}
    if ep >= 8:
        # phase 2 runs in eval mode: give the random-init model running statistics of its own activations (as the fixtures do)
        from test_gpu_b7_golden import _calibrate
        _calibrate(model, cfg, batch["view1"], n)
    np.random.seed(5)                              # get_dynamic_crops draws its patch geometry from numpy's global generator
    bg = None
    if disturb:
        # a second stream keeps the chip busy with unrelated work, so workgroups of the step land in a different order
        bg = torch.cuda.Stream()
        with torch.cuda.stream(bg):
            junk = torch.randn(4096, 4096, device=DEV)
            for _ in range(20):
                junk = junk * 1.0001 + 0.5
    out = muscle_amd.mcl_step(model, opt, batch, ep, drop_u=du)
    torch.cuda.synchronize()
    losses = np.array([float(out[k]) for k in LOSSES], dtype=np.float32)
    arena = model.last_grad_sink.arena.cpu().numpy().copy()
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
    bufs = torch.cat([b.detach().reshape(-1).float() for b in model.buffers()]).cpu().numpy()
    return losses, arena, params, bufs


@pytest.mark.both_arith            # exact fp32 AND the library's default arithmetic (split kernels, weight planes)
@pytest.mark.parametrize("name,n,size,view,ep", [("efficientnet-b0", 4, 64, 32, 4), ("efficientnet-b3", 4, 96, 48, 4),
                                                  ("efficientnet-b0", 4, 96, 64, 12)])        # ep 12: + PixPro + EMD, second backward + Adam
def test_step_is_bit_reproducible(name, n, size, view, ep):
    a = _one_step(name, n, size, view, ep, 5, disturb=False)
    b = _one_step(name, n, size, view, ep, 5, disturb=True)
    c = _one_step(name, n, size, view, ep, 5, disturb=False)
    for tag, x, y, z in zip(("losses", "gradient arena", "parameters", "buffers"), a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z), (tag, float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()))


def test_eval_forward_is_bit_reproducible_and_batch_invariant():
    """Inference (BASELINE.json configs[4]): the folded eval forward has no batch statistics, so a row of a batch equals the
    same image run alone, bit for bit, now that the SE squeeze / GAP sums are joined in a fixed order."""
    name, seed = "efficientnet-b3", 11
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    model = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.to(DEV).eval()
    x = torch.from_numpy(synth.normal(seed, "x", (6, 3, 160, 128)).astype(np.float32)).to(DEV)
    with torch.no_grad():
        full = [t.clone() for t in model(x, cam="cam_lr")]
        again = model(x, cam="cam_lr")
        one = model(x[4:5], cam="cam_lr")
    for f, g in zip(full, again):
        assert torch.equal(f, g)
    for f, o in zip(full, one):
        assert torch.equal(f[4:5], o)


@pytest.mark.parametrize("arith", [0, 1])
def test_headline_step_is_bit_reproducible(arith):
    """BASELINE.json configs[2]'s workload (B7, 448x448, batch 32), where the large-shape kernel variants run (two-level BatchNorm
    reductions, tiled / small-output weight gradients with row groups, the band ER backward): two steps on the same batch with
    lr = 0 must give the same loss bits and the same gradient arena, in both GEMM arithmetics."""
    import bench
    name, N, size = "efficientnet-b7", 32, 448
    torch.manual_seed(0)
    model = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False).to(DEV)
    opt = muscle_amd.FusedAdam(model.parameters(), lr=0.0, weight_decay=0.0)
    batch = bench.make_batch(N, size, size // 2, 1234, DEV)
    muscle_amd.set_gemm_mode(arith)
    try:
        runs = []
        for _ in range(2):
            torch.manual_seed(1)                       # same drop_connect draws
            out = muscle_amd.mcl_step(model, opt, batch, 4, valid_channel=batch["label"].sum())
            torch.cuda.synchronize()
            runs.append((np.array([float(out[k]) for k in LOSSES], dtype=np.float32), model.last_grad_sink.arena.clone()))
    finally:
        muscle_amd.set_gemm_mode(0)
    assert np.array_equal(runs[0][0], runs[1][0]), (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1]), float((runs[0][1].double() - runs[1][1].double()).abs().max())


def test_decoder_step_is_bit_reproducible():
    """train_muscle.py's loop body (decoder mode, cross entropy + BEACON FieldLoss with replayed boundary points, clip, Adam) on the
    fused path: two runs, same bits in both losses, the gradient norm, the gradient arena and the parameters."""
    import random
    import golden_util as gu
    from test_gpu_decoder import build_dec
    G = gu.load("muscle_step_b3_beacon.npz")
    name = str(G["name"]); n, size, seed, tseed, kk, step = (int(v) for v in G["meta"])
    T = lambda a: torch.from_numpy(np.asarray(a))   # noqa: E731
    runs = []
    for disturb in (False, True):
        cfg, sd, model = build_dec(name, seed)
        lab = synth.synth_labels(n, seed)
        b = {"img": T(synth.normal(seed, "img", (n, 3, size, size)).astype(np.float32)).to(DEV), "label": T(lab).to(DEV),
             "mask": T(synth.synth_soft_mask(lab, size, seed)).to(DEV)}
        opt = muscle_amd.FusedAdam(model.parameters(), lr=float(G["lr"]), weight_decay=1e-5)
        du = {int(i): T(u).to(DEV) for i, u in zip(G["drop_idx"], G["drop_u"])}
        random.seed(78)
        crit = muscle_amd.edge.FieldLoss(sobel_size=5, beta=1e2, k=kk)
        crit.replay_points = (G["replay_b"], G["replay_out"], G["replay_in"])
        if disturb:
            bg = torch.cuda.Stream()
            with torch.cuda.stream(bg):
                junk = torch.randn(4096, 4096, device=DEV)
                for _ in range(20):
                    junk = junk * 1.0001 + 0.5
        out = muscle_amd.muscle_step(model, opt, b, lamb=float(G["lamb"]), step=step, k=kk, drop_u=du, fused=True, criterion2=crit)
        torch.cuda.synchronize()
        runs.append((np.array([float(out["loss_seg"]), float(out["loss_beacon"]), float(out["grad_norm"])], dtype=np.float32),
                     model.last_grad_sink.arena.cpu().numpy().copy(),
                     torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()))
    for tag, x, y in zip(("losses + norm", "gradient arena", "parameters"), *runs):
        assert np.array_equal(x, y), (tag, float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()))


@pytest.mark.parametrize("n,hw,c", [(8, 112 * 112, 288), (4, 224 * 224, 32), (2, 56 * 56, 480)])
def test_ordered_pooled_sums_multi_block_join_under_load(n, hw, c):
    """The "last arriver finishes" hand-off of the per-sample reductions (csrc/common.h mx_last_arriver: write-through partials,
    one agent-scope counter add per workgroup, one acquire in the last one) at shapes where a sample IS cut into several row
    blocks (mx_pool_ws > 0 - the 64-96 px step tests mostly are not), with a second stream keeping the chip and the L2s busy:
    pool_sum and se_bn1_pool must give the same bits in every run, equal a float64 reference to fp32 round-off, and leave the
    arrival counters at zero."""
    from muscle_amd import ops
    from muscle_amd._lib import lib
    rows = n * hw
    assert lib().mx_pool_ws(rows, c, hw, 1) > 0 and lib().mx_pool_ws(rows, c, hw, 5) > 0        # several row blocks per sample
    g = torch.Generator(device=DEV).manual_seed(n + c)
    X = torch.randn(rows, c, device=DEV, generator=g)
    dA = torch.randn(rows, c, device=DEV, generator=g)
    st = ops.BNState(torch.rand(c, device=DEV, generator=g) + 0.5, torch.randn(c, device=DEV, generator=g) * 0.2, None, None)
    side = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device=DEV)
    outs = []
    for rep in range(4):
        if rep % 2:
            with torch.cuda.stream(side):
                for _ in range(30):
                    junk = junk * 1.0001 + 0.5
        p1 = ops.pool_sum(X, hw, st=st, act=True)
        p5 = ops.se_bn1_pool(dA, X, st, hw)
        torch.cuda.synchronize()
        outs.append((p1.clone(), p5.clone()))
    for p1, p5 in outs[1:]:
        assert torch.equal(p1, outs[0][0]) and torch.equal(p5, outs[0][1])
    z = X.double() * st.scale.double() + st.shift.double()
    want = (z * torch.sigmoid(z)).view(n, hw, c).sum(1)
    err = float((outs[0][0].double() - want).abs().max() / want.abs().max())
    assert err <= 2e-6, err
    for key, buf in ops._scratch_bufs.items():
        assert int(buf[:65536].view(torch.int32).abs().sum()) == 0, key          # every launch leaves its counters at zero
