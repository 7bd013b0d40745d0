"""GPU parity at the REAL sizes against values the reference itself produced (tests/golden, oracle/gen_golden.py
main_fullsize): EfficientNet-B7 at 448x448 (BASELINE.json configs[2]) and the CAM-generation forward of configs[4]
(infer_mcl.py:107-125: eval mode, B7, batch 64, 448 / 512 / 768) plus the two forward modes that had no GPU test
(cam='logits' is covered in test_gpu_model.py, cam='vis' here).

Tolerances (exact-fp32 mode): loss terms rel <= 1e-4; CAM / SGC max-abs <= 1e-3 * max|ref|;
emb / logits <= 5e-4 * max|ref|; gradient summaries <= 3e-3 of the tensor scale."""
import numpy as np
import pytest
import torch

import golden_util as gu
from muscle_amd import synth
from muscle_amd.arch import net_cfg
from test_gpu_model import build, close, DEV, T

pytestmark = [pytest.mark.gpu, pytest.mark.both_arith]


def _calibrate(model, cfg, x, n):
    """oracle/gen_golden.py::calibrate_bn on the HIP path: one train-mode 'pix' pass with momentum 1.0, drop draws of
    torch seed 7."""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    old = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(x, cam="pix", drop_u={k: v.to(DEV) for k, v in gu.drop_draws(cfg, n, 7).items()})
    for m, o in zip(bns, old):
        m.momentum = o


def _bn_summary(model):
    sd = model.state_dict()
    return np.array([[float(v.double().sum()), float(sd[k.replace("running_mean", "running_var")].double().sum())]
                     for k, v in sd.items() if k.endswith("running_mean")])


@pytest.mark.parametrize("fname", ["step_b7_ep4.npz", "step_b0_ep4.npz", "step_b7_448_ep4.npz"])
def test_train_forward_values_vs_reference_step(fname):
    """emb, logits, CAM and SGC of the train-mode forward the loop body starts with (train_mcl.py:173) against the
    reference's own tensors; at 448x448 this is the value check of the whole B7 forward chain at the headline size."""
    G = gu.load(fname)
    name = str(G["name"]); n, size, view, ep, seed, tseed = (int(v) for v in G["meta"])
    cfg, sd, model = build(name, seed)
    img = T(synth.synth_batch(n, size, view, seed)["img"]).to(DEV)
    du = {int(i): T(u).to(DEV) for i, u in zip(G["drop_idx"], G["drop_u"])}
    model.train()
    with torch.no_grad():
        cams, sgcs, emb, logits = model(img, cam="cam", drop_u=du)
    cs = gu.cam_stride(G)
    close(emb, G["emb"], 5e-4); close(logits, G["logits"], 5e-4)
    close(cams[:, :, ::cs, ::cs], G["raw_cams_s4"], 1e-3); close(sgcs[:, :, ::cs, ::cs], G["raw_sgcs_s4"], 1e-3)
    close([float(cams.double().sum()), float(cams.double().pow(2).sum())], G["raw_cams_stats"], 1e-3)
    close([float(sgcs.double().sum()), float(sgcs.double().pow(2).sum())], G["raw_sgcs_stats"], 1e-3)


@pytest.mark.parametrize("fname", ["forward_b7_eval_448.npz", "forward_b7_eval_563x750.npz"])
def test_eval_forward_b7_vs_reference(fname):
    """configs[4]'s forward (eval mode, BN-calibrated B7) in all three encoder modes, square and non-square odd size."""
    G = gu.load(fname)
    name = str(G["name"]); n, H, W, seed, st = (int(v) for v in G["meta"])
    cfg, sd, model = build(name, seed)
    x = T(synth.normal(seed, "fwd.x", (n, 3, H, W)).astype(np.float32)).to(DEV)
    _calibrate(model, cfg, x, n)
    close(_bn_summary(model), G["bn_calibrated"], 1e-4)
    model.eval()
    with torch.no_grad():
        cams, sgc, emb, logits = model(x, cam="cam")
        cams_p, sgc_p = model(x, cam="pix")
        emb_l, logits_l = model(x, cam="logits")
    assert tuple(cams.shape) == (n, 21, H, W) and tuple(sgc.shape) == (n, 21, H, W)
    close(cams[:, :, ::st, ::st], G["cams_s"], 1e-3); close(sgc[:, :, ::st, ::st], G["sgc_s"], 1e-3)
    close([float(cams.double().sum()), float(cams.double().pow(2).sum()), float(cams.abs().max())], G["cams_stats"], 1e-3)
    close([float(sgc.double().sum()), float(sgc.double().pow(2).sum()), float(sgc.abs().max())], G["sgc_stats"], 1e-3)
    close(emb, G["emb"], 5e-4); close(logits, G["logits"], 5e-4)
    # the three modes run the same kernels; only the SE / GAP pooling's fp32 atomics may reorder sums between two runs
    close(cams_p, cams.cpu(), 2e-5); close(sgc_p, sgc.cpu(), 2e-5)
    close(emb_l, emb.cpu(), 1e-5); close(logits_l, logits.cpu(), 1e-5)


@pytest.mark.parametrize("size", [448, 512, 768])
def test_config5_b7_bs64_batch_invariance(size):
    """BASELINE.json configs[4]: B7 eval forward at batch 64.  Eval mode has no batch statistics, so every row of the
    batch-64 result must equal the same image run alone; at 448 the first two rows are the reference fixture's images
    and must match its values.  The low-resolution maps the CAM-generation path consumes (cam='cam_lr') are checked too."""
    G = gu.load("forward_b7_eval_448.npz")
    name = str(G["name"]); n0, H0, W0, seed, st = (int(v) for v in G["meta"])
    cfg, sd, model = build(name, seed)
    x0 = T(synth.normal(seed, "fwd.x", (n0, 3, H0, W0)).astype(np.float32)).to(DEV)
    _calibrate(model, cfg, x0, n0)
    model.eval()
    model.fold_eval_bn()                      # the CAM-generation setting: BatchNorms folded once per model load
    B = 64
    x = T(synth.normal(seed + 1, "cfg5.x", (B, 3, size, size)).astype(np.float32)).to(DEV)
    if size == H0:
        x[:n0] = x0
    with torch.no_grad():
        cam_lr, sgc_lr, emb, logits = model(x, cam="cam_lr")
        rows = [0, 1, 31, 63]
        cams, sgc, emb2, logits2 = model(x[rows], cam="cam")
    assert bool(torch.isfinite(cam_lr).all()) and bool(torch.isfinite(sgc_lr).all()) and bool(torch.isfinite(logits).all())
    if size == H0:
        close(cams[:n0, :, ::st, ::st], G["cams_s"], 1e-3); close(sgc[:n0, :, ::st, ::st], G["sgc_s"], 1e-3)
        close(emb[:n0], G["emb"], 5e-4); close(logits[:n0], G["logits"], 5e-4)
    close(emb2, emb[rows].cpu(), 1e-5); close(logits2, logits[rows].cpu(), 1e-5)
    for r in rows:
        with torch.no_grad():
            c1, s1, e1, l1 = model(x[r:r + 1], cam="cam_lr")
        # same arithmetic per row whatever the batch; only the SE / GAP pooling's partial-sum order may move
        close(c1, cam_lr[r:r + 1].cpu(), 2e-5); close(s1, sgc_lr[r:r + 1].cpu(), 2e-5)
        close(e1, emb[r:r + 1].cpu(), 1e-5); close(l1, logits[r:r + 1].cpu(), 1e-5)


def test_vis_mode_matches_seg_and_backbone():
    """forward(cam='vis') (src/MuSCLe.py:290-298): no-grad decoder forward returning (seg_map, p7); seg_map equals the
    'seg' mode's and p7 equals the oracle's NCHW backbone tap (train mode: batch statistics)."""
    from oracle import mcl_oracle as O
    from test_gpu_decoder import build_dec
    name, n, size, seed = "efficientnet-b3", 2, 96, 41
    cfg, sd, model = build_dec(name, seed)
    x = T(synth.normal(seed, "x", (n, 3, size, size)).astype(np.float32))
    du = gu.drop_draws(cfg, n, 5)
    dud = {k: v.to(DEV) for k, v in du.items()}
    # train mode: batch statistics (an uncalibrated random-init net is degenerate in eval mode)
    model.train()
    with torch.no_grad():
        seg, _ = model(x.to(DEV), cam="seg", drop_u=dud)
    seg_v, p7 = model(x.to(DEV), cam="vis", drop_u=dud)
    assert not seg_v.requires_grad and not p7.requires_grad
    close(seg_v, seg.cpu(), 2e-5)
    net = O.OracleDecNet(name, sd)
    net.train()
    with torch.no_grad():
        oseg, _ = net.forward_seg(x, du)
        feats = net.features(x, du)
    close(seg_v, oseg, 5e-4)
    ref_p7 = feats[cfg.taps[6]]
    assert tuple(p7.shape) == tuple(ref_p7.shape)
    close(p7, ref_p7, 5e-4)
