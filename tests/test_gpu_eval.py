"""GPU parity of the per-epoch rapid evaluation (train_mcl.py:286-318 + src/evaluation.py).
Integer work: the (TP, P, T) tables must be bit-exact given the same prediction maps; end to end (own forward) the mIoU
may move only by pixels whose fp16-rounded value sits on a threshold."""
import numpy as np
import pytest
import torch

from muscle_amd import synth
from muscle_amd.arch import net_cfg

pytestmark = [pytest.mark.gpu, pytest.mark.both_arith]
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.asarray(a))  # noqa: E731


def _gt(seed, H, W):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 21, size=(H // 8 + 1, W // 8 + 1)).astype(np.uint8)
    g = np.kron(g, np.ones((8, 8), np.uint8))[:H, :W].copy()
    g[rng.random((H, W)) < 0.05] = 255                     # ignore label
    return g


def test_confusion_counts_bit_exact():
    from oracle import mcl_oracle as O
    from muscle_amd.evaluation import RapidEval, RAPID_THRESHOLDS
    rng = np.random.default_rng(5)
    ev = RapidEval(DEV)
    ref = np.zeros((len(RAPID_THRESHOLDS), 21, 3), np.int64)
    for im in range(3):
        H, W = 37 + 20 * im, 53 + 11 * im
        pred = rng.random((21, H, W)).astype(np.float32)
        pred[:, rng.random((H, W)) < 0.1] = 0.3            # exact ties between channels and with threshold 0.30
        lab = np.ones(21, np.float32)
        lab[1 + rng.choice(20, 14, replace=False)] = 0
        gt = _gt(im, H, W)
        ev.add_prediction(T(pred).to(DEV), T(lab), T(gt))
        pd = {i: (pred[i + 1] * lab[i + 1]).astype(np.half) for i in range(20)}
        for ti, t in enumerate(RAPID_THRESHOLDS):
            tp, p, tt = O.eval_compare(pd, gt, t)
            ref[ti, :, 0] += tp; ref[ti, :, 1] += p; ref[ti, :, 2] += tt
    got = ev.counts.cpu().numpy()
    assert np.array_equal(got, ref)
    for ti in range(len(RAPID_THRESHOLDS)):
        m, per = O.eval_miou(ref[ti, :, 0], ref[ti, :, 1], ref[ti, :, 2])
        ll = ev.loglist(ti)
        assert ll['mIoU'] == m and ll['aeroplane'] == per[1]
    mm, mt, mious = ev.best()
    assert mm == max(mious) and mt == RAPID_THRESHOLDS[mious.index(mm)]


def test_rapid_eval_end_to_end():
    import muscle_amd
    from oracle import mcl_oracle as O
    from muscle_amd.evaluation import RapidEval, RAPID_THRESHOLDS
    name, seed = "efficientnet-b0", 11
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    model = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False)
    model.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    model = model.to(DEV).eval()
    net = O.OracleNet(name, sd)
    ev = RapidEval(DEV)
    ref = np.zeros((len(RAPID_THRESHOLDS), 21, 3), np.int64)
    for im in range(2):
        H, W = 72 + 24 * im, 96
        img = T(synth.normal(seed, f"img{im}", (1, 3, H, W)).astype(np.float32))
        label = torch.zeros(1, 20)
        label[0, [3 + im, 9, 17]] = 1
        gt = _gt(100 + im, H, W)
        ev.add(model, img.to(DEV), label, T(gt))
        pd = O.eval_pred_dict(net, img, label)
        for ti, t in enumerate(RAPID_THRESHOLDS):
            tp, p, tt = O.eval_compare(pd, gt, t)
            ref[ti, :, 0] += tp; ref[ti, :, 1] += p; ref[ti, :, 2] += tt
    got = ev.counts.cpu().numpy()
    assert np.array_equal(got[:, :, 2], ref[:, :, 2])                    # T depends on gt only
    npx = ref[0, :, 2].sum()
    assert np.abs(got - ref).sum() <= 6e-3 * npx * len(RAPID_THRESHOLDS)  # a few threshold-straddling pixels at most
    for ti in range(len(RAPID_THRESHOLDS)):
        m, _ = O.eval_miou(ref[ti, :, 0], ref[ti, :, 1], ref[ti, :, 2])
        assert abs(ev.loglist(ti)['mIoU'] - m) <= 0.05                    # percent


def test_rapid_eval_matches_reference_fixture():
    """RapidEval against what src/evaluation.py::do_python_eval itself returned (tests/golden/eval_rapid.npz)."""
    import os
    from muscle_amd.evaluation import RapidEval, categories
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_rapid.npz"))
    ev = RapidEval(DEV, thresholds=[float(t) for t in z["thresholds"]])
    for i in range(5):
        pred = np.concatenate([np.zeros((1,) + z[f"pred{i}"].shape[1:], np.float32), z[f"pred{i}"].astype(np.float32)])
        ev.add_prediction(T(pred).to(DEV), torch.ones(21), T(z[f"gt{i}"]))
    for ti in range(len(z["thresholds"])):
        ll = ev.loglist(ti)
        got = np.array([ll[c] for c in categories] + [ll["mIoU"]])
        assert np.allclose(got, z["loglists"][ti], rtol=0, atol=1e-9), ti
