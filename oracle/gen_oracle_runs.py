"""Write tests/golden/oracle_runs/*.npz: the CPU oracle's fp32 and fp64 forward/backward passes for the seeded cases of
tests/test_gpu_model.py::test_model_forward_backward, tests/test_gpu_backbone.py::test_backbone_forward_backward and
tests/test_gpu_decoder.py::test_seg_forward_backward_vs_oracle (build container; TEST INFRASTRUCTURE, never imported by
the product).

    python oracle/gen_oracle_runs.py [case-substring ...]

Why fixtures: these passes used to run inside the GPU tests, on the GPU box's host cores, where the B7 case alone took
110 s (and the driver's `pytest -m gpu` hit its 900 s limit in round 4).  What is stored is the output of
oracle/mcl_oracle.py - the restatement that tests/test_oracle_golden.py pins against the reference's own fixtures - on
inputs that both sides regenerate from muscle_amd.synth; tests/test_oracle_golden.py::test_oracle_run_fixtures_are_current
re-runs the small cases against the oracle at HEAD, so a fixture cannot go stale silently.  Layout: tests/golden_util.py
(pack_outputs / pack_grads)."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_util as gu  # noqa: E402
from muscle_amd import synth  # noqa: E402
from muscle_amd.arch import net_cfg  # noqa: E402
from oracle import mcl_oracle as O  # noqa: E402

T = lambda a: torch.from_numpy(np.asarray(a))  # noqa: E731

MODEL_CASES = [("efficientnet-b0", 3, 64, "cam", True), ("efficientnet-b0", 2, 96, "pix", False),
               ("efficientnet-b3", 2, 72, "cam", True), ("efficientnet-b7", 2, 64, "cam", True),
               ("efficientnet-b0", 3, 64, "logits", True)]
BACKBONE_CASES = [("efficientnet-b0", 3, 64, True), ("efficientnet-b0", 2, 72, False), ("efficientnet-b3", 2, 96, True)]
SEG_CASES = [("efficientnet-b3", 2, 96, 31)]
MODEL_SEED, BACKBONE_SEED = 23, 17


def case_name(kind, name, *rest):
    return "_".join([kind, name.replace("efficientnet-", "")] + [str(int(r)) if isinstance(r, bool) else str(r) for r in rest])


def _grads(net):
    return {k: p.grad for k, p in net.named_parameters()}


def model_run(name, n, size, mode, training):
    seed = MODEL_SEED
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    x = T(synth.normal(seed, "x", (n, 3, size, size)).astype(np.float32))
    du = gu.drop_draws(cfg, n, 5)
    res = []
    for dt in (torch.float32, torch.float64):
        net = O.OracleNet(name, sd, dtype=dt)
        net.train() if training else net.eval()
        outs = net.forward(x.to(dt), mode, du)
        probes = [T(synth.normal(seed, f"probe{i}", tuple(o.shape)).astype(np.float32)).to(dt) for i, o in enumerate(outs)]
        sum((o * p).sum() for o, p in zip(outs, probes)).backward()
        res.append((net, [o.detach() for o in outs]))
    (n32, o32), (n64, _) = res
    return {**gu.pack_outputs("out", o32), **gu.pack_grads(_grads(n32), _grads(n64))}


def backbone_run(name, n, size, training):
    seed = BACKBONE_SEED
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    x = T(synth.normal(seed, "x", (n, 3, size, size)).astype(np.float32))
    du = gu.drop_draws(cfg, n, 5)
    taps = cfg.taps
    res = []
    for dt in (torch.float32, torch.float64):
        net = O.OracleNet(name, sd, dtype=dt)
        net.train() if training else net.eval()
        feats = net.features(x.to(dt), du)
        probes = {i: T(synth.normal(seed, f"probe{i}", tuple(feats[i].shape)).astype(np.float32)).to(dt)
                  for i in (taps[0], taps[2], taps[4], taps[6])}
        sum((feats[i] * p).sum() for i, p in probes.items()).backward()
        res.append((net, [f.detach() for f in feats]))
    (n32, f32), (n64, _) = res
    pick = lambda net: {k: p.grad for k, p in net.named_parameters() if k.startswith("backbone.")}  # noqa: E731
    d = {**gu.pack_outputs("feat", f32), **gu.pack_grads(pick(n32), pick(n64))}
    rs = [k for k in n32.t if k.startswith("backbone.") and (k.endswith("running_mean") or k.endswith("running_var"))]
    d["rs_keys"] = np.array(rs)
    d["rs_off"] = np.cumsum([0] + [n32.t[k].numel() for k in rs]).astype(np.int64)
    d["rs_vals"] = np.concatenate([n32.t[k].detach().numpy().ravel() for k in rs]).astype(np.float32)
    return d


def seg_run(name, n, size, seed):
    cfg = net_cfg(name, True)
    sd = synth.synth_state_dict(cfg, seed, mode="dec", layers=3)
    x = T(synth.normal(seed, "x", (n, 3, size, size)).astype(np.float32))
    du = gu.drop_draws(cfg, n, 5)
    res = []
    for dt in (torch.float32, torch.float64):
        net = O.OracleDecNet(name, sd, dtype=dt)
        net.train()
        outs = net.forward_seg(x.to(dt), du)
        probes = [T(synth.normal(seed, f"probe{i}", tuple(o.shape)).astype(np.float32)).to(dt) for i, o in enumerate(outs)]
        sum((o * p).sum() for o, p in zip(outs, probes)).backward()
        res.append((net, [o.detach() for o in outs]))
    (n32, o32), (n64, _) = res
    d = {**gu.pack_outputs("out", o32), **gu.pack_grads(_grads(n32), _grads(n64))}
    rs = [k for k in n32.t if k.endswith("running_var")]
    d["rs_keys"] = np.array(rs)
    d["rs_off"] = np.cumsum([0] + [n32.t[k].numel() for k in rs]).astype(np.int64)
    d["rs_vals"] = np.concatenate([n32.t[k].detach().numpy().ravel() for k in rs]).astype(np.float32)
    return d


def all_cases():
    for c in MODEL_CASES:
        yield case_name("model", *c), (lambda c=c: model_run(*c))
    for c in BACKBONE_CASES:
        yield case_name("backbone", *c), (lambda c=c: backbone_run(*c))
    for c in SEG_CASES:
        yield case_name("seg", *c), (lambda c=c: seg_run(*c))


def main(argv):
    os.makedirs(gu.RUNS, exist_ok=True)
    for name, fn in all_cases():
        if argv and not any(a in name for a in argv):
            continue
        t = time.time()
        d = fn()
        path = os.path.join(gu.RUNS, name + ".npz")
        np.savez_compressed(path, **d)
        print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t:.1f} s", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
