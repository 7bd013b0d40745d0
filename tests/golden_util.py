"""Helpers shared by the golden/parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from muscle_amd import synth
from muscle_amd.arch import net_cfg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def cam_stride(G):
    """Spatial stride of the stored CAM / SGC samples (4 in the small fixtures, 16 at 448x448)."""
    return int(G["cam_stride"]) if "cam_stride" in G.files else 4


def tensor_summary(named, seed=123):
    """[l2, probe-dot] per tensor, same convention as oracle/gen_golden.py."""
    rows = []
    for k, v in named:
        if v is None:
            rows.append([np.nan, np.nan])
            continue
        a = v.detach().double().cpu().numpy().ravel()
        pr = synth.normal(seed, k, a.shape)
        rows.append([float(np.sqrt((a * a).sum())), float((a * pr).sum())])
    return np.array(rows, dtype=np.float64)


def drop_draws(cfg, n, torch_seed):
    """Replay the CPU generator in the reference's draw order (utils.py:88)."""
    torch.manual_seed(torch_seed)
    return {b.index: torch.rand([n, 1, 1, 1]).view(-1).clone() for b in cfg.blocks if b.skip and b.drop_rate}


def geometry_from_draws(coord1, draws):
    """Turn the logged np.random.randint results back into per-sample (lh, lw, sh, sw)."""
    it = iter(int(d) for d in draws)
    geo = []
    for c in coord1:
        h, w = int(c[2]), int(c[3])
        if h < 15 or w < 15 or h / w > 5 or w / h > 5:
            geo.append(None)
            continue
        lh, lw = next(it), next(it)
        while lh < 5 or lw < 5:
            lh, lw = next(it), next(it)
        geo.append((lh, lw, next(it), next(it)))
    assert next(it, None) is None
    return geo


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


_CACHE = {}


def cached(key, fn):
    """One CPU oracle run per (test, case) and process: the tests marked `both_arith` run twice (tests/conftest.py), and
    the fp32 + fp64 oracle passes (tens of seconds on the host cores) do not depend on the GPU's GEMM arithmetic."""
    if key not in _CACHE:
        _CACHE[key] = fn()
    return _CACHE[key]


# ---------------------------------------------------------------------------------------------------------------------
# Oracle-run fixtures (tests/golden/oracle_runs/*.npz, written by oracle/gen_oracle_runs.py in the build container).
# The CPU oracle's fp32 AND fp64 passes for the seeded forward/backward cases used to run inside the GPU tests (110 s
# for the B7 case on the GPU box's host share); their results are stored instead.  Per tensor the fixture holds the
# values at `sample_idx` (every element when the tensor is small), the statistics the tolerance is built from
# (max|g64|, the oracle's own fp32-vs-fp64 distance) and two seeded full-tensor projections, so that an error anywhere
# in a tensor still shows although only a sample of its elements is stored.
# ---------------------------------------------------------------------------------------------------------------------
RUNS = os.path.join(GOLDEN, "oracle_runs")
CAP_GRAD, CAP_OUT = 1024, 8192
PROBE_SEEDS = (123, 124)


def sample_idx(numel: int, cap: int) -> np.ndarray:
    """Deterministic element sample: all of a small tensor, else `cap` positions spread by a multiplicative hash (a large
    prime step modulo numel: no alignment with channel / row strides, no dependence on a library's RNG)."""
    if numel <= cap:
        return np.arange(numel, dtype=np.int64)
    return (np.arange(cap, dtype=np.int64) * 2654435761 + 40503) % numel


def _proj(a64: np.ndarray, key: str) -> list:
    return [float((a64 * synth.normal(s, key, a64.shape)).sum()) for s in PROBE_SEEDS]


def pack_outputs(prefix, outs32) -> dict:
    """Oracle fp32 outputs -> fixture entries."""
    d = {}
    for i, o in enumerate(outs32):
        a = o.detach().double().numpy().ravel()
        d[f"{prefix}{i}_shape"] = np.array(o.shape, dtype=np.int64)
        d[f"{prefix}{i}_v"] = a[sample_idx(a.size, CAP_OUT)].astype(np.float32)
        d[f"{prefix}{i}_stat"] = np.array([np.abs(a).max(), np.sqrt((a * a).sum())] + _proj(a, f"{prefix}{i}"))
    d[f"{prefix}_n"] = np.array(len(outs32))
    return d


def pack_grads(named32, named64) -> dict:
    """Oracle gradients (fp32 and fp64 passes, dicts name -> grad or None) -> fixture entries."""
    keys, stat, vals, off = [], [], [], [0]
    for k, g64 in named64.items():
        keys.append(k)
        if g64 is None:
            assert named32[k] is None, k
            stat.append([np.nan] * (3 + len(PROBE_SEEDS)))
            off.append(off[-1])
            continue
        b64 = g64.detach().double().numpy().ravel()
        b32 = named32[k].detach().double().numpy().ravel()
        v = b64[sample_idx(b64.size, CAP_GRAD)].astype(np.float32)
        stat.append([np.abs(b64).max(), np.abs(b32 - b64).max(), np.sqrt((b64 * b64).sum())] + _proj(b64, k))
        vals.append(v)
        off.append(off[-1] + v.size)
    return {"g_keys": np.array(keys), "g_stat": np.array(stat, dtype=np.float64), "g_off": np.array(off, dtype=np.int64),
            "g_vals": np.concatenate(vals) if vals else np.zeros(0, np.float32)}


def load_run(case: str):
    return np.load(os.path.join(RUNS, case + ".npz"), allow_pickle=False)


def _probes_on(dev, key, shape, cache):
    ck = ("probe", key, tuple(shape))
    if cache is None or ck not in cache:
        p = [torch.from_numpy(synth.normal(s, key, tuple(shape)).ravel()).to(dev) for s in PROBE_SEEDS]
        if cache is None:
            return p
        cache[ck] = p
    return cache[ck]


def check_outputs(got, F, prefix, tol, cache=None):
    """`close(g, o, tol)` of the former in-test oracle run (max-abs error relative to max|ref|) on the stored sample, plus
    the whole tensor through its projections (an l2 estimate of the error, held to 2 tol of the reference's norm)."""
    assert len(got) == int(F[f"{prefix}_n"])
    for i, g in enumerate(got):
        assert tuple(g.shape) == tuple(int(v) for v in F[f"{prefix}{i}_shape"]), (i, tuple(g.shape))
        a = g.detach().double().flatten()
        st = F[f"{prefix}{i}_stat"]
        idx = torch.from_numpy(sample_idx(a.numel(), CAP_OUT)).to(a.device)
        ref = torch.from_numpy(F[f"{prefix}{i}_v"]).double().to(a.device)
        err = float((a[idx] - ref).abs().max()) / max(float(st[0]), 1e-30)
        assert err <= tol, (prefix, i, err)
        pr = _probes_on(a.device, f"{prefix}{i}", (a.numel(),), cache)
        d2 = np.mean([(float(a @ p) - float(st[2 + j])) ** 2 for j, p in enumerate(pr)])
        assert np.sqrt(d2) <= 2 * tol * max(float(st[1]), 1e-30), (prefix, i, np.sqrt(d2) / float(st[1]))


def check_grads_fixture(named, F, tol=2e-3, cache=None):
    """The gradient check of the former in-test oracle run, against the stored fp64 oracle gradient: per tensor
    max|a - g64| <= tol * max|g64| + 20 * (the oracle's own fp32-vs-fp64 distance) on the sample (= every element of a
    tensor of <= CAP_GRAD elements), and for well-conditioned tensors cosine >= 0.9999 both on the sample and for the
    whole tensor (|a - g64|^2 estimated from the projections).  `named`: name -> gradient tensor or None."""
    keys = [str(k) for k in F["g_keys"]]
    stat, off, vals = F["g_stat"], F["g_off"], F["g_vals"]
    assert set(keys) == set(named.keys())
    worst = 0.0
    for j, k in enumerate(keys):
        g = named[k]
        if np.isnan(stat[j, 0]):
            assert g is None, k
            continue
        assert g is not None, k
        a = g.detach().double().flatten()
        scale, noise, l2 = max(float(stat[j, 0]), 1e-30), float(stat[j, 1]), float(stat[j, 2])
        idx = torch.from_numpy(sample_idx(a.numel(), CAP_GRAD)).to(a.device)
        ref = torch.from_numpy(vals[off[j]:off[j + 1]]).double().to(a.device)
        s = a[idx]
        err = float((s - ref).abs().max())
        assert err <= tol * scale + 20 * noise, (k, err / scale, noise / scale)
        if noise <= 1e-4 * scale:
            cos = float(s @ ref / (s.norm() * ref.norm() + 1e-30))
            assert cos >= 0.9999, (k, cos)
            pr = _probes_on(a.device, k, (a.numel(),), cache)
            d2 = np.mean([(float(a @ p) - float(stat[j, 3 + i])) ** 2 for i, p in enumerate(pr)])
            na = float(a.norm())
            cos_full = (na * na + l2 * l2 - d2) / (2 * na * l2 + 1e-30)
            assert cos_full >= 0.9999, (k, cos_full)
            worst = max(worst, err / scale)
    return worst
