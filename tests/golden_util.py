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
