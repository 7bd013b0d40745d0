"""Framework-independent synthetic weights and batches for the MCL hot path.

Everything here is numpy only and keyed by (seed, name), so the container that
generates the golden fixtures (with the reference imported) and the GPU box
(which has no reference) regenerate bit-identical tensors.

Streams are raw Philox-4x64 words turned into doubles / Box-Muller normals by
hand, so nothing depends on numpy's distribution code.

Batch layout follows what the reference's loader yields to the loop body
(train_mcl.py:157; src/data.py:215-332): img [N,3,S,S] f32, label [N,20] f32
multi-hot, view1/view2 [N,3,V,V], coord1/coord2 [N,4] int64 = (h0,w0,hl,wl) of the
overlap window inside each view (src/data.py:233-270).
"""
from __future__ import annotations

import hashlib
from typing import Dict, Tuple

import numpy as np

from .arch import NetCfg

# VOC12 train_aug statistics; per-image label count histogram
# and class marginals (class 14 = person).
_COUNT_P = np.array([0.596, 0.289, 0.091, 0.019, 0.005])
_CLASS_W = np.array([590, 504, 705, 468, 714, 393, 1150, 1005, 1228, 267,
                     613, 1188, 445, 492, 4155, 522, 300, 649, 503, 567], dtype=np.float64)


def _bitgen(seed: int, name: str) -> np.random.Philox:
    d = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = np.frombuffer(d[:16], dtype=np.uint64).copy()
    return np.random.Philox(key=key)


def uniform(seed: int, name: str, shape) -> np.ndarray:
    """U[0,1) doubles with 53 random bits."""
    n = int(np.prod(shape)) if len(shape) else 1
    raw = _bitgen(seed, name).random_raw(n)
    return ((raw >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)


def normal(seed: int, name: str, shape) -> np.ndarray:
    """Standard normals (Box-Muller on the uniform stream), float64."""
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u = uniform(seed, name, (2 * m,))
    r = np.sqrt(-2.0 * np.log(1.0 - u[:m]))
    t = 2.0 * np.pi * u[m:]
    z = np.concatenate([r * np.cos(t), r * np.sin(t)])[:n]
    return z.reshape(shape)


# ---------------------------------------------------------------------------
# state dict
# ---------------------------------------------------------------------------
def state_dict_spec(cfg: NetCfg, num_classes: int = 21, mode: str = "enc",
                    bifpn_channels: int = 256, layers: int = 3) -> Dict[str, Tuple[int, ...]]:
    """Ordered {key: shape} of the reference MuSCLe.state_dict()."""
    spec: Dict[str, Tuple[int, ...]] = {}

    def bn(prefix, c):
        spec[prefix + ".weight"] = (c,)
        spec[prefix + ".bias"] = (c,)
        spec[prefix + ".running_mean"] = (c,)
        spec[prefix + ".running_var"] = (c,)
        spec[prefix + ".num_batches_tracked"] = ()

    spec["backbone._conv_stem.weight"] = (cfg.stem_out, 3, 3, 3)
    bn("backbone._bn0", cfg.stem_out)
    for b in cfg.blocks:
        p = f"backbone._blocks.{b.index}."
        if b.expand:
            spec[p + "_expand_conv.weight"] = (b.cexp, b.cin, 1, 1)
            bn(p + "_bn0", b.cexp)
        spec[p + "_depthwise_conv.weight"] = (b.cexp, 1, b.kernel, b.kernel)
        bn(p + "_bn1", b.cexp)
        spec[p + "_se_reduce.weight"] = (b.se, b.cexp, 1, 1)
        spec[p + "_se_reduce.bias"] = (b.se,)
        spec[p + "_se_expand.weight"] = (b.cexp, b.se, 1, 1)
        spec[p + "_se_expand.bias"] = (b.cexp,)
        spec[p + "_project_conv.weight"] = (b.cout, b.cexp, 1, 1)
        bn(p + "_bn2", b.cout)
    spec["backbone._conv_head.weight"] = (cfg.head_out, cfg.blocks[-1].cout, 1, 1)
    bn("backbone._bn1", cfg.head_out)
    spec["backbone._fc.weight"] = (num_classes, cfg.head_out)
    spec["backbone._fc.bias"] = (num_classes,)
    tc = cfg.tap_channels
    if mode == "enc":
        spec["fuse.weight"] = (128, tc[0] + tc[2] + tc[4], 1, 1)
        spec["fuse.bias"] = (128,)
        spec["fc.weight"] = (num_classes, tc[6])
    else:
        c = bifpn_channels
        for i, ci in zip(range(3, 8), tc[2:]):
            spec[f"BIFPN.inp{i}.0.weight"] = (c, ci, 1, 1)
            spec[f"BIFPN.inp{i}.0.bias"] = (c,)
            bn(f"BIFPN.inp{i}.1", c)
        for l in range(layers):
            q = f"BIFPN.BIFPN_Layers.{l}."
            for nm in ("convp67", "convp56", "convp45", "convp34"):
                spec[q + nm + ".0.weight"] = (c, 2 * c, 1, 1)
                spec[q + nm + ".0.bias"] = (c,)
            for nm in ("out4", "out5", "out6", "out7"):
                spec[q + nm + ".0.weight"] = (c, c, 1, 1)
                spec[q + nm + ".0.bias"] = (c,)
                bn(q + nm + ".1", c)
    spec["fuse_dec.weight"] = (num_classes, bifpn_channels, 1, 1)
    spec["fuse_dec.bias"] = (num_classes,)
    return spec


def synth_state_dict(cfg: NetCfg, seed: int = 0, **kw) -> Dict[str, np.ndarray]:
    """Synthetic weights: He-scaled convs, BN gamma in [0.75,1.25], small betas/biases,
    unit running stats (calibrate before eval-mode use, SURVEY.md §7 'Hard parts')."""
    out: Dict[str, np.ndarray] = {}
    for key, shape in state_dict_spec(cfg, **kw).items():
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = np.zeros((), dtype=np.int64)
        elif leaf == "running_mean":
            out[key] = np.zeros(shape, dtype=np.float32)
        elif leaf == "running_var":
            out[key] = np.ones(shape, dtype=np.float32)
        elif len(shape) == 1 and leaf == "weight":          # BN gamma
            out[key] = (0.75 + 0.5 * uniform(seed, key, shape)).astype(np.float32)
        elif len(shape) == 1:                                # BN beta / conv bias
            out[key] = (0.1 * normal(seed, key, shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 1.0 if key == "fc.weight" else np.sqrt(2.0)
            out[key] = (gain / np.sqrt(fan_in) * normal(seed, key, shape)).astype(np.float32)
    return out


# ---------------------------------------------------------------------------
# batches
# ---------------------------------------------------------------------------
def synth_labels(n: int, seed: int = 0, num_fg: int = 20, force_pairs: int = 2) -> np.ndarray:
    """Multi-hot [n, 20] float32 with VOC-like count/class statistics.  The first
    2*force_pairs rows are made pairwise identical so IMC has positive pairs
    (loss_multilabel.py:51) and sum(labels) stays <= 105 (train_mcl.py:188)."""
    u = uniform(seed, "labels", (n, 1 + num_fg))
    cdf = np.cumsum(_COUNT_P / _COUNT_P.sum())
    w = _CLASS_W[:num_fg] / _CLASS_W[:num_fg].sum()
    lab = np.zeros((n, num_fg), dtype=np.float32)
    for i in range(n):
        k = int(np.searchsorted(cdf, u[i, 0], side="right")) + 1
        # Gumbel-top-k draw of k distinct classes with marginals ~ w
        g = np.log(w) - np.log(-np.log(np.clip(u[i, 1:], 1e-12, 1 - 1e-12)))
        lab[i, np.argsort(-g)[:k]] = 1.0
    for p in range(min(force_pairs, n // 2)):
        lab[2 * p + 1] = lab[2 * p]
    return lab


def synth_coords(n: int, view: int, frame: int, seed: int = 0):
    """Two uniformly placed view x view windows in a frame x frame image, redrawn until
    they overlap; returns (coord1, coord2, origins) with coord* int64 [n,4] = (h0, w0, hl, wl) of the
    overlap inside each view (same algebra as src/data.py:233-270)."""
    c1 = np.zeros((n, 4), dtype=np.int64)
    c2 = np.zeros((n, 4), dtype=np.int64)
    org = np.zeros((n, 4), dtype=np.int64)
    for i in range(n):
        t = 0
        while True:
            u = uniform(seed, f"coords:{i}:{t}", (4,))
            i1, j1, i2, j2 = (int(x * (frame - view + 1)) for x in u)
            top, left = max(i1, i2), max(j1, j2)
            bot, right = min(i1, i2) + view, min(j1, j2) + view
            if bot - top > 0 and right - left > 0:
                break
            t += 1
        c1[i] = (top - i1, left - j1, bot - top, right - left)
        c2[i] = (top - i2, left - j2, bot - top, right - left)
        org[i] = (i1, j1, i2, j2)
    return c1, c2, org


def synth_batch(n: int, size: int, view: int, seed: int = 0) -> Dict[str, np.ndarray]:
    c1, c2, org = synth_coords(n, view, 2 * view, seed)
    frame = normal(seed, "frame", (n, 3, 2 * view, 2 * view)).astype(np.float32)
    v1 = np.stack([frame[i, :, o[0]:o[0] + view, o[1]:o[1] + view] for i, o in enumerate(org)])
    v2 = np.stack([frame[i, :, o[2]:o[2] + view, o[3]:o[3] + view] for i, o in enumerate(org)])
    return {
        "img": normal(seed, "img", (n, 3, size, size)).astype(np.float32),
        "view1": np.ascontiguousarray(v1),
        "view2": np.ascontiguousarray(v2),
        "label": synth_labels(n, seed),
        "coord1": c1,
        "coord2": c2,
    }


def synth_drop_masks(cfg: NetCfg, n: int, seed: int = 0, tag: str = "drop") -> Dict[int, np.ndarray]:
    """Per skip-block uniform draws U[0,1) of shape [n] (utils.py:88); the binary mask is
    floor(keep + u).  Replayed on both sides so parity does not depend on torch's RNG."""
    return {b.index: uniform(seed, f"{tag}:{b.index}", (n,)).astype(np.float32)
            for b in cfg.blocks if b.skip and b.drop_rate > 0}


def synth_soft_mask(label: np.ndarray, size: int, seed: int = 0) -> np.ndarray:
    """Config-4 pseudo-label [N, 21, S, S] float32 (what VOC12SegDataset yields, src/data.py:69-123): background
    0.35 everywhere plus one soft disc per positive class."""
    n, nf = label.shape
    m = np.zeros((n, nf + 1, size, size), dtype=np.float32)
    m[:, 0] = 0.35
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    for i in range(n):
        u = uniform(seed, f"mask:{i}", (nf, 3))
        for c in np.nonzero(label[i])[0]:
            cy, cx, r = (0.2 + 0.6 * u[c, 0]) * size, (0.2 + 0.6 * u[c, 1]) * size, (0.15 + 0.2 * u[c, 2]) * size
            d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
            m[i, c + 1] = np.clip(1.0 - d / r, 0.0, 1.0).astype(np.float32)
    return m
