"""EfficientNet-B0..B7 block geometry for the MCL/MuSCLe hot path.

Pure-Python, no torch.  Produces, for a backbone name, the flat list of MBConv
block configurations (channels, kernel, stride, SE width, static "same" pads)
that the HIP path, the oracle and the roofline accounting all share.

Reference behaviour restated here (file:line are relative to /root/reference):
  * compound scaling table ............ src/efficientnet_pytorch/utils.py:161-174
  * stage strings, last_pooling switch  src/efficientnet_pytorch/utils.py:252-274
  * width rounding (round_filters) .... src/efficientnet_pytorch/utils.py:59-71
  * depth rounding (round_repeats) .... src/efficientnet_pytorch/utils.py:74-79
  * repeat blocks take stage *output* width as input and stride 1
                                        src/efficientnet_pytorch/model.py:136-150
  * SE squeeze width = max(1, int(block_in * 0.25)) ... model.py:57
  * static same-padding computed for the *nominal* image size, not the actual
    input ............................. src/efficientnet_pytorch/utils.py:125-140
  * drop_connect rate = 0.2 * idx / n_blocks, block 0 draws nothing
                                        model.py:180-183, utils.py:82-91
  * feature taps p1..p7 ............... src/MuSCLe.py:167-178 (B0 added by the
    same last-block-of-stage rule; the reference has no B0 table)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

# name -> (width_mult, depth_mult, nominal_resolution, dropout)
_SCALING = {
    "efficientnet-b0": (1.0, 1.0, 224, 0.2),
    "efficientnet-b1": (1.0, 1.1, 240, 0.2),
    "efficientnet-b2": (1.1, 1.2, 260, 0.3),
    "efficientnet-b3": (1.2, 1.4, 300, 0.3),
    "efficientnet-b4": (1.4, 1.8, 380, 0.4),
    "efficientnet-b5": (1.6, 2.2, 456, 0.4),
    "efficientnet-b6": (1.8, 2.6, 528, 0.5),
    "efficientnet-b7": (2.0, 3.1, 600, 0.5),
}

# (kernel, repeats, in, out, expand, stride) per stage, base (B0) widths.
# Stage 6 (index 5) has stride 2 only when last_pooling=True.
_STAGES = [
    (3, 1, 32, 16, 1, 1),
    (3, 2, 16, 24, 6, 2),
    (5, 2, 24, 40, 6, 2),
    (3, 3, 40, 80, 6, 2),
    (5, 3, 80, 112, 6, 1),
    (5, 4, 112, 192, 6, 2),
    (3, 1, 192, 320, 6, 1),
]

BN_EPS = 1e-3          # utils.py:281
BN_MOMENTUM = 0.01     # 1 - 0.99, model.py:32 / utils.py:280
DROP_CONNECT_RATE = 0.2
SE_RATIO = 0.25
DEPTH_DIVISOR = 8


def _round_filters(filters: int, width_mult: float) -> int:
    f = filters * width_mult
    new_f = max(DEPTH_DIVISOR, int(f + DEPTH_DIVISOR / 2) // DEPTH_DIVISOR * DEPTH_DIVISOR)
    if new_f < 0.9 * f:
        new_f += DEPTH_DIVISOR
    return int(new_f)


def _round_repeats(repeats: int, depth_mult: float) -> int:
    return int(math.ceil(depth_mult * repeats))


def static_same_pad(kernel: int, stride: int, nominal: int) -> Tuple[int, int]:
    """(before, after) zero padding along one axis, frozen at model build time."""
    out = math.ceil(nominal / stride)
    pad = max((out - 1) * stride + (kernel - 1) + 1 - nominal, 0)
    return pad // 2, pad - pad // 2


@dataclass(frozen=True)
class BlockCfg:
    index: int
    cin: int
    cexp: int          # == cin when expand_ratio == 1 (no expand conv / bn0)
    cout: int
    kernel: int
    stride: int
    se: int            # squeeze channels
    expand: bool
    skip: bool         # stride 1 and cin == cout
    pad_lo: int        # static pad before (top/left)
    pad_hi: int        # static pad after (bottom/right)
    drop_rate: float   # train-mode drop_connect rate (0.0 => no draw)

    def out_size(self, size: int) -> int:
        return (size + self.pad_lo + self.pad_hi - self.kernel) // self.stride + 1


@dataclass(frozen=True)
class NetCfg:
    name: str
    nominal: int
    stem_out: int
    stem_pad: Tuple[int, int]
    blocks: Tuple[BlockCfg, ...]
    head_out: int              # dead _conv_head width (present in state_dict)
    taps: Tuple[int, ...]      # block indices of p1..p7
    tap_channels: Tuple[int, ...]
    last_pooling: bool

    def stem_out_size(self, size: int) -> int:
        lo, hi = self.stem_pad
        return (size + lo + hi - 3) // 2 + 1


def net_cfg(name: str, last_pooling: bool = True) -> NetCfg:
    if name not in _SCALING:
        raise ValueError("model_name should be one of: " + ", ".join(sorted(_SCALING)))
    wm, dm, nominal, _ = _SCALING[name]
    # the static pads are derived from the nominal size divided down per stage:
    # every Conv2dStaticSamePadding is built with image_size == nominal, so the
    # pad depends only on (k, s, nominal), not on the running feature size.
    blocks: List[BlockCfg] = []
    stage_last: List[int] = []
    stage_cout: List[int] = []
    for si, (k, r, ci, co, e, s) in enumerate(_STAGES):
        if si == 5 and not last_pooling:
            s = 1
        ci_r, co_r, r_r = _round_filters(ci, wm), _round_filters(co, wm), _round_repeats(r, dm)
        for rep in range(r_r):
            cin = ci_r if rep == 0 else co_r
            stride = s if rep == 0 else 1
            lo, hi = static_same_pad(k, stride, nominal)
            blocks.append(BlockCfg(
                index=len(blocks), cin=cin, cexp=cin * e, cout=co_r, kernel=k, stride=stride,
                se=max(1, int(cin * SE_RATIO)), expand=(e != 1),
                skip=(stride == 1 and cin == co_r), pad_lo=lo, pad_hi=hi, drop_rate=0.0))
        stage_last.append(len(blocks) - 1)
        stage_cout.append(co_r)
    n = len(blocks)
    blocks = [BlockCfg(**{**b.__dict__, "drop_rate": DROP_CONNECT_RATE * b.index / n}) for b in blocks]
    return NetCfg(
        name=name, nominal=nominal, stem_out=_round_filters(32, wm),
        stem_pad=static_same_pad(3, 2, nominal), blocks=tuple(blocks),
        head_out=_round_filters(1280, wm), taps=tuple(stage_last),
        tap_channels=tuple(stage_cout), last_pooling=last_pooling)


# ---------------------------------------------------------------------------
# algorithmic work; used by bench.py's roofline object
# ---------------------------------------------------------------------------
def forward_macs(cfg: NetCfg, size: int) -> dict:
    """Per-image forward MACs split by kernel family, for an HxH input."""
    h = cfg.stem_out_size(size)
    macs = {"stem": 27 * cfg.stem_out * h * h, "pointwise": 0, "depthwise": 0, "se": 0}
    for b in cfg.blocks:
        ho = b.out_size(h)
        if b.expand:
            macs["pointwise"] += b.cin * b.cexp * h * h
        macs["depthwise"] += b.kernel * b.kernel * b.cexp * ho * ho
        macs["se"] += 2 * b.cexp * b.se
        macs["pointwise"] += b.cexp * b.cout * ho * ho
        h = ho
    macs["total"] = sum(macs.values())
    return macs


def min_materialisation_bytes(cfg: NetCfg, size: int) -> int:
    """Per-image forward HBM bytes (fp32) of the minimum-materialisation training
    schedule of SURVEY.md §8(d): per block in + 2*exp + 3*dw + out (+ skip re-read)."""
    h = cfg.stem_out_size(size)
    elems = 3 * size * size + cfg.stem_out * h * h
    for b in cfg.blocks:
        ho = b.out_size(h)
        elems += b.cin * h * h
        if b.expand:
            elems += 2 * b.cexp * h * h
        elems += 3 * b.cexp * ho * ho + b.cout * ho * ho
        if b.skip:
            elems += b.cin * h * h
        h = ho
    return 4 * elems
