"""EfficientNet backbone container with the reference's parameter names.

The reference backbone (src/efficientnet_pytorch/model.py:101-240) is an nn.Module tree whose
state_dict keys are the compatibility contract.  Here the same tree of
nn.Conv2d / nn.BatchNorm2d / nn.Linear objects exists only to *hold* parameters and buffers
under those names (and to give BatchNorm its momentum/eps attributes); none of their forward()
methods is ever called — arithmetic goes through muscle_amd.engine on the HIP kernels.
"""
from __future__ import annotations

import torch
from torch import nn

from .arch import BN_EPS, BN_MOMENTUM, NetCfg, net_cfg


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container only: the computation runs in muscle_amd.engine (HIP)")


class MBConvBlock(_NoForward):
    """Parameters of model.py:29-65."""

    def __init__(self, b):
        super().__init__()
        self.cfg = b
        if b.expand:
            self._expand_conv = nn.Conv2d(b.cin, b.cexp, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(b.cexp, momentum=BN_MOMENTUM, eps=BN_EPS)
        self._depthwise_conv = nn.Conv2d(b.cexp, b.cexp, b.kernel, stride=b.stride, groups=b.cexp, bias=False)
        self._bn1 = nn.BatchNorm2d(b.cexp, momentum=BN_MOMENTUM, eps=BN_EPS)
        self._se_reduce = nn.Conv2d(b.cexp, b.se, 1)
        self._se_expand = nn.Conv2d(b.se, b.cexp, 1)
        self._project_conv = nn.Conv2d(b.cexp, b.cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(b.cout, momentum=BN_MOMENTUM, eps=BN_EPS)


class EfficientNet(_NoForward):
    """Parameters of model.py:114-162 (the head conv / bn / fc are dead on this path but present
    in every checkpoint, so they are kept)."""

    def __init__(self, cfg: NetCfg, num_classes: int = 1000):
        super().__init__()
        self.cfg = cfg
        self._conv_stem = nn.Conv2d(3, cfg.stem_out, 3, stride=2, bias=False)
        self._bn0 = nn.BatchNorm2d(cfg.stem_out, momentum=BN_MOMENTUM, eps=BN_EPS)
        self._blocks = nn.ModuleList([MBConvBlock(b) for b in cfg.blocks])
        self._conv_head = nn.Conv2d(cfg.blocks[-1].cout, cfg.head_out, 1, bias=False)
        self._bn1 = nn.BatchNorm2d(cfg.head_out, momentum=BN_MOMENTUM, eps=BN_EPS)
        self._fc = nn.Linear(cfg.head_out, num_classes)

    @classmethod
    def from_name(cls, model_name: str, override_params=None, last_pooling: bool = True):
        """model.py:204-208.  Only `num_classes` is honoured in override_params."""
        nc = (override_params or {}).get("num_classes", 1000)
        return cls(net_cfg(model_name, last_pooling), nc)
