"""Decoder mode (config 4): BiFPN + fuse_dec on the HIP kernels — MuSCLe.forward(cam='seg'), src/MuSCLe.py:30-58,115-148,281-287.

The BiFPN is a small DAG of 1x1 convolutions (+bias), BatchNorm (torch defaults eps 1e-5 / momentum 0.1), SiLU, bilinear
align_corners resizes, 3x3/s2 average pools, concatenations and sums on 256-channel maps of at most 1/8 resolution.  It is
run through a minimal reverse-mode tape over the NHWC kernels: every op records a closure that turns the gradient of its
output into gradients of its inputs / parameters, so fan-out and the dead branches of the last layer need no hand derivation.
A channel concatenation followed by a 1x1 conv is computed as two GEMMs accumulating into one output (never materialised).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn

from . import ops
from ._lib import call, ptr, stream
from .ops import BNState

_CPAD = 24


class Tape:
    def __init__(self, sink, training: bool):
        self.sink, self.training = sink, training
        self.steps: List = []
        self.g: Dict[int, torch.Tensor] = {}
        self.keep: List[torch.Tensor] = []       # keeps ids alive
        self.touched = set()

    def add_grad(self, t, g):
        k = id(t)
        self.g[k] = g if k not in self.g else ops.ew(1, self.g[k], g, alpha=1.0)

    def grad(self, t):
        return self.g.get(id(t))

    def pgrad(self, p):
        self.touched.add(id(p))
        return self.sink.of(p)

    def run_backward(self):
        for fn in reversed(self.steps):
            fn()


def _unit(C, dev):
    return BNState(torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev))


def conv1x1(tp: Tape, xs: List[torch.Tensor], conv: nn.Conv2d, want_stats=False):
    """y = conv(cat(xs, channel)) + bias, NHWC; xs share N,H,W.  Returns (y, stats or None)."""
    N, H, W, _ = xs[0].shape
    M, Co = N * H * W, conv.out_channels
    Wt = conv.weight.view(Co, -1)
    parts, off = [], 0
    for x in xs:
        c = x.shape[3]
        parts.append(Wt[:, off:off + c].contiguous() if len(xs) > 1 else Wt)
        off += c
    y, stats = None, None
    for i, (x, w) in enumerate(zip(xs, parts)):
        last = i == len(xs) - 1
        r = ops.pw_fwd(x.view(M, x.shape[3]), w, Co, bias=conv.bias if last else None, residual=y,
                       want_stats=want_stats and last)
        if want_stats and last:
            y, stats = r
        else:
            y = r
    y = y.view(N, H, W, Co)
    tp.keep.append(y)

    def bwd():
        g = tp.grad(y)
        if g is None:
            return
        g2 = g.reshape(M, Co)
        tp.pgrad(conv.bias).add_(ops.pool_sum(g2, M).view(Co))
        dW = tp.pgrad(conv.weight).view(Co, -1)
        off = 0
        for x, w in zip(xs, parts):
            c = x.shape[3]
            if len(xs) > 1:
                dwp = torch.zeros(Co, c, dtype=torch.float32, device=g.device)
                ops.pw_wgrad(g2, x.view(M, c), dwp)
                dW[:, off:off + c].add_(dwp)
            else:
                ops.pw_wgrad(g2, x.view(M, c), dW)
            tp.add_grad(x, ops.pw_dgrad(g2, w, c).view(N, H, W, c))
            off += c
    tp.steps.append(bwd)
    return y, stats


def bn_swish(tp: Tape, z, stats, bn: nn.BatchNorm2d):
    N, H, W, C = z.shape
    M = N * H * W
    st = ops.bn_finalize(stats, M, bn, tp.training)
    y = ops.bn_apply(z.view(M, C), st, act=True).view(N, H, W, C)
    tp.keep.append(y)

    def bwd():
        g = tp.grad(y)
        if g is None:
            return
        dz = ops.bn_backward(g.reshape(M, C), z.view(M, C), bn, st, tp.pgrad(bn.weight), tp.pgrad(bn.bias), tp.training, act=st)
        tp.add_grad(z, dz.view(N, H, W, C))
    tp.steps.append(bwd)
    return y


def swish(tp: Tape, z):
    N, H, W, C = z.shape
    M = N * H * W
    u = _unit(C, z.device)
    y = ops.bn_apply(z.view(M, C), u, act=True).view(N, H, W, C)
    tp.keep.append(y)

    def bwd():
        g = tp.grad(y)
        if g is None:
            return
        g2, z2 = g.reshape(M, C).contiguous(), z.view(M, C)
        out = torch.empty_like(z2)
        zero = torch.zeros(C, device=z.device)
        call("mx_bn_bwd_apply", ptr(g2), ptr(z2), None, None, None, ptr(u.scale), ptr(u.shift), ptr(u.scale), ptr(zero), ptr(zero),
             ptr(out), M, C, 1, stream())
        tp.add_grad(z, out.view(N, H, W, C))
    tp.steps.append(bwd)
    return y


def resize(tp: Tape, x, Hd, Wd):
    N, Hs, Ws, C = x.shape
    if (Hs, Ws) == (Hd, Wd):
        return x
    y = torch.empty(N, Hd, Wd, C, dtype=torch.float32, device=x.device)
    ops.resize_nhwc(x, y, 0, relu=False)
    tp.keep.append(y)

    def bwd():
        g = tp.grad(y)
        if g is None:
            return
        gx = torch.zeros_like(x)
        call("mx_resize_nhwc_bwd", ptr(g.contiguous()), ptr(gx), N, Hs, Ws, C, Hd, Wd, stream())
        tp.add_grad(x, gx)
    tp.steps.append(bwd)
    return y


def avgpool(tp: Tape, x):
    N, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(N, Ho, Wo, C, dtype=torch.float32, device=x.device)
    call("mx_avgpool3s2", ptr(x), ptr(y), N, H, W, C, 0, stream())
    tp.keep.append(y)

    def bwd():
        g = tp.grad(y)
        if g is None:
            return
        gx = torch.empty_like(x)
        call("mx_avgpool3s2", ptr(g.contiguous()), ptr(gx), N, H, W, C, 1, stream())
        tp.add_grad(x, gx)
    tp.steps.append(bwd)
    return y


def add(tp: Tape, *xs):
    y = xs[0]
    for x in xs[1:]:
        y = ops.ew(1, y, x, alpha=1.0)
    tp.keep.append(y)

    def bwd():
        g = tp.grad(y)
        if g is None:
            return
        for x in xs:
            tp.add_grad(x, g)
    tp.steps.append(bwd)
    return y


def bifpn_forward(tp: Tape, bifpn, feats, last_pooling: bool):
    """feats: [p3..p7] NHWC backbone taps.  Returns p3_dec."""
    def cbs(xs, seq):      # Sequential(Conv2d, BatchNorm2d, swish)
        z, st = conv1x1(tp, xs, seq[0], want_stats=tp.training)
        return bn_swish(tp, z, st, seq[1])

    def cs(xs, seq):       # Sequential(Conv2d, swish)
        z, _ = conv1x1(tp, xs, seq[0])
        return swish(tp, z)

    p3, p4, p5, p6, p7 = (cbs([f], getattr(bifpn, f"inp{i}")) for i, f in zip(range(3, 8), feats))
    for layer in bifpn.BIFPN_Layers:
        p6_mid = cs([p6, p7], layer.convp67)
        p5_mid = cs([p5, resize(tp, p6_mid, p5.shape[1], p5.shape[2])], layer.convp56)
        p4_mid = cs([p4, p5], layer.convp45)
        p3_out = cs([p3, resize(tp, p4_mid, p3.shape[1], p3.shape[2])], layer.convp34)
        p4_out = cbs([add(tp, p4, p4_mid, resize(tp, avgpool(tp, p3_out), p4.shape[1], p4.shape[2]))], layer.out4)
        p5_out = cbs([add(tp, p5, p5_mid, p4_out)], layer.out5)
        if last_pooling:
            p6_out = cbs([add(tp, p6, p6_mid, resize(tp, avgpool(tp, p5_out), p6.shape[1], p6.shape[2]))], layer.out6)
        else:
            p6_out = cbs([add(tp, p6, p6_mid, p5_out)], layer.out6)
        p7_out = cbs([add(tp, p7, p6_out)], layer.out7)
        p3, p4, p5, p6, p7 = p3_out, p4_out, p5_out, p6_out, p7_out
    return p3


class _BIFPN_Layer(nn.Module):
    """Parameter container of MuSCLe.py:30-44."""

    def __init__(self, c):
        super().__init__()
        for nm in ("convp67", "convp56", "convp45", "convp34"):
            setattr(self, nm, nn.Sequential(nn.Conv2d(2 * c, c, 1)))
        for nm in ("out4", "out5", "out6", "out7"):
            setattr(self, nm, nn.Sequential(nn.Conv2d(c, c, 1), nn.BatchNorm2d(c)))


class BIFPN(nn.Module):
    """Parameter container of MuSCLe.py:115-135 (forward never called; see bifpn_forward)."""

    def __init__(self, tap_channels, layers, c):
        super().__init__()
        for i, ci in zip(range(3, 8), tap_channels[2:]):
            setattr(self, f"inp{i}", nn.Sequential(nn.Conv2d(ci, c, 1), nn.BatchNorm2d(c)))
        self.BIFPN_Layers = nn.ModuleList([_BIFPN_Layer(c) for _ in range(layers)])
