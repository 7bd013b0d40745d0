"""MCL loss callables — drop-in for the reference's `src/loss_multilabel.py` live functions
(FocalLoss :68-91, Log_Sum_Exp_Pairwise_Loss :24-33, image_level_contrast :36-66), on HIP kernels.

Each callable keeps the reference's signature and return convention (0-dim tensors; IMC returns the
Python float 0.0 when no anchor row qualifies, which callers test with torch.is_tensor,
train_mcl.py:194).  Inputs must be CUDA tensors; gradients flow through torch.autograd.Functions whose
backward is the analytic gradient computed by the same kernel launch as the forward.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from ._lib import call, ptr, stream


def _prep(t):
    return t.contiguous().float()


class _ClsLoss(torch.autograd.Function):
    """mode 0 focal(p, y), 1 soft margin(x, y), 2 pairwise(p, y) -> [N]."""

    @staticmethod
    def forward(ctx, x, y, mode):
        x, y = _prep(x), _prep(y)
        N, C = x.shape
        loss = torch.zeros(N if mode == 2 else 1, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        call("mx_cls_loss", mode, ptr(x), C, ptr(y), C, ptr(loss), ptr(grad), C, N, C, stream())
        ctx.save_for_backward(grad)
        ctx.mode = mode
        return loss if mode == 2 else loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        if ctx.mode == 2:
            return grad * g.reshape(-1, 1), None, None
        return grad * g, None, None


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _prep(x)
        N, C = x.shape
        y = torch.empty_like(x)
        call("mx_cls_loss", 3, ptr(x), C, ptr(x), C, None, ptr(y), C, N, C, stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _prep(g)
        N, C = y.shape
        out = torch.empty_like(y)
        call("mx_cls_loss", 4, ptr(y), C, ptr(g), C, None, ptr(out), C, N, C, stream())
        return out


def sigmoid(x):
    """torch.sigmoid on a [N, C] slice (train_mcl.py:180,182) as a HIP kernel."""
    return _Sigmoid.apply(x)


class FocalLoss(nn.Module):
    def __init__(self, gamma=2, alpha=0.5, size_average=True, weight=None):
        super().__init__()
        if gamma != 2 or alpha != 0.5:
            raise NotImplementedError("the HIP kernel implements the reference defaults gamma=2, alpha=0.5")
        self.gamma, self.alpha, self.size_average, self.weight = gamma, alpha, size_average, weight

    def forward(self, input, target):
        return _ClsLoss.apply(input, target, 0)


class MultiLabelSoftMarginLoss(nn.Module):
    """nn.MultiLabelSoftMarginLoss() as used at train_mcl.py:146,181 (mean reduction, no weights)."""

    def forward(self, input, target):
        return _ClsLoss.apply(input, target, 1)


def Log_Sum_Exp_Pairwise_Loss(pred, labels):
    return _ClsLoss.apply(pred, labels, 2)


class _IMC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, label):
        emb, label = _prep(emb), _prep(label)
        N, D = emb.shape
        out = torch.empty(2, dtype=torch.float32, device=emb.device)
        gemb = torch.empty_like(emb)
        ws = torch.empty(N * D + 2 * N * N + 4 * N + 16, dtype=torch.float32, device=emb.device)
        call("mx_imc", ptr(emb), ptr(label), N, D, label.shape[1], ptr(out), ptr(gemb), ptr(ws), stream())
        ctx.save_for_backward(gemb)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g, _):
        (gemb,) = ctx.saved_tensors
        return gemb * g, None


def image_level_contrast_nosync(emb, label):
    """(loss, info) with loss a 0-dim tensor that is exactly 0 (and has zero gradient) when no anchor row
    qualifies; info = [loss, #valid rows] on the device.  No host synchronisation."""
    return _IMC.apply(emb, label)


def image_level_contrast(emb, label):
    """The IMC loss with the reference's return convention: a tensor, or the Python float 0.0 when no
    anchor row qualifies (one device->host read of the valid-row count; the reference loop has O(N^2))."""
    loss, info = _IMC.apply(emb, label)
    if float(info[1]) == 0.0:
        return 0.0
    return loss
