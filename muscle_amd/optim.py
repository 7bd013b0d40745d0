"""Fused Adam for the MCL step — torch.optim.Adam(lr, weight_decay) semantics (train_mcl.py:134): L2 decay added
to the gradient (not AdamW), bias correction, per-parameter step counts, parameters whose grad is None skipped.

MI355X layout: parameters, both moments and (from MuSCLe's backward) gradients live in flat fp32 arenas, so a step
is a handful of launches of one streaming kernel over contiguous ranges instead of ~1200 small ones.
"""
from __future__ import annotations

import math
from typing import List

import torch

from ._lib import call, ptr, stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = None

    def _flatten(self, group):
        """Move every parameter of the group into one arena (views keep the nn.Parameter objects intact)."""
        ps: List[torch.nn.Parameter] = [p for p in group["params"]]
        dev = ps[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in ps]
        arena = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        offs, off = [], 0
        for p, s in zip(ps, sizes):
            arena[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = arena[off:off + p.numel()].view(p.shape)
            offs.append(off)
            off += s
        group["_arena"] = arena
        group["_offs"] = offs
        group["_sizes"] = sizes
        group["_m"] = torch.zeros_like(arena)
        group["_v"] = torch.zeros_like(arena)
        group["_steps"] = [0] * len(ps)
        group["_dev"] = dev

    def _is_flat(self, group):
        a = group.get("_arena")
        if a is None or group["params"][0].device != group["_dev"]:
            return False
        p0 = group["params"][0]
        return p0.data_ptr() == a.data_ptr() + 4 * group["_offs"][0]

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            if not group["params"]:
                continue
            if not group["params"][0].is_cuda:
                raise RuntimeError("FusedAdam runs on the HIP kernel only: move the model to the GPU first")
            if not self._is_flat(group):
                self._flatten(group)
            b1, b2 = group["betas"]
            ps, offs, sizes, steps = group["params"], group["_offs"], group["_sizes"], group["_steps"]
            arena, m, v = group["_arena"], group["_m"], group["_v"]
            # runs of consecutive parameters that have gradients laid out contiguously and share a step count
            i, n = 0, len(ps)
            while i < n:
                p = ps[i]
                if p.grad is None:
                    i += 1
                    continue
                g0 = p.grad
                j, span = i, 0
                while (j < n and ps[j].grad is not None and steps[j] == steps[i]
                       and ps[j].grad.data_ptr() == g0.data_ptr() + 4 * (offs[j] - offs[i])
                       and ps[j].grad.is_contiguous()):
                    span = offs[j] - offs[i] + ps[j].numel()
                    j += 1
                if j == i:       # gradient not where the arena layout expects it: single-tensor launch
                    j, span = i + 1, p.numel()
                    gptr = ptr(p.grad.contiguous())
                else:
                    gptr = g0.data_ptr()
                t = steps[i] + 1
                call("mx_adam", arena.data_ptr() + 4 * offs[i], gptr, m.data_ptr() + 4 * offs[i], v.data_ptr() + 4 * offs[i],
                     span, float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                     1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), stream())
                for q in range(i, j):
                    steps[q] = t
                i = j
        return None
