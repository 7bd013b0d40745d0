"""Data parallelism for the MCL step: one process per GPU, gradient averaging with RCCL over xGMI.

The reference is single-GPU; the semantics fixed in SURVEY.md §8(e) are: replica-local
BatchNorm statistics, replica-local losses (IMC pairs, ER k, EMD matching), one gradient average per
optimizer step.  MuSCLe's backward leaves all live gradients in ONE flat fp32 arena
(`model.last_grad_sink.arena`), so the exchange is a single large all-reduce (248.8 MB for B7) with no
bucketing logic and no unused-parameter bookkeeping: dead parameters are simply not in the arena.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradAverager:
    """grad_hook for muscle_amd.mcl_step: averages the flat gradient arena across ranks."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bytes_reduced = 0

    def __call__(self, model, phase: int):
        if self.world == 1:
            return
        check = getattr(model.last_grad_sink, "check_aliases", None)
        if check is not None:
            check(model)                 # a gradient outside the arena would silently miss the average
        arena = model.last_grad_sink.arena
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(arena, op=dist.ReduceOp.AVG, group=self.group)         # RCCL over xGMI
        else:
            # gloo: CPU unit tests, and single-GPU rehearsals of the multi-process flow (no AVG, host staging)
            buf = arena.cpu() if arena.is_cuda else arena
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            buf.div_(self.world)
            if arena.is_cuda:
                arena.copy_(buf)
        self.bytes_reduced += arena.numel() * 4


def broadcast_parameters(model, src: int = 0, group=None):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    gloo = dist.get_backend(group) != "nccl"
    for t in list(model.parameters()) + list(model.buffers()):
        if gloo and t.is_cuda:
            buf = t.data.cpu()
            dist.broadcast(buf, src=src, group=group)
            t.data.copy_(buf)
        else:
            dist.broadcast(t.data, src=src, group=group)
