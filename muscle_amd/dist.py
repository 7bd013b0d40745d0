"""Data parallelism for the MCL step: one process per GPU, gradient averaging with RCCL over xGMI.

The reference is single-GPU; the semantics fixed in SURVEY.md §8(e) are: replica-local
BatchNorm statistics, replica-local losses (IMC pairs, ER k, EMD matching), one gradient average per
optimizer step.  MuSCLe's backward leaves all live gradients in ONE flat fp32 arena
(`model.last_grad_sink.arena`, 248.8 MB for B7), laid out in forward order: stem, blocks 0..54, heads.  Backward
fills it back to front, so the exchange is cut into fixed ~25 MB chunks and a chunk is handed to RCCL as soon as the
backward of the blocks it covers has been enqueued (`async_op=True`: the collective runs on RCCL's stream behind the
kernels enqueued so far and overlaps the rest of backward); the hook before the optimizer step launches what is left
(the first chunk) and makes the compute stream wait for all of them.  No unused-parameter bookkeeping: dead
parameters are not in the arena.  Chunk boundaries do not depend on timing and every element is reduced exactly once
with the same operation, so the result is identical to one all-reduce of the whole arena.
"""
from __future__ import annotations

from typing import List

import torch.distributed as dist

CHUNK_BYTES = 25 * 1024 * 1024


class GradAverager:
    """grad_hook for muscle_amd.mcl_step: averages the flat gradient arena across ranks.

    `attach(model)` additionally lets the model's backward report progress (arena[lo:] complete), which starts the
    chunks early; without it everything is launched from the hook (same result, no overlap)."""

    def __init__(self, group=None, chunk_bytes: int = CHUNK_BYTES, single_rank_collectives: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a one-rank group normally skips the exchange; `single_rank_collectives` issues it anyway (the average over one rank is the
        # identity): the only way to run the RCCL calls and their stream ordering on a one-GPU box (tests/test_gpu_dist.py)
        self._force = bool(single_rank_collectives and dist.is_initialized())
        self.bytes_reduced = 0
        self.chunk = max(1, chunk_bytes // 4)          # elements
        self._sink = None
        self._next = -1                                # highest chunk index not launched yet
        self._works: List = []
        self.launched_early = 0                        # chunks that went out before the hook (statistics / tests)

    @property
    def active(self) -> bool:
        return self.world > 1 or self._force

    # ---- wiring ---------------------------------------------------------------------------------------------
    def attach(self, model):
        model.grad_ready_callback = self.on_ready
        return self

    def _nccl(self) -> bool:
        return dist.get_backend(self.group) == "nccl"

    def _begin(self, sink):
        self._sink = sink
        n = sink.arena.numel()
        self._next = (n + self.chunk - 1) // self.chunk - 1
        self._works = []

    def _launch(self, k: int):
        arena = self._sink.arena
        part = arena[k * self.chunk: min((k + 1) * self.chunk, arena.numel())]
        if self._nccl():
            # RCCL over xGMI; async: its stream waits for the kernels enqueued so far, the compute stream goes on
            self._works.append(dist.all_reduce(part, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:
            # gloo: CPU unit tests, and single-GPU rehearsals of the multi-process flow (no AVG, host staging)
            buf = part.cpu() if part.is_cuda else part
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            buf.div_(self.world)
            if part.is_cuda:
                part.copy_(buf)
        self.bytes_reduced += part.numel() * 4

    # ---- progress from the backward: arena[lo:] is complete on the current stream ---------------------------------
    def on_ready(self, sink, lo: int):
        if not self.active:
            return
        if sink is not self._sink:
            self._begin(sink)
        while self._next >= 1 and self._next * self.chunk >= lo:     # chunk 0 always waits for the hook
            self._launch(self._next)
            self._next -= 1
            self.launched_early += 1

    # ---- the hook: after backward, before optimizer.step() ----------------------------------------------------------
    def __call__(self, model, phase: int):
        if not self.active:
            return
        sink = model.last_grad_sink
        check = getattr(sink, "check_aliases", None)
        if check is not None:
            check(model)                 # a gradient outside the arena would silently miss the average
        if sink is not self._sink:
            self._begin(sink)
        while self._next >= 0:
            self._launch(self._next)
            self._next -= 1
        for w in self._works:
            w.wait()                     # the compute stream waits for RCCL's stream; the host does not block
        self._works = []
        self._sink = None


def broadcast_parameters(model, src: int = 0, group=None, single_rank_collectives: bool = False):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not single_rank_collectives):
        return
    gloo = dist.get_backend(group) != "nccl"
    for t in list(model.parameters()) + list(model.buffers()):
        if gloo and t.is_cuda:
            buf = t.data.cpu()
            dist.broadcast(buf, src=src, group=group)
            t.data.copy_(buf)
        else:
            dist.broadcast(t.data, src=src, group=group)


def sync_buffers_from_rank0(model, group=None):
    """BatchNorm running statistics stay replica-local during training (no SyncBN); a checkpoint takes
    rank 0's.  Call this before saving (or before an evaluation that must agree across ranks) to give every rank those
    statistics, as torch DDP's broadcast_buffers would at each forward."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    gloo = dist.get_backend(group) != "nccl"
    for t in model.buffers():
        if gloo and t.is_cuda:
            buf = t.data.cpu()
            dist.broadcast(buf, src=0, group=group)
            t.data.copy_(buf)
        else:
            dist.broadcast(t.data, src=0, group=group)
