"""hipGraph replay of the MCL loop body for launch-bound configurations (BASELINE.json configs[2]: B0 / 448 / batch 16).

One B0 step is ~1 900 kernel launches of a few microseconds each; enqueued from Python the host is the limit, not the
GPU.  Phase 1 of `mcl_step` (train_mcl.py:153-199: forward, classification + ER (+ IMC) losses, backward, Adam) has no
data-dependent host decision once
  * ER's top-k count k = int(0.2 * label.sum() * H * W) is taken on the device (`er_loss_lowres` with a tensor),
  * IMC uses the device-side fall-through (`image_level_contrast_nosync`),
  * Adam reads lr and its bias corrections from device memory (`FusedAdam.use_device_scalars`),
so its launches are captured once into a hipGraph (HIP stream capture through torch.cuda.CUDAGraph: plumbing) and
replayed per step with the batch copied into static input buffers.  Drop-connect draws come from torch's generator,
which registers its Philox offset with the graph, so every replay draws fresh numbers.

Phase 2 (ep >= 8: PixPro + EMD) builds its crop geometry on the host from coord1/coord2 and is not captured; a
`GraphedStep` refuses ep >= 8.  Epoch gate 4 (IMC on/off) changes the captured work: one GraphedStep per gate value.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .train_step import mcl_step

_KEYS = ("img", "label")


class GraphedStep:
    """step = GraphedStep(model, optimizer, ep); out = step(batch) for every batch of the same shapes.

    The first `warmup` calls run eagerly (they are ordinary training steps on the caller's batches: lazily created
    workspaces, the optimizer's flat arenas and the allocator's pools settle there); the next call captures and
    replays; later calls copy the batch in and replay.  The returned dict holds the same seven loss terms as
    `mcl_step`; the tensors in it are overwritten by the next replay."""

    def __init__(self, model, optimizer, ep: int, warmup: int = 2, grad_hook=None):
        if ep >= 8:
            raise ValueError("GraphedStep covers phase 1 of the MCL step (ep < 8); phase 2 plans crops on the host")
        if grad_hook is not None:
            raise ValueError("GraphedStep is the single-GPU launch-bound path; multi-GPU runs use mcl_step + GradAverager")
        if warmup < 1:
            raise ValueError("at least one eager step must precede the capture (it creates the optimizer's device state)")
        self.model, self.opt, self.ep, self.warmup = model, optimizer, int(ep), int(warmup)
        optimizer.use_device_scalars(True)
        self.calls = 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static: Dict[str, torch.Tensor] = {}
        self.out = None
        self.replays = 0

    def _body(self, batch):
        return mcl_step(self.model, self.opt, batch, self.ep, valid_channel=batch["label"].float().sum())

    def close(self):
        """Hand the optimizer back to host-side scalars (after the last replay: e.g. before the ep >= 8 iterations, which
        run through mcl_step).  Also what leaving a `with GraphedStep(...) as step:` block does."""
        self.opt.use_device_scalars(False)
        self.graph = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __call__(self, batch: Dict[str, torch.Tensor]):
        self.calls += 1
        self.opt.sync_lr()                          # an lr scheduler may have stepped since the last call
        if self.graph is None and self.calls <= self.warmup:
            return self._body(batch)
        if self.graph is None:
            self.static = {k: batch[k].clone() for k in _KEYS}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    self.out = self._body(self.static)
            torch.cuda.current_stream().wait_stream(side)
            # capture enqueued nothing: undo the host-side step count the captured optimizer.step() advanced
            for group, idx in self.opt._touched:
                for q in idx:
                    group["_steps"][q] -= 1
            self.graph = g
        else:
            for k in _KEYS:
                if batch[k].shape != self.static[k].shape:
                    raise ValueError(f"GraphedStep was captured for {k} of shape {tuple(self.static[k].shape)}, "
                                     f"got {tuple(batch[k].shape)}")
                self.static[k].copy_(batch[k], non_blocking=True)
        self.opt.sync_lr()
        self.graph.replay()
        self.opt.note_replayed_step()
        self.replays += 1
        return self.out
