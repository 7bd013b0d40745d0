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
        self._device_scalars = False
        self._touched = []

    # ---- hipGraph support: lr and the bias corrections read from device memory --------------------------------------
    def use_device_scalars(self, on: bool = True):
        """With this on, a step advances a per-group step count that lives on the device and hands the kernel
        {lr, 1 - b1^t, sqrt(1 - b2^t)} through a 3-float device buffer, so the enqueued work has no host-side scalar
        that changes from step to step and can be replayed from a captured graph (muscle_amd.graph).  Requires what the
        MCL loop has anyway: every parameter that receives a gradient has taken the same number of steps."""
        self._device_scalars = bool(on)
        return self

    def _dev_state(self, group, t_host: int):
        st = group.get("_dyn")
        if st is None:
            dev = group["_dev"]
            st = {"dyn": torch.zeros(3, dtype=torch.float32, device=dev),
                  "t": torch.full((), float(t_host), dtype=torch.float64, device=dev),
                  "lr": torch.full((), float(group["lr"]), dtype=torch.float64, device=dev)}
            group["_dyn"] = st
        return st

    def sync_lr(self):
        """Push group['lr'] (an lr scheduler's product) into the device scalar; call before each replay."""
        for group in self.param_groups:
            st = group.get("_dyn")
            if st is not None:
                st["lr"].fill_(float(group["lr"]))

    def note_replayed_step(self):
        """Host bookkeeping after a graph replay: the parameters the captured step touched took one more step."""
        for group, idx in self._touched:
            steps = group["_steps"]
            for q in idx:
                steps[q] += 1

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

    # ---- checkpointing: torch.optim.Adam's layout (state[idx] = {step, exp_avg, exp_avg_sq}), never the arenas ------
    _PRIVATE = ("_arena", "_offs", "_sizes", "_m", "_v", "_steps", "_dev", "_dyn")

    def state_dict(self):
        """{'state': {param index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} - the same shape
        torch.optim.Adam produces, so a resumed run keeps its moments, step counts and bias correction.  The flat
        arenas (a full copy of the parameters) are not serialised."""
        state, groups, base = {}, [], 0
        for group in self.param_groups:
            n = len(group["params"])
            if group.get("_arena") is not None:
                for i, p in enumerate(group["params"]):
                    if group["_steps"][i] == 0:
                        continue
                    o, k = group["_offs"][i], p.numel()
                    state[base + i] = {"step": int(group["_steps"][i]),
                                       "exp_avg": group["_m"][o:o + k].view(p.shape).clone(),
                                       "exp_avg_sq": group["_v"][o:o + k].view(p.shape).clone()}
            g = {k: v for k, v in group.items() if k != "params" and k not in self._PRIVATE}
            g["params"] = list(range(base, base + n))
            groups.append(g)
            base += n
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        base = 0
        self._pending = {}
        for group, saved in zip(self.param_groups, groups):
            if len(saved["params"]) != len(group["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            for k, v in saved.items():
                if k != "params" and k not in self._PRIVATE:
                    group[k] = v
            for k in self._PRIVATE:
                group.pop(k, None)                      # re-flatten on the next step, then apply the pending moments
            for i in range(len(group["params"])):
                st = sd["state"].get(base + i, sd["state"].get(str(base + i)))
                if st is not None:
                    self._pending[(id(group), i)] = st
            base += len(group["params"])

    def _apply_pending(self, group):
        pend = getattr(self, "_pending", None)
        if not pend:
            return
        for i, p in enumerate(group["params"]):
            st = pend.pop((id(group), i), None)
            if st is None:
                continue
            o, k = group["_offs"][i], p.numel()
            group["_m"][o:o + k].copy_(torch.as_tensor(st["exp_avg"]).reshape(-1).to(group["_m"].device, torch.float32))
            group["_v"][o:o + k].copy_(torch.as_tensor(st["exp_avg_sq"]).reshape(-1).to(group["_v"].device, torch.float32))
            group["_steps"][i] = int(st["step"])

    def _is_flat(self, group):
        a = group.get("_arena")
        if a is None or group["params"][0].device != group["_dev"]:
            return False
        p0 = group["params"][0]
        return p0.data_ptr() == a.data_ptr() + 4 * group["_offs"][0]

    @torch.no_grad()
    def step(self, closure=None):
        self._touched = []
        for group in self.param_groups:
            if not group["params"]:
                continue
            if not group["params"][0].is_cuda:
                raise RuntimeError("FusedAdam runs on the HIP kernel only: move the model to the GPU first")
            if not self._is_flat(group):
                self._flatten(group)
                self._apply_pending(group)
            b1, b2 = group["betas"]
            ps, offs, sizes, steps = group["params"], group["_offs"], group["_sizes"], group["_steps"]
            arena, m, v = group["_arena"], group["_m"], group["_v"]
            dyn, touched = None, []
            use_dev = self._device_scalars
            capturing = torch.cuda.is_current_stream_capturing()
            if not use_dev:
                group.pop("_dyn", None)          # a later device-scalar step re-seeds its count from the host's
            else:
                live = [steps[q] for q, p in enumerate(ps) if p.grad is not None]
                if not live:
                    continue
                if min(live) != max(live):
                    # e.g. phase 2 of the MCL step never gives fc.weight a gradient, so from the first ep >= 8 iteration
                    # on its count trails the others.  One device-side count cannot express that: an eager step falls
                    # back to the host-side scalars of each run (exactly torch.optim.Adam); a capture cannot.
                    if capturing:
                        raise RuntimeError("FusedAdam device scalars need one step count for all parameters with gradients "
                                           "to be captured into a graph")
                    use_dev = False
            if use_dev:
                st = self._dev_state(group, live[0])
                if not capturing:
                    # an eager step knows the truth on the host: the count (replays and host-scalar steps may have
                    # moved it) and the scheduler's current lr
                    st["t"].fill_(float(live[0]))
                    st["lr"].fill_(float(group["lr"]))
                # same expressions as the host path (double, then rounded to float by the kernel argument)
                st["t"].add_(1.0)
                st["dyn"][0] = st["lr"]
                st["dyn"][1] = 1.0 - torch.pow(torch.full_like(st["t"], b1), st["t"])
                st["dyn"][2] = torch.sqrt(1.0 - torch.pow(torch.full_like(st["t"], b2), st["t"]))
                dyn = st["dyn"]
                self._touched.append((group, touched))
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
                     1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), ptr(dyn), stream())
                for q in range(i, j):
                    steps[q] = t
                    touched.append(q)
                i = j
        return None
