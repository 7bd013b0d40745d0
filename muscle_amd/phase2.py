"""Phase 2 of the MCL loop body (train_mcl.py:201-229): PixPro on the two views (epoch >= 8) and the
Sinkhorn-EMD crop matching (epoch >= 12), with the second optimizer step — on the HIP kernels of csrc/phase2.hip.

Public, reference-named entry points: cam_maxnorm (train_mcl.py:21-28), PixPro (loss_multilabel.py:93-105),
get_dynamic_crops (torchutils.py:217-291), EMD (loss_multilabel.py:108-338, 'dynamic' mode).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from ._lib import call, ptr, stream
from .train_step import cam_softmaxnorm

FP = 24


# ---------------------------------------------------------------------------
# cam_maxnorm
# ---------------------------------------------------------------------------
class _MaxNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        N, K, H, W = x.shape
        out = torch.empty_like(x)
        stats = torch.empty(N * K * 4, dtype=torch.float32, device=x.device)
        call("mx_maxnorm", ptr(x), None, ptr(out), ptr(stats), N * K, H * W, 0, stream())
        ctx.save_for_backward(x, stats)
        return out

    @staticmethod
    def backward(ctx, g):
        x, stats = ctx.saved_tensors
        N, K, H, W = x.shape
        gx = torch.empty_like(x)
        call("mx_maxnorm", ptr(x), ptr(g.contiguous()), ptr(gx), ptr(stats), N * K, H * W, 1, stream())
        return gx


def cam_maxnorm(cams):
    return _MaxNorm.apply(cams)


# ---------------------------------------------------------------------------
# PixPro
# ---------------------------------------------------------------------------
class _PixPro(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, c1, c2, mask):
        f1, f2 = f1.contiguous().float(), f2.contiguous().float()
        N, K, H, W = f1.shape
        dev = f1.device
        c1 = c1.to(dev, torch.int64).contiguous()
        c2 = c2.to(dev, torch.int64).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        g1 = torch.zeros_like(f1)
        ws = torch.empty(N, dtype=torch.int64, device=dev)
        call("mx_pixpro", ptr(f1), ptr(f2), ptr(mask.contiguous().float()) if mask is not None else None, ptr(c1), ptr(c2),
             ptr(loss), ptr(g1), N, K, H, W, ws.data_ptr(), 8 * N, stream())
        ctx.save_for_backward(g1)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (g1,) = ctx.saved_tensors
        return g1 * g, None, None, None, None


def PixPro(fm1s, fm2s, coord1s, coord2s, mask=None):
    """1 - mean_b mean_window cosine_similarity(fm1[b, :, window1], fm2[b, :, window2].detach(), dim=0).
    `mask` ([N,K], optional) multiplies both maps first (the label mask of train_mcl.py:209) inside the kernel."""
    return _PixPro.apply(fm1s, fm2s, coord1s, coord2s, mask)


# ---------------------------------------------------------------------------
# F.normalize(dim=1)
# ---------------------------------------------------------------------------
class _ChanNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        N, K, H, W = x.shape
        out = torch.empty_like(x)
        call("mx_chan_l2norm", ptr(x), None, ptr(out), N, K, H * W, 0, stream())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        N, K, H, W = x.shape
        out = torch.empty_like(x)
        call("mx_chan_l2norm", ptr(x), ptr(g.contiguous()), ptr(out), N, K, H * W, 1, stream())
        return out


def normalize_channels(x):
    return _ChanNorm.apply(x)


# ---------------------------------------------------------------------------
# dynamic crops
# ---------------------------------------------------------------------------
def crop_geometry(h: int, w: int, randint=np.random.randint):
    """Draw order of torchutils.py:240-254 (np.random, data dependent).  None = sample skipped."""
    if h < 15 or w < 15 or h / w > 5 or w / h > 5:
        return None
    lh = randint(h // 3, h // 2 + 1)
    lw = randint(w // 3, w // 2 + 1)
    while lh < 5 or lw < 5:
        lh = randint(h // 3, h // 2 + 1)
        lw = randint(w // 3, w // 2 + 1)
    sh = randint(lh // 2, lh + 1)
    sw = randint(lw // 2, lw + 1)
    return int(lh), int(lw), int(sh), int(sw)


class CropSet:
    """Packed result of get_dynamic_crops for one view: features [pixels, 24] on the device plus host-side
    geometry.  Behaves like the reference's list-of-lists for inspection (`len`, `cs[i][j]` -> detached
    [1,K,h,w] tensor) and carries what EMD needs."""

    def __init__(self, feat, per_sample, K):
        self.feat = feat                    # shared [P, 24] buffer (both views)
        self.per_sample = per_sample        # list (samples with crops) of list of (offset, h, w)
        self.K = K

    def __len__(self):
        return len(self.per_sample)

    def __getitem__(self, i):
        out = []
        for off, h, w in self.per_sample[i]:
            out.append(self.feat[off:off + h * w, :self.K].detach().t().reshape(1, self.K, h, w))
        return out


class _CropPlan:
    """Everything get_dynamic_crops decides on the host: resize / pool tables, offsets, pair table."""
    pass


_copy_streams: dict = {}


class HostCoords:
    """coord1 / coord2 on their way to the host.  get_dynamic_crops decides the crop windows on the host
    (torchutils.py:217-291 is Python), and a plain `.cpu()` in the middle of the step makes the host wait for everything
    enqueued so far - phase 1 and both phase-2 forwards - and then enqueue the ~1000 launches of the backward with the GPU
    idle.  The coordinates are inputs of the step: mcl_step starts this copy on a stream of its own before it enqueues
    anything, and planning waits for that copy alone."""

    def __init__(self, coord1: torch.Tensor, coord2: torch.Tensor):
        dev = coord1.device
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        side = _copy_streams.get(key)
        if side is None:
            side = _copy_streams[key] = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))          # whatever produced the batch
        self.host = []
        with torch.cuda.stream(side):
            for c in (coord1, coord2):
                c = c.detach()
                h = torch.empty(c.shape, dtype=c.dtype, pin_memory=True)
                h.copy_(c, non_blocking=True)
                c.record_stream(side)
                self.host.append(h)
            self.done = side.record_event()

    def numpy(self):
        self.done.synchronize()
        return tuple(h.numpy().astype(np.int64) for h in self.host)


def prefetch_coords(batch):
    """Start the device-to-host copy of the view coordinates (None when they already live on the host)."""
    c1, c2 = batch["coord1"], batch["coord2"]
    return HostCoords(c1, c2) if c1.is_cuda and c2.is_cuda else None


def _plan_crops(coord1, coord2, geometry, host: Optional[HostCoords] = None):
    if host is not None:
        c1, c2 = host.numpy()
    else:
        c1 = coord1.detach().cpu().numpy().astype(np.int64)
        c2 = coord2.detach().cpu().numpy().astype(np.int64)
    t1, t2, pool = [], [], []            # resize tables for view 1 / view 2, pool table
    per1, per2, bidx = [], [], []
    fixups = []                          # (list, index) entries whose offset must be shifted into the pooled region
    off = 0
    crops_meta = []
    for b in range(c1.shape[0]):
        h, w = int(c1[b, 2]), int(c1[b, 3])
        g = geometry[b] if geometry is not None else crop_geometry(h, w)
        if g is None:
            continue
        lh, lw, sh, sw = g
        gh, gw = h / sh, w / sw
        s1, s2 = [], []
        for i1 in range(0, h, sh):
            for j1 in range(0, w, sw):
                if i1 + lh > h or j1 + lw > w:
                    continue
                rh, rw = round(h / gh), round(w / gw)
                if rh < 7 or rw < 7:
                    continue
                t1.append([b, int(c1[b, 0]) + i1, int(c1[b, 1]) + j1, lh, lw, rh, rw, off])
                s1.append([off, rh, rw, rh > 28 or rw > 28])
                off += rh * rw
        if not s1:
            continue
        for i2 in range(0, h - 1, h // 2):
            for j2 in range(0, w - 1, w // 2):
                ph, pw = min(h // 2, h - i2), min(w // 2, w - j2)
                t2.append([b, int(c2[b, 0]) + i2, int(c2[b, 1]) + j2, ph, pw, ph, pw, off])
                s2.append([off, ph, pw, True])
                off += ph * pw
        crops_meta.append((s1, s2))
        bidx.append(b)
    nA = off
    # pooled copies live after the resized ones
    for s1, s2 in crops_meta:
        for lst in (s1, s2):
            for e in lst:
                if e[3]:
                    ph, pw = e[1] // 4, e[2] // 4
                    pool.append([e[0], e[1], e[2], off])
                    e[0], e[1], e[2] = off, ph, pw
                    off += ph * pw
        per1.append([(e[0], e[1], e[2]) for e in s1])
        per2.append([(e[0], e[1], e[2]) for e in s2])
    pairs = []
    for s, (a, bb) in enumerate(zip(per1, per2)):
        for (o1, h1, w1) in a:
            for (o2, h2, w2) in bb:
                pairs.append([o1, h1 * w1, o2, h2 * w2, s, 0])
    p = _CropPlan()
    p.t1, p.t2, p.pool, p.pairs = t1, t2, pool, pairs
    p.per1, p.per2, p.bidx, p.total, p.nA = per1, per2, bidx, off, nA
    return p


def _table(rows, dev):
    """int32 table to the device through pinned memory: a copy from pageable memory makes the host wait for the stream."""
    return torch.tensor(rows, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)


class _Crops(torch.autograd.Function):
    """feat = packed crops of (x1 -> gradient, x2 -> no gradient)."""

    @staticmethod
    def forward(ctx, x1, x2, plan):
        x1, x2 = x1.contiguous().float(), x2.contiguous().float()
        N, K, H, W = x1.shape
        dev = x1.device
        feat = torch.zeros(plan.total + 4, FP, dtype=torch.float32, device=dev)
        it = lambda rows: _table(rows, dev)  # noqa: E731
        tabs = {"t1": it(plan.t1), "t2": it(plan.t2), "pool": it(plan.pool) if plan.pool else None}
        call("mx_crop_resize", ptr(x1), ptr(tabs["t1"]), len(plan.t1), ptr(feat), K, H, W, stream())
        call("mx_crop_resize", ptr(x2), ptr(tabs["t2"]), len(plan.t2), ptr(feat), K, x2.shape[2], x2.shape[3], stream())
        if plan.pool:
            call("mx_avgpool4", ptr(feat), ptr(tabs["pool"]), len(plan.pool), ptr(feat), 0, stream())
        ctx.tabs, ctx.plan, ctx.shape = tabs, plan, (N, K, H, W)
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        plan, tabs = ctx.plan, ctx.tabs
        N, K, H, W = ctx.shape
        g = gfeat.contiguous().clone()
        if plan.pool:
            call("mx_avgpool4", ptr(g), ptr(tabs["pool"]), len(plan.pool), ptr(g), 1, stream())
        gx1 = torch.zeros(N, K, H, W, dtype=torch.float32, device=g.device)
        call("mx_crop_resize_bwd", ptr(g), ptr(tabs["t1"]), len(plan.t1), ptr(gx1), N, K, H, W, stream())
        return gx1, None, None


def get_dynamic_crops(x1, coord1, x2, coord2, geometry: Optional[Sequence] = None, host_coords: Optional[HostCoords] = None):
    """torchutils.get_dynamic_crops.  Returns (crops1, crops2, batch_indices) where crops1/crops2 are CropSet
    views of one packed device buffer (gradient flows to x1 only, as x2 arrives detached at train_mcl.py:220).
    `geometry` replays recorded draws; otherwise np.random is consumed in the reference's order.
    `host_coords`: a copy of the coordinates started earlier (prefetch_coords), so that planning does not drain the stream."""
    plan = _plan_crops(coord1, coord2, geometry, host_coords)
    if not plan.per1:
        return CropSet(None, [], x1.shape[1]), CropSet(None, [], x1.shape[1]), []
    feat = _Crops.apply(x1, x2.detach(), plan)
    cs1, cs2 = CropSet(feat, plan.per1, x1.shape[1]), CropSet(feat, plan.per2, x1.shape[1])
    cs1.plan = cs2.plan = plan
    return cs1, cs2, plan.bidx


# ---------------------------------------------------------------------------
# EMD
# ---------------------------------------------------------------------------
class _EMD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, plan):
        dev = feat.device
        # one workgroup per pair: longest first, so the short pairs fill the tail; column 5 keeps the reference's enumeration
        # order for the tie-break of the per-sample minimum
        rows = sorted(([*p[:5], rank] for rank, p in enumerate(plan.pairs)), key=lambda p: -p[1] * p[3])
        pairs = _table(rows, dev)
        npairs, ns = len(rows), len(plan.per1)
        m1 = max(p[1] for p in rows)
        m2 = max(p[3] for p in rows)
        score = torch.empty(npairs, dtype=torch.float32, device=dev)
        best = torch.empty(ns, dtype=torch.int32, device=dev)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        traj = torch.empty(npairs * 11 * (m1 + m2), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        call("mx_emd_scores", ptr(feat), ptr(pairs), npairs, m1, m2, ptr(score), ptr(traj) if traj is not None else None, stream())
        call("mx_emd_best", ptr(score), ptr(pairs), npairs, ns, ptr(best), ptr(loss), stream())
        ctx.save_for_backward(feat, pairs, best)
        ctx.traj = traj
        ctx.dims = (ns, m1, m2)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        feat, pairs, best = ctx.saved_tensors
        ns, m1, m2 = ctx.dims
        gx = torch.zeros_like(feat)
        gup = g.contiguous().float().reshape(1)
        call("mx_emd_grad", ptr(feat), ptr(pairs), ptr(best), ns, m1, m2, ptr(ctx.traj), ptr(gup), 1.0 / ns, ptr(gx), stream())
        ctx.traj = None
        return gx, None


class EMD(object):
    """loss_multilabel.EMD, 'dynamic' mode (the only live one, train_mcl.py:221)."""

    def __call__(self, crops1, crops2, mode="static"):
        if mode != "dynamic":
            raise NotImplementedError("only mode='dynamic' is on the reference's live path (train_mcl.py:221)")
        if len(crops1) == 0:
            raise ZeroDivisionError("division by zero")      # losses / len(crops1), loss_multilabel.py:326
        return _EMD.apply(crops1.feat, crops1.plan)


# ---------------------------------------------------------------------------
# the second half of the loop body
# ---------------------------------------------------------------------------
def run(model, optimizer, batch, ep, label_with_bg, out, crop_geom=None, grad_hook=None, host_coords=None):
    model.eval()
    view1, view2 = batch["view1"], batch["view2"]
    _, sgcs_vw1 = model(view1, cam="pix")
    with torch.no_grad():
        cams_vw2, _ = model(view2, cam="pix")
    out["loss_pixpro"] = PixPro(cam_maxnorm(sgcs_vw1), cam_maxnorm(cams_vw2), batch["coord1"], batch["coord2"],
                                mask=label_with_bg)
    loss = out["loss_pixpro"]
    if ep >= 12:
        vw1 = normalize_channels(cam_softmaxnorm(sgcs_vw1))
        vw2 = normalize_channels(cam_softmaxnorm(cams_vw2))
        c1, c2, _ = get_dynamic_crops(vw1, batch["coord1"], vw2.detach(), batch["coord2"], crop_geom, host_coords)
        out["loss_emd"] = EMD()(c1, c2, mode="dynamic")
        # train_mcl.py:211 binds `loss` to the loss_pixpro tensor and :224 adds in place: the value reported as
        # loss_pixpro from epoch 12 on is pixpro + emd
        loss = loss + out["loss_emd"]
        out["loss_pixpro"] = loss
    optimizer.zero_grad()
    loss.backward()
    if grad_hook is not None:
        grad_hook(model, 2)
    optimizer.step()
