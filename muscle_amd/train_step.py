"""The MCL loop body — the build's counterpart of train_mcl.py:153-229 — and the
script-level helpers it uses (cam_maxnorm / cam_softmaxnorm, train_mcl.py:21-36; the ER expression, :185-188).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import loss_multilabel as L
from ._lib import call, lib, ptr, stream

_RBINS = 2048
ER_PARK_VALUES = os.environ.get("MUSCLE_ER_PARK", "1") == "1"     # fused ER: later digit passes read parked values (measured: DESIGN.md 3)
_er_vals_buf: dict = {}
_er_vals_retired: list = []


def _er_vals(dev, n):
    """Scratch for the parked |delta| values of the ER select: one persistent buffer per (device, stream) instead of an allocation
    of N*K*H*W floats per step (0.54 GB at B7 / batch 32 / 448 px; only the labelled classes' planes, ~12 %, are ever touched).
    It only grows; an outgrown buffer is kept (a captured hipGraph has its address baked in)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), stream())
    b = _er_vals_buf.get(key)
    if b is None or b.numel() < n:
        if b is not None:
            _er_vals_retired.append(b)
        b = _er_vals_buf[key] = torch.empty(n, dtype=torch.float32, device=dev)
    return b


class _SoftmaxNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        N, K, H, W = x.shape
        out = torch.empty_like(x)
        call("mx_softmaxnorm", ptr(x), None, ptr(out), N, K, H * W, 0, stream())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        N, K, H, W = x.shape
        out = torch.empty_like(x)
        call("mx_softmaxnorm", ptr(x), ptr(g.contiguous()), ptr(out), N, K, H * W, 1, stream())
        return out


def cam_softmaxnorm(cams):
    """train_mcl.py:30-36."""
    return _SoftmaxNorm.apply(cams)


class _ERLoss(torch.autograd.Function):
    """mean(topk(flatten(|softmaxnorm(cams).detach()*m - softmaxnorm(sgcs)*m|), k)) fused (train_mcl.py:175-188);
    the gradient goes to raw_sgcs only, as in the reference (cams is detached at :175)."""

    @staticmethod
    def forward(ctx, raw_cams, raw_sgcs, lwb, k):
        raw_cams, raw_sgcs, lwb = raw_cams.contiguous().float(), raw_sgcs.contiguous().float(), lwb.contiguous().float()
        N, K, H, W = raw_cams.shape
        HW = H * W
        if k > K * HW:
            raise RuntimeError(f"selected index k={k} out of range for rows of {K * HW} (torch.topk raises the same)")
        dev = raw_cams.device
        d = torch.empty(N * K * HW, dtype=torch.float32, device=dev)
        st_u = torch.zeros(3, N, dtype=torch.int32, device=dev)          # krem, prefix, cnt_eq
        sum_gt = torch.zeros(N, dtype=torch.int64, device=dev)           # 64-bit fixed point (include/muscle_hip.h)
        hcnt = torch.empty(N * _RBINS, dtype=torch.int32, device=dev)
        hsum = torch.empty(N * _RBINS, dtype=torch.int64, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        call("mx_er_fwd", ptr(raw_cams), ptr(raw_sgcs), ptr(lwb), N, K, HW, int(k), ptr(d), ptr(st_u[0]), ptr(st_u[1]),
             ptr(sum_gt), ptr(st_u[2]), ptr(hcnt), ptr(hsum), ptr(loss), stream())
        ctx.save_for_backward(raw_cams, raw_sgcs, lwb, st_u)
        ctx.k = int(k)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        raw_cams, raw_sgcs, lwb, st_u = ctx.saved_tensors
        N, K, H, W = raw_cams.shape
        out = torch.empty_like(raw_sgcs)
        gup = g.contiguous().float().reshape(1)
        call("mx_er_bwd", ptr(raw_cams), ptr(raw_sgcs), ptr(lwb), ptr(st_u[1]), ptr(st_u[0]), ptr(st_u[2]), ptr(gup),
             1.0 / (N * ctx.k), ptr(out), N, K, H * W, stream())
        return None, out, None, None


class _ERLossLowRes(torch.autograd.Function):
    """The same loss from the low-resolution NHWC maps [N,h,w,24] of MuSCLe.forward(cam='cam_lr'): upsampling,
    cam_softmaxnorm, mask, |diff| and the top-k select are recomputed per pixel inside the kernels."""

    @staticmethod
    def forward(ctx, cam_lr, sgc_lr, lwb, k, H, W):
        cam_lr, sgc_lr, lwb = cam_lr.contiguous().float(), sgc_lr.contiguous().float(), lwb.contiguous().float()
        N, h, w, L = cam_lr.shape
        K = lwb.shape[1]
        k_dev = None
        if torch.is_tensor(k):                    # device-side count (int32[1]): hipGraph replays, see muscle_amd.graph
            k_dev, k = k.contiguous(), 0
        elif k > K * H * W:
            raise RuntimeError(f"selected index k={k} out of range for rows of {K * H * W} (torch.topk raises the same)")
        dev = cam_lr.device
        st_u = torch.zeros(3, N, dtype=torch.int32, device=dev)          # krem, prefix, cnt_eq
        sum_gt = torch.zeros(N, dtype=torch.int64, device=dev)           # 64-bit fixed point (include/muscle_hip.h)
        hcnt = torch.empty(N * _RBINS, dtype=torch.int32, device=dev)
        hsum = torch.empty(N * _RBINS, dtype=torch.int64, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        # scratch for the parked values of the select: address space for every plane, only the labelled classes' planes are
        # ever written or read (~12 %)
        vals = _er_vals(dev, N * K * H * W) if ER_PARK_VALUES else None
        call("mx_er_lr_fwd", ptr(cam_lr), ptr(sgc_lr), ptr(lwb), N, h, w, L, K, H, W, int(k), ptr(k_dev), ptr(st_u[0]), ptr(st_u[1]),
             ptr(sum_gt), ptr(st_u[2]), ptr(hcnt), ptr(hsum), ptr(vals), ptr(loss), stream())
        ctx.save_for_backward(cam_lr, sgc_lr, lwb, st_u)
        ctx.dims = (int(k), H, W)
        ctx.k_dev = k_dev
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        cam_lr, sgc_lr, lwb, st_u = ctx.saved_tensors
        k, H, W = ctx.dims
        N, h, w, L = cam_lr.shape
        out = torch.empty_like(sgc_lr)
        gup = g.contiguous().float().reshape(1)
        need = lib().mx_er_lr_bwd_ws(N, h, w, L, lwb.shape[1])
        ws = torch.empty(max(need, 8), dtype=torch.uint8, device=out.device)
        call("mx_er_lr_bwd", ptr(cam_lr), ptr(sgc_lr), ptr(lwb), ptr(st_u[1]), ptr(st_u[0]), ptr(st_u[2]), ptr(gup),
             (1.0 / (N * k) if ctx.k_dev is None else 0.0), ptr(ctx.k_dev), ptr(out), N, h, w, L, lwb.shape[1], H, W,
             ws.data_ptr(), need, stream())
        return None, out, None, None, None, None


def er_loss_lowres(cam_lr, sgc_lr, label_with_bg, valid_channel, H: int, W: int):
    """valid_channel: Python int, or a device scalar (label.sum()) - then k = int(0.2 * valid_channel * H * W) of
    train_mcl.py:179 is evaluated on the device in the same double arithmetic and never read back."""
    if torch.is_tensor(valid_channel):
        k = (((0.2 * valid_channel.detach().double()) * H) * W).to(torch.int32).reshape(1)
        return _ERLossLowRes.apply(cam_lr, sgc_lr, label_with_bg, k, H, W)
    return _ERLossLowRes.apply(cam_lr, sgc_lr, label_with_bg, int(0.2 * valid_channel * H * W), H, W)


def er_loss(raw_cams, raw_sgcs, label_with_bg, valid_channel: int):
    n, c, h, w = raw_cams.shape
    return _ERLoss.apply(raw_cams, raw_sgcs, label_with_bg, int(0.2 * valid_channel * h * w))


# ---------------------------------------------------------------------------
# loop body
# ---------------------------------------------------------------------------
def mcl_step(model, optimizer, batch: Dict[str, torch.Tensor], ep: int, *, drop_u=None, crop_geom=None,
             valid_channel: Optional[int] = None, grad_hook=None, imc_sync: bool = False, fused_er: bool = True):
    """One iteration of train_mcl.py:153-229 on the HIP path: same order, same epoch gates (4/8/12), same
    loss composition, two optimizer steps once ep >= 8.

    batch: {"img" [N,3,S,S], "label" [N,20], "view1","view2" [N,3,V,V], "coord1","coord2" [N,4] int64}, CUDA.
    valid_channel: int(label.sum()) if the caller already has it on the host (train_mcl.py:178 reads it
    back from the device every iteration; passing it avoids that synchronisation), or the device scalar label.sum()
    itself (fused_er only): then nothing in phase 1 touches the host and the step can be captured (muscle_amd.graph).
    grad_hook(model, phase): called after each backward and before the optimizer step — the data-parallel
    gradient all-reduce plugs in here (muscle_amd.dist).
    imc_sync: reproduce the reference's Python-float fall-through for IMC with a device->host read
    (default: add the device-side loss, which is exactly 0 with zero gradient in that case).
    fused_er: compute the ER term from the low-resolution CAM / SGC (same numbers; the two [N,21,H,W] maps of
    MuSCLe.py:256-257 are never materialised).  False = go through the public forward(cam='cam') tensors.
    Returns the dict of the seven loss terms train_mcl.py:243-249 prints.
    """
    img, label = batch["img"], batch["label"].float()
    out: Dict[str, object] = {}
    host_coords = None
    if ep >= 12:
        from . import phase2
        host_coords = phase2.prefetch_coords(batch)     # the crop planning of phase 2 reads them on the host
    optimizer.zero_grad()
    if not model.training:             # (walking ~150 submodules costs 0.65 ms of host time per call)
        model.train()
    n = label.shape[0]
    label_with_bg = torch.cat((torch.ones((n, 1), dtype=label.dtype, device=label.device), label), dim=1)
    if fused_er:
        cam_lr, sgc_lr, emb, logits = model(img, cam="cam_lr", drop_u=drop_u)
    else:
        raw_cams, raw_sgcs, emb, logits = model(img, cam="cam", drop_u=drop_u)
    if valid_channel is None:
        valid_channel = int(label.sum().cpu())
    p = L.sigmoid(logits[:, 1:])
    out["loss_focal"] = L.FocalLoss()(p, label)
    out["loss_softmargin"] = L.MultiLabelSoftMarginLoss()(logits[:, 1:], label)
    out["loss_pair"] = L.Log_Sum_Exp_Pairwise_Loss(p, label).mean()
    loss_cls = out["loss_pair"] + out["loss_softmargin"] + out["loss_focal"]
    if fused_er:
        out["loss_er"] = er_loss_lowres(cam_lr.detach(), sgc_lr, label_with_bg, valid_channel, img.shape[2], img.shape[3])
    else:
        out["loss_er"] = er_loss(raw_cams, raw_sgcs, label_with_bg, valid_channel)
    loss = loss_cls + out["loss_er"]
    out["loss_imc"] = 0
    if ep >= 4:
        if imc_sync:
            out["loss_imc"] = L.image_level_contrast(emb, label)
            if torch.is_tensor(out["loss_imc"]):
                loss = loss_cls + out["loss_imc"] + out["loss_er"]
        else:
            out["loss_imc"], _ = L.image_level_contrast_nosync(emb, label)
            loss = loss_cls + out["loss_imc"] + out["loss_er"]
    optimizer.zero_grad()
    loss.backward()
    if grad_hook is not None:
        grad_hook(model, 1)
    optimizer.step()
    out["loss_pixpro"] = 0
    out["loss_emd"] = 0
    if ep >= 8:
        from . import phase2
        phase2.run(model, optimizer, batch, ep, label_with_bg, out, crop_geom=crop_geom, grad_hook=grad_hook, host_coords=host_coords)
    return out


def muscle_step(model, optimizer, batch: Dict[str, torch.Tensor], *, lamb: float = 0.05, step: int = 7, k: int = 128,
                criterion2=None, drop_u=None, grad_hook=None, fused: bool = True):
    """One iteration of train_muscle.py:171-203 on the HIP path: seg forward of MuSCLe(mode='dec'), cross entropy against the
    arg-max of the soft pseudo-label, lamb * BEACON FieldLoss, backward, clip_grad_norm_(9), optimizer step.
    batch: {"img" [N,3,S,S], "label" [N,20], "mask" [N,21,S,S]} on the GPU.
    fused: the FieldLoss samples the decoder's 1/8-resolution features at the chosen points (the [N,256,S,S] dense_ft of
    MuSCLe.py:285 is never materialised); False goes through the public forward(cam='seg') tensors."""
    from . import edge
    img, label, mask = batch["img"], batch["label"].float(), batch["mask"]
    optimizer.zero_grad()
    if not model.training:             # (walking ~150 submodules costs 0.65 ms of host time per call)
        model.train()
    n = label.shape[0]
    label_with_bg = torch.cat((torch.ones((n, 1), dtype=label.dtype, device=label.device), label), dim=1)
    seg_map, ft = model(img, cam="seg_p3" if fused else "seg", drop_u=drop_u)
    l1 = edge.cross_entropy_argmax(seg_map, mask)
    loss, l2 = l1, 0
    if lamb > 0:
        crit = criterion2 if criterion2 is not None else edge.FieldLoss(sobel_size=5, beta=1e2, k=k)
        l2, _ = crit(seg_map, ft, mask, label_with_bg, step, dense_is_lowres_nhwc=fused)
        if torch.is_tensor(l2):
            loss = l1 + lamb * l2
    optimizer.zero_grad()
    loss.backward()
    if grad_hook is not None:
        grad_hook(model, 1)
    total = edge.clip_grad_norm_(model, 9)
    optimizer.step()
    return {"loss_seg": l1, "loss_beacon": l2, "grad_norm": total}
