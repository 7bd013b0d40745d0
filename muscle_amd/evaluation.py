"""Per-epoch rapid evaluation that drives ReduceLROnPlateau — train_mcl.py:286-318 + src/evaluation.py:10-68 — on the GPU.

The reference writes one `{class: float16[H,W]}` .npy per training image (1464 files), then for each of 16 thresholds
forks 8 processes that reload every file, argmax against the threshold, and count P / T / TP per class under locks.
Here each image's SGC goes `cam_maxnorm -> * label -> fp16 -> argmax vs all thresholds -> counts` in one kernel that
accumulates an int64 [thresholds, 21, 3] table on the device; the table is read once per epoch.  Same integers, same
`loglist` dict (per-category IoU in percent + 'mIoU').
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from ._lib import MuscleHipError, call, ptr, stream
from .phase2 import cam_maxnorm

categories = ['background', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
              'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']

RAPID_THRESHOLDS = tuple(t / 100.0 for t in range(20, 52, 2))          # train_mcl.py:308-309


class RapidEval:
    """Accumulates the (TP, P, T) table over the evaluation images of one epoch."""

    def __init__(self, device, thresholds: Sequence[float] = RAPID_THRESHOLDS, num_cls: int = 21):
        self.thresholds = tuple(float(t) for t in thresholds)
        self.num_cls = num_cls
        self.thr = torch.tensor(self.thresholds, dtype=torch.float32, device=device)
        self.counts = torch.zeros(len(self.thresholds), num_cls, 3, dtype=torch.int64, device=device)

    def add_prediction(self, pred: torch.Tensor, label_with_bg: torch.Tensor, gt: torch.Tensor) -> None:
        """pred: [K,H,W] fp32 CUDA, already cam_maxnorm'ed (train_mcl.py:297); label_with_bg: [K]; gt: uint8 [H,W]
        (255 = ignore), the SegmentationClass png."""
        if not pred.is_cuda:
            raise MuscleHipError("RapidEval runs on the HIP kernels only")
        K, H, W = pred.shape
        if gt.shape != (H, W) or gt.dtype != torch.uint8:
            raise ValueError(f"gt must be uint8 [{H},{W}] (got {gt.dtype} {tuple(gt.shape)})")
        pred = pred.contiguous().float()
        lab = label_with_bg.to(pred.device, torch.float32).contiguous().view(-1)
        g = gt.to(pred.device).contiguous()
        call("mx_eval_confusion", ptr(pred), ptr(lab), ptr(g), ptr(self.thr), len(self.thresholds), K, H, W, ptr(self.counts),
             stream())

    def add(self, model, img: torch.Tensor, label: torch.Tensor, gt: torch.Tensor) -> None:
        """One image of the eval loader (train_mcl.py:290-303): img [1,3,H,W], label [1,20], gt uint8 [H,W]."""
        if not model.training and getattr(model.backbone, "_eval_fold", None) is None and hasattr(model, "fold_eval_bn"):
            model.fold_eval_bn()          # once per evaluation sweep: model.train() drops it again (weights change)
        with torch.no_grad():
            _, pred, _, _ = model(img.float(), cam="cam")
            pred = cam_maxnorm(pred)
        lwb = torch.cat([torch.ones(1, device=pred.device), label.view(-1).to(pred.device).float()])
        self.add_prediction(pred[0], lwb, gt)

    def loglist(self, ti: int) -> Dict[str, float]:
        """do_python_eval's return value for threshold index ti (src/evaluation.py:56-68)."""
        c = self.counts[ti].cpu().numpy().astype(np.int64)
        TP, P, T = c[:, 0], c[:, 1], c[:, 2]
        iou = [TP[i] / (T[i] + P[i] - TP[i] + 1e-10) for i in range(self.num_cls)]
        out = {categories[i] if i < len(categories) else str(i): iou[i] * 100 for i in range(self.num_cls)}
        out['mIoU'] = np.mean(np.array(iou)) * 100
        return out

    def best(self):
        """(max_miou, max_t) exactly as train_mcl.py:311-312 (first maximum; max_t = index*0.02 + 0.2 there)."""
        mious: List[float] = [self.loglist(i)['mIoU'] for i in range(len(self.thresholds))]
        max_miou = max(mious)
        return max_miou, self.thresholds[mious.index(max_miou)], mious
