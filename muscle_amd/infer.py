"""CAM generation for one image, the loop body of infer_mcl.py:107-182, on the HIP path.

Per forward pass of the multi-scale / flip list the reference moves the [1,21,Hs,Ws] maps to the host, transposes,
cv2.resize()s them to the original image size, flips the odd passes back, drops the background channel and appends to a
list; after the list it sums, clamps, min-max normalises per channel and keeps the channels of the image's labels.  Here
the model is asked for its 1/16-resolution NHWC maps and one kernel per pass does the model's align_corners upsample,
the half-pixel resize to (H, W), the un-flip and the accumulation into a resident [20,H,W] sum; a second kernel
normalises in place.  Only the kept channels leave the GPU.

The file layout of the result is the reference's: `np.save(path, {class_index: float32[H,W]})`, read back with
`np.load(path, allow_pickle=True).item()` by src/evaluation.py:25-33 and infer_irn.py:68-73.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from ._lib import call, ptr, stream

_CPAD = 24


def infer_cam(model, img_list: List[torch.Tensor], label: torch.Tensor, H: int, W: int
              ) -> Tuple[Dict[int, np.ndarray], Dict[int, np.ndarray], torch.Tensor]:
    """img_list: [1,3,Hs,Ws] device tensors in the order of VOC12ClsDatasetMSF (scale-major, plain then flipped);
    label: [1,20].  Returns (cam_dict, sgc_dict, score[20]) exactly as infer_mcl.py builds them."""
    if label.dim() != 2 or label.shape[0] != 1:
        raise ValueError("infer_cam handles one image per call (infer_mcl.py's DataLoader has batch_size 1)")
    model.eval()
    if getattr(model.backbone, "_eval_fold", None) is None and hasattr(model, "fold_eval_bn"):
        model.fold_eval_bn()              # eval BatchNorm folded into the 1x1 weights once, not per pass (train() drops it)
    dev = img_list[0].device
    K = model.classes
    acc_cam = torch.zeros(K - 1, H, W, dtype=torch.float32, device=dev)
    acc_sgc = torch.zeros(K - 1, H, W, dtype=torch.float32, device=dev)
    scores = []
    with torch.no_grad():
        i = 0
        while i < len(img_list):
            img = img_list[i]
            if img.shape[0] != 1:
                raise ValueError("each entry of img_list is one image [1,3,Hs,Ws]")
            # a scale and its flipped copy have the same size: one batch-2 forward instead of two batch-1 forwards (eval
            # mode is per-sample: same maps; at these sizes a B7 forward is launch-bound, so the pass costs the same)
            pair = i % 2 == 0 and i + 1 < len(img_list) and img_list[i + 1].shape == img.shape
            x = torch.cat([img, img_list[i + 1]], dim=0).float() if pair else img.float()
            cam_lr, sgc_lr, _emb, score = model(x, cam="cam_lr")                 # NHWC [b,h,w,24]
            _, h, w, lds = cam_lr.shape
            Hs, Ws = img.shape[2], img.shape[3]
            for b in range(x.shape[0]):
                call("mx_infer_accum", ptr(cam_lr[b]), ptr(acc_cam), h, w, lds, K, Hs, Ws, H, W, (i + b) % 2, stream())
                call("mx_infer_accum", ptr(sgc_lr[b]), ptr(acc_sgc), h, w, lds, K, Hs, Ws, H, W, (i + b) % 2, stream())
                scores.append(score[b:b + 1, 1:])
            i += x.shape[0]
        call("mx_infer_norm", ptr(acc_cam), K - 1, H * W, stream())
        call("mx_infer_norm", ptr(acc_sgc), K - 1, H * W, stream())
        score = torch.sigmoid(torch.mean(torch.cat(scores, dim=0), dim=0))
    keep = [i for i in range(K - 1) if float(label[0, i]) > 1e-5]
    cam_dict: Dict[int, np.ndarray] = {}
    sgc_dict: Dict[int, np.ndarray] = {}
    if keep:
        idx = torch.tensor(keep, device=dev)
        cams = acc_cam.index_select(0, idx).cpu().numpy()
        sgcs = acc_sgc.index_select(0, idx).cpu().numpy()
        for j, i in enumerate(keep):
            cam_dict[i] = cams[j]
            sgc_dict[i] = sgcs[j]
    return cam_dict, sgc_dict, score


def save_cam_dict(path: str, d: Dict[int, np.ndarray]) -> None:
    """infer_mcl.py:177-178: np.save of the {class: map} dict (an object array holding the dict)."""
    np.save(path, d)


def load_cam_dict(path: str) -> Dict[int, np.ndarray]:
    """src/evaluation.py:27: np.load(...).item()"""
    return np.load(path, allow_pickle=True).item()
