"""Input path of the MCL loop body: the two-view sampler and the tuple layout of the reference's `VOC12ClsPix`
(src/data.py:215-332) with the per-pixel work moved to the GPU.

What the reference does per image in a DataLoader worker (src/data.py:306-315, train_mcl.py:104-115):
    PIL open -> random hflip -> get_views (two overlapping 224x224 crops + overlap coordinates)
    img   = HWC_to_CHW(RandomCrop(448)(color_norm(np.asarray(ColorJitter(RandomResizeLong(448,768)(img))))))   [+ RandomErasing]
    viewK = HWC_to_CHW(color_norm(np.asarray(ColorJitter(viewK))))
and what the loop body then does with the batch (train_mcl.py:161-165): `.cuda().float()`.

Here the host keeps what is geometry and PIL (decode, flip, the bicubic RandomResizeLong, the crops: same PIL calls,
same random-number draws in the same order from the same generators: `torch` for flip / views, Python's `random` for the
resize target and the crop box) and ships uint8 crops through one pinned buffer; `mx_input_stage` does color_norm,
RandomCrop's zero container, HWC -> CHW and the fp32 cast on the device, bit-exact with the numpy expressions.
Per 448x448 image that is 0.9 MB of uint8 over PCIe instead of 2.4 MB fp32 + 2 x 1.2 MB fp64, and no fp64 numpy
passes on the host cores (at 8 x 219 img/s the node needs ~1 750 img/s from them).

NOT reproduced (torchvision is not installed where the fixtures are generated, so nothing could pin them):
`ColorJitter` (image and views) and `RandomErasing`.  With them absent the draws they would consume from torch's
generator are absent too; everything else follows the reference's draw order.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import call, ptr, stream


# ---- geometry (pure host logic) ------------------------------------------------------------------------------------
def get_inter(coord1, coord2):
    """Overlap of two crop boxes (h0, w0, hl, wl) -> (rel1, rel2, ori) or (False, False, False) (src/data.py:233-270).
    rel* = (h0, w0, h_inter, w_inter) of the overlap inside each view; ori = (x_left, y_top, h_inter, w_inter)."""
    h11, w11, h12, w12 = coord1[0], coord1[1], coord1[0] + coord1[2], coord1[1] + coord1[3]
    h21, w21, h22, w22 = coord2[0], coord2[1], coord2[0] + coord2[2], coord2[1] + coord2[3]
    y_top, x_left = max(h11, h21), max(w11, w21)
    y_bot, x_right = min(h12, h22), min(w12, w22)
    if y_bot - y_top <= 0 or x_right - x_left <= 0:
        return False, False, False
    h_inter, w_inter = y_bot - y_top, x_right - x_left
    # the reference walks four corner cases (:247-267); they reduce to "offset of the overlap inside each view"
    rel1 = (y_top - h11, x_left - w11, h_inter, w_inter)
    rel2 = (y_top - h21, x_left - w21, h_inter, w_inter)
    return rel1, rel2, (x_left, y_top, h_inter, w_inter)


def sample_views(h: int, w: int, out_size=(224, 224)):
    """The draws of get_views (src/data.py:273-304): four torch.randint per attempt, redrawn until the views overlap."""
    th, tw = out_size
    if h + 1 < th or w + 1 < tw:
        raise ValueError("Required crop size {} is larger then input image size {}".format((th, tw), (h, w)))
    while True:
        i_1 = torch.randint(0, h - th + 1, size=(1,)).item()
        j_1 = torch.randint(0, w - tw + 1, size=(1,)).item()
        i_2 = torch.randint(0, h - th + 1, size=(1,)).item()
        j_2 = torch.randint(0, w - tw + 1, size=(1,)).item()
        rel1, rel2, ori = get_inter((i_1, j_1, th, tw), (i_2, j_2, th, tw))
        if rel1 is not False:
            return (i_1, j_1), (i_2, j_2), rel1, rel2, ori


def resize_long_target(w: int, h: int, min_long=448, max_long=768):
    """RandomResizeLong's draw and target (w, h) (src/imutils.py:127-141)."""
    target_long = random.randint(min_long, max_long)
    if w < h:
        return (int(round(w * target_long / h)), target_long)
    return (target_long, int(round(h * target_long / w)))


def random_crop_box(h: int, w: int, cropsize: int):
    """RandomCrop's draws (src/imutils.py:143-172; w first, then h) -> (cont_top, cont_left, img_top, img_left, ch, cw)."""
    ch, cw = min(cropsize, h), min(cropsize, w)
    w_space, h_space = w - cropsize, h - cropsize
    if w_space > 0:
        cont_left, img_left = 0, random.randrange(w_space + 1)
    else:
        cont_left, img_left = random.randrange(-w_space + 1), 0
    if h_space > 0:
        cont_top, img_top = 0, random.randrange(h_space + 1)
    else:
        cont_top, img_top = random.randrange(-h_space + 1), 0
    return cont_top, cont_left, img_top, img_left, ch, cw


class ItemPlan:
    """uint8 crops + placements of one training item, ready for the device stage."""
    __slots__ = ("img_u8", "img_place", "view1_u8", "view2_u8", "coord1", "coord2", "ori_coord")


def plan_item(pil_img, crop_size: int = 448, view_size=(224, 224), resize_long=(448, 768)) -> ItemPlan:
    """Host side of VOC12ImageViews.__getitem__ (src/data.py:306-315) + the train transform (train_mcl.py:104-115) for
    one decoded RGB PIL image, in the reference's draw order."""
    from PIL import Image
    p = ItemPlan()
    if torch.rand(1) < 0.5:                                     # :309-310
        pil_img = pil_img.transpose(Image.FLIP_LEFT_RIGHT)
    w, h = pil_img.size
    views_src = pil_img
    if w < 448 or h < 448:                                      # :274-276 (F.resize of a PIL image: bilinear)
        views_src = pil_img.resize((448, 448), Image.BILINEAR)
    w2, h2 = views_src.size
    (i1, j1), (i2, j2), rel1, rel2, ori = sample_views(h2, w2, view_size)
    th, tw = view_size
    p.view1_u8 = np.ascontiguousarray(np.asarray(views_src.crop((j1, i1, j1 + tw, i1 + th))))
    p.view2_u8 = np.ascontiguousarray(np.asarray(views_src.crop((j2, i2, j2 + tw, i2 + th))))
    p.coord1, p.coord2, p.ori_coord = rel1, rel2, ori
    # transform(img): RandomResizeLong (bicubic, PIL) -> [ColorJitter: not reproduced] -> color_norm -> RandomCrop -> CHW
    big = pil_img.resize(resize_long_target(w, h, *resize_long), resample=Image.BICUBIC)
    arr = np.asarray(big)
    ct, cl, it, il, ch, cw = random_crop_box(arr.shape[0], arr.shape[1], crop_size)
    p.img_u8 = np.ascontiguousarray(arr[it:it + ch, il:il + cw])
    p.img_place = (ct, cl)
    return p


# ---- device stage -----------------------------------------------------------------------------------------------------
class InputStager:
    """Packs the uint8 crops of a batch into one pinned buffer, copies it once and runs `mx_input_stage` three times
    (img, view1, view2).  Two pinned buffers alternate so that packing batch t+1 does not wait for the copy of batch t."""

    def __init__(self, device, batch: int, crop_size: int = 448, view_size=(224, 224)):
        self.dev, self.n, self.crop, self.view = device, batch, crop_size, view_size
        cap = batch * (crop_size * crop_size + 2 * view_size[0] * view_size[1]) * 3
        self._pin = [torch.empty(cap, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else torch.empty(cap, dtype=torch.uint8)
                     for _ in range(2)]
        self._jobs_pin = [torch.empty(3 * batch * 8, dtype=torch.int32).pin_memory() if torch.cuda.is_available()
                          else torch.empty(3 * batch * 8, dtype=torch.int32) for _ in range(2)]
        self._dev_u8 = torch.empty(cap, dtype=torch.uint8, device=device)
        self._dev_jobs = torch.empty(3 * batch * 8, dtype=torch.int32, device=device)
        self._flip = 0
        self._evt = [None, None]

    def __call__(self, plans: Sequence[ItemPlan], labels: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        n = len(plans)
        assert n <= self.n
        k = self._flip
        self._flip ^= 1
        if self._evt[k] is not None:
            self._evt[k].synchronize()                      # the copy out of this pinned buffer two batches ago is done
        buf, jobs = self._pin[k].numpy(), self._jobs_pin[k].numpy().reshape(3 * self.n, 8)
        off = 0
        jobs[:] = 0
        for kind, (get, place) in enumerate(((lambda p: p.img_u8, lambda p: p.img_place),
                                             (lambda p: p.view1_u8, lambda p: (0, 0)),
                                             (lambda p: p.view2_u8, lambda p: (0, 0)))):
            for i, p in enumerate(plans):
                a = get(p)
                sz = a.size
                buf[off:off + sz] = a.reshape(-1)
                top, left = place(p)
                jobs[kind * self.n + i, :5] = (off, a.shape[0], a.shape[1], top, left)
                off += sz
        self._dev_u8[:off].copy_(self._pin[k][:off], non_blocking=True)
        self._dev_jobs.copy_(self._jobs_pin[k], non_blocking=True)
        evt = torch.cuda.Event()
        evt.record()
        self._evt[k] = evt
        img = torch.empty(n, 3, self.crop, self.crop, dtype=torch.float32, device=self.dev)
        v1 = torch.empty(n, 3, self.view[0], self.view[1], dtype=torch.float32, device=self.dev)
        v2 = torch.empty_like(v1)
        for kind, dst in enumerate((img, v1, v2)):
            call("mx_input_stage", ptr(self._dev_u8), self._dev_jobs.data_ptr() + 4 * 8 * kind * self.n, ptr(dst), n, dst.shape[2],
                 dst.shape[3], stream())
        out = {"img": img, "view1": v1, "view2": v2,
               "coord1": torch.tensor([p.coord1 for p in plans], dtype=torch.int64, device=self.dev),
               "coord2": torch.tensor([p.coord2 for p in plans], dtype=torch.int64, device=self.dev),
               "ori_coord": torch.tensor([p.ori_coord for p in plans], dtype=torch.int64, device=self.dev)}
        if labels is not None:
            out["label"] = labels.to(self.dev, non_blocking=True)
        return out


class VOC12ClsPix:
    """Drop-in for the reference dataset's role (src/data.py:317-332): `plan(idx)` is the host half of `__getitem__`
    (decode + geometry + uint8 crops); a batch of plans goes through an `InputStager` to become the loop body's
    `(img, label, view1, view2, coord1, coord2, ori_coord)` tensors on the GPU."""

    def __init__(self, img_name_list_path: str, voc12_root: str, labels: Optional[Dict[str, np.ndarray]] = None,
                 crop_size: int = 448, view_size=(224, 224)):
        import os
        self.names = [ln.split(" ")[0].split("/")[-1].split(".")[0] for ln in open(img_name_list_path).read().splitlines()]
        self.root, self.crop, self.view = voc12_root, crop_size, view_size
        if labels is None and os.path.exists("data/cls_labels.npy"):
            labels = np.load("data/cls_labels.npy", allow_pickle=True).item()          # src/data.py:54-57
        self.labels = labels

    def __len__(self):
        return len(self.names)

    def plan(self, idx: int) -> Tuple[str, ItemPlan, Optional[np.ndarray]]:
        import os
        import PIL.Image
        name = self.names[idx]
        img = PIL.Image.open(os.path.join(self.root, "JPEGImages", name + ".jpg")).convert("RGB")
        lab = None if self.labels is None else np.asarray(self.labels[name], dtype=np.float32)
        return name, plan_item(img, self.crop, self.view), lab
