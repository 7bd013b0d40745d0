"""Input path of the MCL loop body: the two-view sampler and the tuple layout of the reference's `VOC12ClsPix`
(src/data.py:215-332) with the per-pixel work moved to the GPU.

What the reference does per image in a DataLoader worker (src/data.py:306-315, train_mcl.py:104-115):
    PIL open -> random hflip -> get_views (two overlapping 224x224 crops + overlap coordinates)
    img   = HWC_to_CHW(RandomCrop(448)(color_norm(np.asarray(ColorJitter(RandomResizeLong(448,768)(img))))))   [+ RandomErasing]
    viewK = HWC_to_CHW(color_norm(np.asarray(ColorJitter(viewK))))
and what the loop body then does with the batch (train_mcl.py:161-165): `.cuda().float()`.

Here the host keeps the decode, the flip, the geometry (same random-number draws in the same order from the same
generators: `torch` for flip / views / jitter / erasing, Python's `random` for the resize target and the crop box) and the
view crops, and ships uint8 images through one pinned buffer; the device does the pixel work: `mx_resample` the bicubic
RandomResizeLong (Pillow's fixed-point resample, bit-exact), `mx_color_jitter` the ColorJitter (Pillow's blend / HSV
arithmetic, bit-exact), `mx_input_stage` color_norm, RandomCrop's zero container, RandomErasing's box, HWC -> CHW and the
fp32 cast, bit-exact with the numpy expressions.  (`plan_item(device_resize=False, device_jitter=False)` does the PIL
calls on the host instead: that is the comparator of the tests.)
Per 448x448 image that is 0.9 MB of uint8 over PCIe instead of 2.4 MB fp32 + 2 x 1.2 MB fp64, and no fp64 numpy
passes on the host cores (at 8 x 219 img/s the node needs ~1 750 img/s from them).

`ColorJitter` (image and both views, train_mcl.py:108, src/data.py:223) and `RandomErasing` (train_mcl.py:114) live in
torchvision, which the reference pins at 0.9.0 (environment.yaml:226) and which is NOT installed in the build container:
they are restated here from torchvision 0.9.0's published source (transforms.py `ColorJitter.get_params/forward`,
`RandomErasing.get_params/forward`; functional_pil.py `adjust_*`) - same draws from torch's generator in the same order,
same PIL calls - and are **parity-unpinned** as far as torchvision's glue goes (which draws, in which order): no fixture
could be generated for them.  The pixel arithmetic is Pillow's, and `mx_color_jitter` (device_jitter=True) is tested
bit-exact against Pillow itself.  `plan_item(..., augment=False)`
switches both off (the bit-exact fixtures of tests/test_input_path.py were made with identity stand-ins).
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


# ---- torchvision 0.9.0 ColorJitter / RandomErasing, restated (parity unpinned, see the module docstring) ---------------
def color_jitter_params(brightness=0.2, contrast=0.2, saturation=0.2, hue=0.1):
    """ColorJitter.get_params: a permutation of the four adjustments, then one factor each, all from torch's generator.
    Ranges as ColorJitter._check_input builds them: [max(0, 1 - v), 1 + v] and [-hue, hue]."""
    order = torch.randperm(4).tolist()
    draw = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi))  # noqa: E731
    b = draw(max(0.0, 1.0 - brightness), 1.0 + brightness) if brightness else None
    c = draw(max(0.0, 1.0 - contrast), 1.0 + contrast) if contrast else None
    s = draw(max(0.0, 1.0 - saturation), 1.0 + saturation) if saturation else None
    h = draw(-hue, hue) if hue else None
    return order, b, c, s, h


def apply_color_jitter(pil_img, params):
    """ColorJitter.forward on a PIL image: functional_pil.adjust_brightness / _contrast / _saturation are PIL's ImageEnhance,
    adjust_hue shifts the H channel of the HSV image with uint8 wrap-around."""
    from PIL import Image, ImageEnhance
    order, b, c, s, h = params
    for fn in order:
        if fn == 0 and b is not None:
            pil_img = ImageEnhance.Brightness(pil_img).enhance(b)
        elif fn == 1 and c is not None:
            pil_img = ImageEnhance.Contrast(pil_img).enhance(c)
        elif fn == 2 and s is not None:
            pil_img = ImageEnhance.Color(pil_img).enhance(s)
        elif fn == 3 and h is not None:
            if not -0.5 <= h <= 0.5:
                raise ValueError("hue_factor ({}) is not in [-0.5, 0.5].".format(h))
            mode = pil_img.mode
            if mode in {"L", "1", "I", "F"}:
                continue
            hh, ss, vv = pil_img.convert("HSV").split()
            np_h = np.array(hh, dtype=np.uint8)
            # torchvision: `np_h += np.uint8(hue_factor * 255)` under numpy 1.19, where the float -> uint8 conversion
            # truncates toward zero and wraps (numpy 2 raises instead); the addition itself wraps in uint8
            with np.errstate(over="ignore"):
                np_h += np.uint8(int(h * 255) & 0xFF)
            pil_img = Image.merge("HSV", (Image.fromarray(np_h, "L"), ss, vv)).convert(mode)
    return pil_img


def random_erasing_params(img_h: int, img_w: int, p=0.5, scale=(0.02, 0.2), ratio=(0.3, 3.3)):
    """RandomErasing.forward + get_params for value=0: torch.rand(1) < p, then up to ten attempts of (area, aspect) ->
    (h, w) with the box position drawn by two torch.randint.  Returns (i, j, h, w) or None (not applied / no box fits:
    torchvision then 'erases' with the image itself)."""
    import math
    if not (torch.rand(1) < p):
        return None
    area = img_h * img_w
    for _ in range(10):
        erase_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect_ratio = torch.empty(1).uniform_(ratio[0], ratio[1]).item()
        h = int(round(math.sqrt(erase_area * aspect_ratio)))
        w = int(round(math.sqrt(erase_area / aspect_ratio)))
        if not (h < img_h and w < img_w):
            continue
        i = torch.randint(0, img_h - h + 1, size=(1,)).item()
        j = torch.randint(0, img_w - w + 1, size=(1,)).item()
        return i, j, h, w
    return None


# ---- Pillow's resample coefficient tables (Resample.c precompute_coeffs + normalize_coeffs_8bpc), for mx_resample ------
def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def _bilinear(x):
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


_FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


def _resample_axis(in_size: int, out_size: int, filt: str):
    """(bounds [out, 2] int32, coefficients [out, ksize] int32 in 22-bit fixed point) of one resample pass: the same double
    arithmetic in the same order as Pillow's C (the weights of a window are summed left to right)."""
    import math
    f, support0 = _FILTERS[filt]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = 0.0 + (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    j = np.arange(ksize, dtype=np.int64)
    w = f(((j[None, :] + xmin[:, None]) - center[:, None] + 0.5) * ss)
    w = np.where(j[None, :] < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for c in range(ksize):                                   # sequential, as the C loop adds them
        ww = ww + w[:, c]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    ki = np.where(k < 0, (-0.5 + k * (1 << 22)).astype(np.int64), (0.5 + k * (1 << 22)).astype(np.int64)).astype(np.int32)
    return np.stack([xmin, xmax], 1).astype(np.int32), ki


def resample_tables(win: int, hin: int, wout: int, hout: int, filt: str = "bicubic") -> np.ndarray:
    """The int32 table block mx_resample reads for one image: ksize_h, ksize_v, bounds_h, kk_h, bounds_v, kk_v."""
    bh, kh = _resample_axis(win, wout, filt)
    bv, kv = _resample_axis(hin, hout, filt)
    return np.concatenate([np.array([kh.shape[1], kv.shape[1]], dtype=np.int32), bh.reshape(-1), kh.reshape(-1), bv.reshape(-1), kv.reshape(-1)])


class ItemPlan:
    """uint8 crops + placements of one training item, ready for the device stage."""
    __slots__ = ("img_u8", "img_place", "view1_u8", "view2_u8", "coord1", "coord2", "ori_coord", "erase", "jitter", "img_crop",
                 "resize_to", "tables")
    # resize_to: None, or (W, H) of RandomResizeLong still TO BE done - on the device (mx_resample with `tables`); then img_u8 is
    # the ORIGINAL image and img_crop the RandomCrop window inside the resized one
    # jitter: None, or the three ColorJitter parameter sets (image, view 1, view 2) still TO BE applied - on the device;
    # then img_u8 is the whole resized image and img_crop = (top, left, h, w) the RandomCrop window inside it


def plan_item(pil_img, crop_size: int = 448, view_size=(224, 224), resize_long=(448, 768), augment: bool = True,
              device_jitter: bool = False, device_resize: bool = False) -> ItemPlan:
    """Host side of VOC12ImageViews.__getitem__ (src/data.py:306-315) + the train transform (train_mcl.py:104-115) for
    one decoded RGB PIL image, in the reference's draw order: flip, views, [img: resize, jitter, crop, erasing], view1
    jitter, view2 jitter.  augment=False leaves ColorJitter and RandomErasing out (and their draws with them).
    device_jitter: draw the ColorJitter parameters here but leave the pixel work to `mx_color_jitter` (bit-exact with the
    PIL calls, 4x less host time per item): the plan then carries the whole resized image.
    device_resize: leave RandomResizeLong's bicubic resize to `mx_resample` as well (bit-exact with PIL.Image.resize): the
    plan carries the ORIGINAL image and Pillow's coefficient tables."""
    from PIL import Image
    p = ItemPlan()
    p.erase = p.jitter = p.img_crop = p.resize_to = p.tables = None
    if torch.rand(1) < 0.5:                                     # :309-310
        pil_img = pil_img.transpose(Image.FLIP_LEFT_RIGHT)
    w, h = pil_img.size
    views_src = pil_img
    if w < 448 or h < 448:                                      # :274-276 (F.resize of a PIL image: bilinear)
        views_src = pil_img.resize((448, 448), Image.BILINEAR)
    w2, h2 = views_src.size
    (i1, j1), (i2, j2), rel1, rel2, ori = sample_views(h2, w2, view_size)
    th, tw = view_size
    view1 = views_src.crop((j1, i1, j1 + tw, i1 + th))
    view2 = views_src.crop((j2, i2, j2 + tw, i2 + th))
    p.coord1, p.coord2, p.ori_coord = rel1, rel2, ori
    # transform(img): RandomResizeLong (bicubic, PIL) -> ColorJitter -> color_norm -> RandomCrop -> CHW -> RandomErasing
    target = resize_long_target(w, h, *resize_long)
    jit = []
    if device_resize:
        if augment and not device_jitter:
            raise ValueError("device_resize needs the ColorJitter on the device too (the resized image never exists on the host)")
        if augment:
            jit.append(color_jitter_params())
        ct, cl, it, il, ch, cw = random_crop_box(target[1], target[0], crop_size)
        p.img_u8, p.img_crop = np.ascontiguousarray(np.asarray(pil_img)), (it, il, ch, cw)
        p.resize_to, p.tables = target, resample_tables(w, h, target[0], target[1], "bicubic")
    else:
        big = pil_img.resize(target, resample=Image.BICUBIC)
        if augment:
            jit.append(color_jitter_params())
            if not device_jitter:
                big = apply_color_jitter(big, jit[0])
        arr = np.asarray(big)
        ct, cl, it, il, ch, cw = random_crop_box(arr.shape[0], arr.shape[1], crop_size)
        if augment and device_jitter:
            p.img_u8, p.img_crop = np.ascontiguousarray(arr), (it, il, ch, cw)   # contrast needs the whole image's mean
        else:
            p.img_u8 = np.ascontiguousarray(arr[it:it + ch, il:il + cw])
    p.img_place = (ct, cl)
    if augment:
        p.erase = random_erasing_params(crop_size, crop_size)          # on the cropped [3, crop, crop] tensor
        jit += [color_jitter_params(), color_jitter_params()]          # view_transform, src/data.py:222-228
        if device_jitter:
            p.jitter = jit
        else:
            view1, view2 = apply_color_jitter(view1, jit[1]), apply_color_jitter(view2, jit[2])
    p.view1_u8 = np.ascontiguousarray(np.asarray(view1))
    p.view2_u8 = np.ascontiguousarray(np.asarray(view2))
    return p


# ---- device stage -----------------------------------------------------------------------------------------------------
class InputStager:
    """Packs the uint8 images of a batch into one pinned buffer, copies it once and runs the device half:
    [mx_resample: RandomResizeLong] -> [mx_color_jitter: ColorJitter of image and views] -> mx_input_stage x 3 (color_norm,
    RandomCrop container, RandomErasing box, CHW, fp32).  Which of the bracketed steps run is decided by what the plans
    carry (`plan_item(device_resize=..., device_jitter=...)`).  Two pinned buffer sets alternate so that packing batch t+1
    does not wait for the copy of batch t."""

    def __init__(self, device, batch: int, crop_size: int = 448, view_size=(224, 224), max_long: int = 768, max_src: int = 1024):
        self.dev, self.n, self.crop, self.view = device, batch, crop_size, view_size
        side = max(crop_size, max_long)
        cap = batch * (max(side, max_src) ** 2 + 2 * view_size[0] * view_size[1]) * 3     # whole source / resized images + views
        pin = (lambda t: t.pin_memory()) if torch.cuda.is_available() else (lambda t: t)
        words = 3 * batch * 8
        tab_words = batch * (2 + 2 * side * (2 + 13))       # per image: two axes, <= `side` outputs, bounds + <= 13 taps
        self._pin = [pin(torch.empty(cap, dtype=torch.uint8)) for _ in range(2)]
        self._jobs_pin = [pin(torch.empty(words, dtype=torch.int32)) for _ in range(2)]
        self._jit_pin = [pin(torch.empty(words, dtype=torch.int32)) for _ in range(2)]
        self._rs_pin = [pin(torch.empty(batch * 8, dtype=torch.int32)) for _ in range(2)]
        self._tab_pin = [pin(torch.empty(tab_words, dtype=torch.int32)) for _ in range(2)]
        self._dev_u8 = torch.empty(cap, dtype=torch.uint8, device=device)
        self._dev_jobs = torch.empty(words, dtype=torch.int32, device=device)
        self._dev_jit = torch.empty(words, dtype=torch.int32, device=device)
        self._dev_sums = torch.empty(3 * batch, dtype=torch.int64, device=device)
        self._dev_rs_jobs = torch.empty(batch * 8, dtype=torch.int32, device=device)
        self._dev_tab = torch.empty(tab_words, dtype=torch.int32, device=device)
        self._dev_rs = self._dev_tmp = None                 # resized images / horizontal-pass images: allocated on first use
        self._side, self._max_src = side, max_src
        self._flip = 0
        self._evt = [None, None]

    @staticmethod
    def _jitter_words(params):
        order, fb, fc, fs, fh = params
        code = 0
        for pos in range(4):
            fn = order[pos]
            code |= (fn if (fb, fc, fs, fh)[fn] is not None else 15) << (4 * pos)
        return code, (fb or 0.0, fc or 0.0, fs or 0.0), ((int(fh * 255) & 0xFF) if fh is not None else 0)

    def __call__(self, plans: Sequence[ItemPlan], labels: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        n = len(plans)
        assert n <= self.n
        k = self._flip
        self._flip ^= 1
        if self._evt[k] is not None:
            self._evt[k].synchronize()                      # the copies out of this pinned set two batches ago are done
        buf, jobs = self._pin[k].numpy(), self._jobs_pin[k].numpy().reshape(3 * self.n, 8)
        jit = self._jit_pin[k].numpy().reshape(3 * self.n, 8)
        jit_f = jit.view(np.float32)
        rsj, tab = self._rs_pin[k].numpy().reshape(self.n, 8), self._tab_pin[k].numpy()
        resize = getattr(plans[0], "resize_to", None) is not None
        assert all((getattr(p, "resize_to", None) is not None) == resize for p in plans), "one resize mode per batch"
        jobs[:] = 0
        jit[:] = 0
        jit[:, 3] = 0xFFFF                                  # order nibbles: nothing to do
        off = tab_off = tmp_off = rs_off = 0
        jit_px = [1, 1]                                      # largest image among the jitter jobs of (image, views)
        any_jit = False
        for kind, (get, place) in enumerate(((lambda p: p.img_u8, lambda p: p.img_place),
                                             (lambda p: p.view1_u8, lambda p: (0, 0)),
                                             (lambda p: p.view2_u8, lambda p: (0, 0)))):
            for i, p in enumerate(plans):
                a = get(p)
                sz = a.size
                buf[off:off + sz] = a.reshape(-1)
                top, left = place(p)
                row = kind * self.n + i
                h, w, base = a.shape[0], a.shape[1], off    # the image the jitter / the stage will see, and where it starts
                if kind == 0 and resize:                     # RandomResizeLong on the device: source -> tmp -> resized
                    tw, th = p.resize_to
                    if max(th, tw) > self._side or max(h, w) > self._max_src:
                        raise ValueError(f"image {w}x{h} -> {tw}x{th} exceeds the stager's max_src / max_long")
                    t = p.tables
                    if tab_off + t.size > tab.size:
                        raise ValueError("resample tables exceed the stager's table buffer (a very strong downscale): raise max_src / max_long")
                    tab[tab_off:tab_off + t.size] = t
                    rsj[i] = (off, h, w, tmp_off, rs_off, tw, th, tab_off)
                    tab_off += t.size
                    tmp_off += h * tw * 3
                    h, w, base = th, tw, rs_off
                    rs_off += th * tw * 3
                jobs[row, :5] = (base, h, w, top, left)
                pj = getattr(p, "jitter", None)
                if pj is not None:                                   # ColorJitter still to be applied: on the device
                    code, facs, hue = self._jitter_words(pj[kind])
                    jit[row, :4] = (base, h, w, code)
                    jit_f[row, 4:7] = facs
                    jit[row, 7] = hue
                    any_jit = True
                    jit_px[min(kind, 1)] = max(jit_px[min(kind, 1)], h * w)
                if kind == 0 and getattr(p, "img_crop", None) is not None:   # the stage reads the crop window inside the whole image
                    it, il, ch, cw = p.img_crop
                    jobs[row, :3] = (base + (it * w + il) * 3, ch, cw)
                    jobs[row, 7] = w
                er = getattr(p, "erase", None) if kind == 0 else None
                if er is not None:                                   # RandomErasing box of the image, output coordinates
                    jobs[row, 5] = er[0] | (er[1] << 16)
                    jobs[row, 6] = er[2] | (er[3] << 16)
                off += sz
        self._dev_u8[:off].copy_(self._pin[k][:off], non_blocking=True)
        self._dev_jobs.copy_(self._jobs_pin[k], non_blocking=True)
        if any_jit:
            self._dev_jit.copy_(self._jit_pin[k], non_blocking=True)
        if resize:
            self._dev_rs_jobs.copy_(self._rs_pin[k], non_blocking=True)
            self._dev_tab[:tab_off].copy_(self._tab_pin[k][:tab_off], non_blocking=True)
        evt = torch.cuda.Event()
        evt.record()
        self._evt[k] = evt
        img_src = self._dev_u8
        if resize:
            if self._dev_rs is None:
                self._dev_rs = torch.empty(self.n * self._side * self._side * 3, dtype=torch.uint8, device=self.dev)
                self._dev_tmp = torch.empty(self.n * self._max_src * self._side * 3, dtype=torch.uint8, device=self.dev)
            call("mx_resample", ptr(self._dev_u8), ptr(self._dev_rs_jobs), ptr(self._dev_tab), ptr(self._dev_tmp), ptr(self._dev_rs), n,
                 self._max_src * self._side, stream())
            img_src = self._dev_rs
        if any_jit:
            call("mx_color_jitter", ptr(img_src), ptr(self._dev_jit), ptr(self._dev_sums), n, int(jit_px[0]), stream())
            call("mx_color_jitter", ptr(self._dev_u8), self._dev_jit.data_ptr() + 4 * 8 * self.n, self._dev_sums.data_ptr() + 8 * self.n,
                 2 * self.n, int(jit_px[1]), stream())
        img = torch.empty(n, 3, self.crop, self.crop, dtype=torch.float32, device=self.dev)
        v1 = torch.empty(n, 3, self.view[0], self.view[1], dtype=torch.float32, device=self.dev)
        v2 = torch.empty_like(v1)
        for kind, dst in enumerate((img, v1, v2)):
            call("mx_input_stage", ptr(img_src if kind == 0 else self._dev_u8), self._dev_jobs.data_ptr() + 4 * 8 * kind * self.n, ptr(dst), n,
                 dst.shape[2], dst.shape[3], stream())
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
                 crop_size: int = 448, view_size=(224, 224), augment: bool = True, device_jitter: bool = True,
                 device_resize: bool = True):
        import os
        self.names = [ln.split(" ")[0].split("/")[-1].split(".")[0] for ln in open(img_name_list_path).read().splitlines()]
        self.root, self.crop, self.view, self.augment, self.device_jitter = voc12_root, crop_size, view_size, augment, device_jitter
        self.device_resize = device_resize and (device_jitter or not augment)
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
        return name, plan_item(img, self.crop, self.view, augment=self.augment, device_jitter=self.device_jitter,
                               device_resize=self.device_resize), lab

    __getitem__ = plan          # a torch.utils.data map-style dataset: DataLoader workers run the host half


def _keep(items):               # collate_fn: the plans stay a list (ragged uint8 crops); module-level so workers can pickle it
    return items


def _reference_worker_init(worker_id):
    np.random.seed(1 + worker_id)                       # train_mcl.py:127-128


class StagedLoader:
    """The reference's `DataLoader(train_dataset, batch_size, num_workers, pin_memory=True, drop_last=True,
    worker_init_fn, shuffle=True)` (train_mcl.py:130-132) for this input path: torch's DataLoader runs `VOC12ClsPix.plan`
    in its worker processes (decode, flip, views, PIL resize, jitter, crop: ~10 ms of one core per image, so 245 img/s per
    GPU want ~3 cores per GPU) with the SAME per-worker seeding of torch / random as in the reference (base_seed + worker
    id), and each batch of plans goes through the `InputStager` in the training process: one pinned copy + three launches.
    Iterating yields `(names, batch)`, `batch` being the dict `mcl_step` takes."""

    def __init__(self, dataset: VOC12ClsPix, batch_size: int, device, num_workers: int = 0, shuffle: bool = True,
                 drop_last: bool = True, generator=None):
        from torch.utils.data import DataLoader
        self.dataset, self.stager = dataset, InputStager(device, batch_size, dataset.crop, dataset.view)
        self.loader = DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, drop_last=drop_last,
                                 collate_fn=_keep, worker_init_fn=_reference_worker_init if num_workers else None,
                                 generator=generator, persistent_workers=bool(num_workers))

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for items in self.loader:
            names = [it[0] for it in items]
            labels = None
            if items[0][2] is not None:
                labels = torch.from_numpy(np.stack([it[2] for it in items]))
            yield names, self.stager([it[1] for it in items], labels=labels)


class MSFStager:
    """The multi-scale + flip list of `VOC12ClsDatasetMSF.__getitem__` (src/data.py:336-365) as `infer_mcl.py:123-125` feeds
    it to the model (`img.cuda().float()`), built on the device: the decoded image crosses PCIe once as uint8; per scale
    `mx_resample` (= `img.resize(target, PIL.Image.CUBIC)`, bit-exact) and `mx_input_stage` (color_norm, HWC -> CHW, the
    float64 -> float32 rounding of `.float()`); the flipped copy is `np.flip(x, -1)`.  On the host the four bicubic resizes
    of a 500 x 375 image cost ~40 ms of a core per image; here ~0.5 ms of coefficient tables."""

    def __init__(self, device, max_side: int = 2048):
        self.dev, self.side = device, max_side
        cap = max_side * max_side * 3
        self._src = torch.empty(cap, dtype=torch.uint8, device=device)
        self._tmp = torch.empty(cap, dtype=torch.uint8, device=device)
        self._dst = torch.empty(cap, dtype=torch.uint8, device=device)

    def __call__(self, pil_img, scales=(0.5, 1.0, 1.5, 2.0), unit: int = 1) -> List[torch.Tensor]:
        a = np.array(pil_img.convert("RGB"))                    # (a writable copy: torch.from_numpy wants one)
        h, w = a.shape[:2]
        rounded = (int(round(w / unit) * unit), int(round(h / unit) * unit))                  # data.py:347
        if max(h, w) > self.side:
            raise ValueError(f"image {w}x{h} exceeds MSFStager(max_side={self.side})")
        self._src[:a.size].copy_(torch.from_numpy(a).reshape(-1), non_blocking=False)
        out: List[torch.Tensor] = []
        for s in scales:
            tw, th = round(rounded[0] * s), round(rounded[1] * s)                               # data.py:351-352
            if max(tw, th) > self.side:
                raise ValueError(f"scale {s}: {tw}x{th} exceeds MSFStager(max_side={self.side})")
            tab = torch.from_numpy(resample_tables(w, h, tw, th, "bicubic")).to(self.dev)
            rs = torch.tensor([0, h, w, 0, 0, tw, th, 0], dtype=torch.int32, device=self.dev)
            call("mx_resample", ptr(self._src), ptr(rs), ptr(tab), ptr(self._tmp), ptr(self._dst), 1, max(h * tw, th * tw), stream())
            job = torch.tensor([0, th, tw, 0, 0, 0, 0, 0], dtype=torch.int32, device=self.dev)
            x = torch.empty(1, 3, th, tw, dtype=torch.float32, device=self.dev)
            call("mx_input_stage", ptr(self._dst), ptr(job), ptr(x), 1, th, tw, stream())
            out.append(x)
            out.append(torch.flip(x, dims=[-1]))                                               # data.py:363
        return out
