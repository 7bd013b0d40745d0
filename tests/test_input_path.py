"""Input path: the reference's own VOC12ImageViews.__getitem__ outputs
(tests/golden/input_views.npz, oracle/gen_golden.py::gen_input_units) against
  * the oracle's numpy restatement (CPU),
  * the host-side planner of muscle_amd.data (geometry, draw order; CPU),
  * the device stage mx_input_stage through InputStager (GPU): bit for bit (CRC-32 of the fp32 bytes)."""
import io
import random
import zlib

import numpy as np
import pytest
import torch

import golden_util as gu

G = gu.load("input_views.npz")
N_ITEMS = 6


def _images():
    import PIL.Image
    return [PIL.Image.open(io.BytesIO(G[f"jpg{i}"].tobytes())).convert("RGB") for i in range(N_ITEMS)]


def _seed():
    torch.manual_seed(int(G["seeds"][0]))
    random.seed(int(G["seeds"][1]))


def test_oracle_item_matches_reference():
    from oracle import mcl_oracle as O
    _seed()
    for i, im in enumerate(_images()):
        img, v1, v2, c1, c2, oc = O.input_item(im)
        assert img.dtype == np.float32 and v1.dtype == np.float64 and img.shape == (3, 448, 448) and v1.shape == (3, 224, 224)
        assert np.array_equal(np.array([c1, c2, oc]), G[f"coords{i}"]), i
        v1f, v2f = v1.astype(np.float32), v2.astype(np.float32)
        assert [zlib.crc32(img.tobytes()), zlib.crc32(v1f.tobytes()), zlib.crc32(v2f.tobytes())] == G[f"crc{i}"].tolist(), i
        assert np.array_equal(img[:, ::16, ::16], G[f"img_s{i}"]) and np.array_equal(v1f[:, ::16, ::16], G[f"v1_s{i}"])


def test_get_inter_general_form_equals_reference_branches():
    """muscle_amd.data.get_inter collapses the reference's four corner branches (src/data.py:247-267) into one formula:
    identical on every pair of equal-size boxes, including ties and disjoint boxes."""
    from muscle_amd import data as D
    from oracle import mcl_oracle as O
    rng = np.random.default_rng(0)
    for _ in range(3000):
        th, tw = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        c1 = (int(rng.integers(0, 50)), int(rng.integers(0, 50)), th, tw)
        c2 = (int(rng.integers(0, 50)), int(rng.integers(0, 50)), th, tw)
        assert D.get_inter(c1, c2) == O.input_get_inter(c1, c2)
    assert D.get_inter((0, 0, 5, 5), (5, 0, 5, 5)) == (False, False, False)


def test_planner_follows_the_reference_draw_order():
    from muscle_amd import data as D
    _seed()
    for i, im in enumerate(_images()):
        p = D.plan_item(im, augment=False)
        assert np.array_equal(np.array([p.coord1, p.coord2, p.ori_coord]), G[f"coords{i}"]), i
        assert p.view1_u8.shape == (224, 224, 3) and p.view1_u8.dtype == np.uint8
        assert p.img_u8.shape[0] <= 448 and p.img_u8.shape[1] <= 448


@pytest.mark.gpu
def test_device_stage_bit_exact_with_reference():
    from muscle_amd import data as D
    _seed()
    plans = [D.plan_item(im, augment=False) for im in _images()]
    dev = torch.device("cuda:0")
    stager = D.InputStager(dev, batch=8)
    labels = torch.zeros(N_ITEMS, 20)
    for rep in range(3):                                   # both pinned buffers, reuse
        b = stager(plans, labels)
        torch.cuda.synchronize()
        assert b["img"].shape == (N_ITEMS, 3, 448, 448) and b["view1"].shape == (N_ITEMS, 3, 224, 224) and b["label"].is_cuda
        img, v1, v2 = b["img"].cpu().numpy(), b["view1"].cpu().numpy(), b["view2"].cpu().numpy()
        for i in range(N_ITEMS):
            got = [zlib.crc32(img[i].tobytes()), zlib.crc32(v1[i].tobytes()), zlib.crc32(v2[i].tobytes())]
            assert got == G[f"crc{i}"].tolist(), (rep, i)
            assert b["coord1"][i].tolist() == G[f"coords{i}"][0].tolist() and b["ori_coord"][i].tolist() == G[f"coords{i}"][2].tolist()
    # an item smaller than the container in one dimension: zero border, placement honoured (synthetic, against the oracle)
    from oracle import mcl_oracle as O
    import PIL.Image
    torch.manual_seed(5); random.seed(6)
    im = PIL.Image.fromarray((np.arange(500 * 460 * 3) % 251).astype(np.uint8).reshape(460, 500, 3))
    st = (torch.get_rng_state(), random.getstate())
    want = O.input_item(im, resize_long=(300, 400))
    torch.set_rng_state(st[0]); random.setstate(st[1])
    p = D.plan_item(im, resize_long=(300, 400), augment=False)
    b = stager([p])
    assert np.array_equal(b["img"][0].cpu().numpy(), want[0]) and float((b["img"][0] == 0).float().mean()) > 0.1
    assert np.array_equal(b["view2"][0].cpu().numpy(), want[2].astype(np.float32))


# ---- ColorJitter / RandomErasing: restated from torchvision 0.9.0 (not installed here): PARITY UNPINNED ----------------
# What can be checked without torchvision: the draws they take from torch's generator (count, order, ranges - as the
# published source has them), the identities of the PIL adjustments, the erase box on the device, and that switching the
# augmentations off leaves exactly the draw sequence the pinned fixtures were made with.
def test_color_jitter_draws_and_identities():
    from muscle_amd import data as D
    import PIL.Image
    torch.manual_seed(5)
    order, b, c, s, h = D.color_jitter_params()
    torch.manual_seed(5)
    want_order = torch.randperm(4).tolist()
    draws = [float(torch.empty(1).uniform_(lo, hi)) for lo, hi in ((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))]
    assert order == want_order and [b, c, s, h] == draws
    assert 0.8 <= b <= 1.2 and 0.8 <= c <= 1.2 and 0.8 <= s <= 1.2 and -0.1 <= h <= 0.1
    im = _images()[0]
    same = D.apply_color_jitter(im, ([0, 1, 2, 3], 1.0, 1.0, 1.0, None))
    assert np.array_equal(np.asarray(same), np.asarray(im))                       # factor 1 leaves the image alone
    dark = D.apply_color_jitter(im, ([0, 1, 2, 3], 0.5, None, None, None))
    a, d = np.asarray(im).astype(np.int32), np.asarray(dark).astype(np.int32)
    assert np.abs(d - a // 2).max() <= 1                                           # ImageEnhance.Brightness: blend with black
    grey = D.apply_color_jitter(im, ([2], None, None, 0.0, None))
    g = np.asarray(grey)
    assert np.abs(g[..., 0].astype(int) - g[..., 1]).max() <= 1 and np.abs(g[..., 1].astype(int) - g[..., 2]).max() <= 1
    hue0 = D.apply_color_jitter(im, ([3], None, None, None, 0.0))                  # HSV round trip only
    assert np.abs(np.asarray(hue0).astype(int) - a).max() <= 4
    with pytest.raises(ValueError):
        D.apply_color_jitter(im, ([3], None, None, None, 0.7))
    assert isinstance(D.apply_color_jitter(im, D.color_jitter_params()), PIL.Image.Image)


def test_random_erasing_params_follow_the_published_loop():
    from muscle_amd import data as D
    import math
    for seed in range(40):
        torch.manual_seed(seed)
        box = D.random_erasing_params(448, 448)
        torch.manual_seed(seed)
        if not (torch.rand(1) < 0.5):
            assert box is None
            continue
        want = None
        for _ in range(10):
            area = 448 * 448 * torch.empty(1).uniform_(0.02, 0.2).item()
            ar = torch.empty(1).uniform_(0.3, 3.3).item()
            hh, ww = int(round(math.sqrt(area * ar))), int(round(math.sqrt(area / ar)))
            if not (hh < 448 and ww < 448):
                continue
            want = (torch.randint(0, 448 - hh + 1, size=(1,)).item(), torch.randint(0, 448 - ww + 1, size=(1,)).item(), hh, ww)
            break
        assert box == want
        if box is not None:
            i, j, hh, ww = box
            assert 0 <= i and i + hh <= 448 and 0 <= j and j + ww <= 448 and 0.02 * 448 * 448 * 0.9 <= hh * ww <= 0.2 * 448 * 448 * 1.1


def test_augment_off_consumes_exactly_the_pinned_draws():
    """augment=True takes extra draws (3 x (randperm + 4 uniforms) + the erasing draws) but must not move the ones the
    reference fixture pins: flip, view boxes and coordinates come BEFORE any augmentation draw."""
    from muscle_amd import data as D
    im = _images()[0]
    _seed()
    p0 = D.plan_item(im, augment=False)
    _seed()
    p1 = D.plan_item(im, augment=True)
    assert (p0.coord1, p0.coord2, p0.ori_coord) == (p1.coord1, p1.coord2, p1.ori_coord)
    assert p0.erase is None and p1.view1_u8.shape == p0.view1_u8.shape
    assert p1.img_u8.shape[2] == 3 and p1.img_u8.dtype == np.uint8


@pytest.mark.gpu
def test_device_stage_applies_the_erase_box():
    from muscle_amd import data as D
    dev = torch.device("cuda:0")
    _seed()
    plans = [D.plan_item(im, augment=False) for im in _images()[:3]]
    st = D.InputStager(dev, batch=3)
    base = st(plans)["img"].clone()
    plans[0].erase = (10, 20, 100, 57)
    plans[2].erase = (0, 0, 447, 1)
    out = st(plans)["img"]
    want = base.clone()
    want[0, :, 10:110, 20:77] = 0
    want[2, :, 0:447, 0:1] = 0
    assert torch.equal(out, want)
    assert torch.equal(out[1], base[1])


@pytest.mark.gpu
def test_staged_batch_feeds_the_full_loop_body():
    """End to end: decoded images -> plan_item (with the augmentations) -> InputStager -> mcl_step at epoch 12 (phase 2 reads
    the views and the overlap coordinates): the batch dict is what the loop body expects and every loss is finite."""
    import muscle_amd
    from muscle_amd import data as D
    dev = torch.device("cuda:0")
    _seed()
    ims = _images()[:4]
    plans = [D.plan_item(im) for im in ims]
    labels = torch.zeros(4, 20)
    labels[:, 3] = 1; labels[1, 7] = 1; labels[2, 7] = 1
    batch = D.InputStager(dev, batch=4)(plans, labels=labels)
    assert batch["img"].shape == (4, 3, 448, 448) and batch["view1"].shape == (4, 3, 224, 224)
    assert batch["coord1"].dtype == torch.int64 and batch["coord1"].shape == (4, 4)
    torch.manual_seed(0)
    model = muscle_amd.MuSCLe(21, "efficientnet-b0", layers=3, last_pooling=False).to(dev)
    opt = muscle_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=5e-5)
    out = muscle_amd.mcl_step(model, opt, batch, 12)
    for k, v in out.items():
        assert np.isfinite(float(v)), k


def _voc_tree(tmp_path):
    """A tiny VOC-style tree from the fixture JPEGs: JPEGImages/*.jpg, a train list in the reference's format, labels."""
    root = tmp_path / "VOC2012"
    (root / "JPEGImages").mkdir(parents=True)
    names = []
    for i in range(N_ITEMS):
        nm = f"2007_{i:06d}"
        (root / "JPEGImages" / (nm + ".jpg")).write_bytes(G[f"jpg{i}"].tobytes())
        names.append(nm)
    lst = tmp_path / "train_aug.txt"
    lst.write_text("\n".join(f"/JPEGImages/{n}.jpg /SegmentationClassAug/{n}.png" for n in names) + "\n")
    labels = {n: np.eye(20, dtype=np.float32)[i % 20] for i, n in enumerate(names)}
    return str(lst), str(root), labels, names


def test_dataset_and_worker_loader(tmp_path):
    """VOC12ClsPix reads the reference's list format; torch DataLoader workers run the host half and hand the plans back
    intact; without workers the loader is exactly sequential `plan` calls on the caller's generators."""
    from torch.utils.data import DataLoader
    from muscle_amd import data as D
    lst, root, labels, names = _voc_tree(tmp_path)
    ds = D.VOC12ClsPix(lst, root, labels=labels, augment=True)
    assert len(ds) == N_ITEMS and ds.names == names
    _seed()
    torch.empty((), dtype=torch.int64).random_()      # the base seed a DataLoader iterator draws first (also in the reference)
    direct = [ds.plan(i) for i in range(N_ITEMS)]
    _seed()
    seq = [it for batch in DataLoader(ds, batch_size=2, shuffle=False, num_workers=0, collate_fn=D._keep) for it in batch]
    for (n0, p0, l0), (n1, p1, l1) in zip(direct, seq):
        assert n0 == n1 and np.array_equal(l0, l1) and p0.coord1 == p1.coord1 and p0.erase == p1.erase
        assert np.array_equal(p0.img_u8, p1.img_u8) and np.array_equal(p0.view2_u8, p1.view2_u8)
    got = []
    for batch in DataLoader(ds, batch_size=3, shuffle=False, num_workers=2, collate_fn=D._keep, worker_init_fn=D._reference_worker_init):
        assert len(batch) == 3
        for name, p, lab in batch:
            assert p.img_u8.dtype == np.uint8 and p.img_u8.shape[2] == 3 and p.view1_u8.shape == (224, 224, 3)
            assert max(p.img_u8.shape[:2]) <= 768 and p.img_crop[2] <= 448 and p.img_crop[3] <= 448 and lab.shape == (20,)   # whole resized image: the jitter runs on the device
            got.append(name)
    assert got == names


def _check_staged_batches(loader, names):
    assert len(loader) == 3
    seen = []
    for nm, batch in loader:
        seen += nm
        assert batch["img"].shape == (2, 3, 448, 448) and batch["img"].is_cuda and batch["label"].shape == (2, 20)
        assert batch["view2"].shape == (2, 3, 224, 224) and batch["ori_coord"].shape == (2, 4)
        assert torch.isfinite(batch["img"]).all()
    assert sorted(seen) == names


@pytest.mark.gpu
def test_staged_loader_yields_loop_body_batches(tmp_path):
    """In this process the host half runs inline (num_workers=0): the pytest process has the HIP runtime's threads and
    locks by now, and DataLoader workers are fork()ed children - in round 4 the driver's run died inside this test with a
    worker forked out of the GPU-initialised test process.  The worker path on the GPU box is the next test's, in a fresh
    child; the workers' seeding / hand-back is test_dataset_and_worker_loader's (CPU)."""
    from muscle_amd import data as D
    lst, root, labels, names = _voc_tree(tmp_path)
    ds = D.VOC12ClsPix(lst, root, labels=labels)
    loader = D.StagedLoader(ds, batch_size=2, device=torch.device("cuda:0"), num_workers=0, shuffle=True,
                            generator=torch.Generator().manual_seed(3))
    _check_staged_batches(loader, names)


@pytest.mark.gpu
def test_staged_loader_with_workers_in_a_fresh_process(tmp_path):
    """StagedLoader(num_workers=2) the way a training script runs it: a fresh process creates the stager (GPU initialised),
    then the DataLoader forks its workers at the first iteration (train_mcl.py:130-153 has the same order).  Child process
    with its own time limit, like tests/test_gpu_dist.py."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); torch.set_num_threads(2)\n"
            "import pathlib, test_input_path as t\n"
            "from muscle_amd import data as D\n"
            "lst, root, labels, names = t._voc_tree(pathlib.Path(sys.argv[1]))\n"
            "ds = D.VOC12ClsPix(lst, root, labels=labels)\n"
            "loader = D.StagedLoader(ds, batch_size=2, device=torch.device('cuda:0'), num_workers=2, shuffle=True,\n"
            "                        generator=torch.Generator().manual_seed(3))\n"
            "t._check_staged_batches(loader, names); t._check_staged_batches(loader, names)\n"      # second epoch: persistent workers
            "print('LOADER-OK')\n") % (here, os.path.dirname(here))
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], capture_output=True, text=True, timeout=150)
    assert r.returncode == 0 and "LOADER-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_device_color_jitter_is_bit_exact_with_pil():
    """mx_color_jitter against the PIL calls torchvision's ColorJitter makes (Pillow itself is the comparator): same seeds,
    once with the pixel work on the host, once on the device - the staged fp32 tensors must be identical, for every image,
    for many parameter draws (all adjustment orders occur), with and without the erase box."""
    from muscle_amd import data as D
    dev = torch.device("cuda:0")
    ims = _images()
    st = D.InputStager(dev, batch=len(ims))
    orders = set()
    for rep in range(6):
        torch.manual_seed(100 + rep); random.seed(200 + rep)
        host = [D.plan_item(im, augment=True, device_jitter=False) for im in ims]
        torch.manual_seed(100 + rep); random.seed(200 + rep)
        devp = [D.plan_item(im, augment=True, device_jitter=True) for im in ims]
        for h, d in zip(host, devp):
            assert h.coord1 == d.coord1 and h.erase == d.erase and h.img_place == d.img_place and d.jitter is not None
            orders.add(tuple(d.jitter[0][0]))
        a = {k: v.clone() for k, v in st(host).items()}
        b = st(devp)
        for k in ("img", "view1", "view2"):
            assert torch.equal(a[k], b[k]), (rep, k, float((a[k] - b[k]).abs().max()))
    assert len(orders) >= 8


@pytest.mark.gpu
def test_device_color_jitter_extreme_factors():
    """Factors outside [0, 1] take ImagingBlend's extrapolation branch (clamping); 0 and 1 the interpolation branch."""
    from muscle_amd import data as D
    dev = torch.device("cuda:0")
    im = _images()[2]
    st = D.InputStager(dev, batch=1)
    for params in (([0, 1, 2, 3], 1.9, 1.7, 0.0, 0.5), ([3, 2, 1, 0], 0.0, 1.0, 2.0, -0.5), ([1, 3, 0, 2], 0.31, 0.0, 1.0, 0.123),
                   ([2, 0, 3, 1], 1.0, 3.0, 0.5, -0.25)):
        torch.manual_seed(1); random.seed(1)
        d = D.plan_item(im, augment=True, device_jitter=True)
        d.jitter = [params, params, params]
        torch.manual_seed(1); random.seed(1)
        h = D.plan_item(im, augment=False)
        # host reference: the same geometry, PIL adjustments applied to the whole resized image / the views
        import PIL.Image
        big = PIL.Image.fromarray(d.img_u8, "RGB")
        it, il, ch, cw = d.img_crop
        h.img_u8 = np.ascontiguousarray(np.asarray(D.apply_color_jitter(big, params))[it:it + ch, il:il + cw])
        h.view1_u8 = np.ascontiguousarray(np.asarray(D.apply_color_jitter(PIL.Image.fromarray(d.view1_u8, "RGB"), params)))
        h.view2_u8 = np.ascontiguousarray(np.asarray(D.apply_color_jitter(PIL.Image.fromarray(d.view2_u8, "RGB"), params)))
        h.erase = d.erase
        a = {k: v.clone() for k, v in st([h]).items()}
        b = st([d])
        for k in ("img", "view1", "view2"):
            assert torch.equal(a[k], b[k]), (params, k)


@pytest.mark.gpu
def test_device_hue_and_blends_over_all_16m_colours():
    """mx_color_jitter called directly on a 4096 x 4096 image holding every RGB colour once: the hue path (Pillow's
    rgb2hsv_row / hsv2rgb) for several shifts and the three blends at an interpolating and an extrapolating factor must
    equal Pillow's own output for every colour."""
    import PIL.Image
    from muscle_amd import data as D
    from muscle_amd._lib import call, ptr, stream
    dev = torch.device("cuda:0")
    r, g, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    rgb = np.ascontiguousarray(np.stack([r.ravel(), g.ravel(), b.ravel()], 1).reshape(4096, 4096, 3))
    im = PIL.Image.fromarray(rgb, "RGB")
    sums = torch.zeros(1, dtype=torch.int64, device=dev)
    for params in (([3, 0, 1, 2], None, None, None, 37 / 255 + 1e-4), ([3, 0, 1, 2], None, None, None, -0.5), ([3, 0, 1, 2], None, None, None, 0.0),
                   ([0, 1, 2, 3], 0.83, None, None, None), ([0, 1, 2, 3], 1.19, None, None, None), ([0, 1, 2, 3], None, 0.9, None, None),
                   ([0, 1, 2, 3], None, 1.15, None, None), ([0, 1, 2, 3], None, None, 0.87, None), ([0, 1, 2, 3], None, None, 1.2, None),
                   ([2, 3, 1, 0], 1.1, 0.9, 1.05, 0.07)):
        want = np.asarray(D.apply_color_jitter(im, params))
        order, fb, fc, fs, fh = params
        code = 0
        for pos in range(4):
            fn = order[pos]
            code |= (fn if (fb, fc, fs, fh)[fn] is not None else 15) << (4 * pos)
        job = np.zeros(8, dtype=np.int32)
        job[:4] = (0, 4096, 4096, code)
        job.view(np.float32)[4:7] = (fb or 0.0, fc or 0.0, fs or 0.0)
        job[7] = (int(fh * 255) & 0xFF) if fh is not None else 0
        buf = torch.from_numpy(rgb.copy()).to(dev)
        jobs = torch.from_numpy(job).to(dev)
        call("mx_color_jitter", ptr(buf), ptr(jobs), ptr(sums), 1, 4096 * 4096, stream())
        got = buf.cpu().numpy()
        assert np.array_equal(got, want), (params, int((got != want).any(-1).sum()))


def test_resample_tables_reproduce_pil_resize():
    """The coefficient tables (Pillow's precompute_coeffs / normalize_coeffs_8bpc restated) applied with the integer
    arithmetic of ImagingResample give PIL.Image.resize bit for bit (numpy here, the same loops as the HIP kernels)."""
    import PIL.Image
    from muscle_amd import data as D

    def apply(a, t, wout, hout):
        ksh, ksv = int(t[0]), int(t[1])
        o = 2
        bh = t[o:o + 2 * wout].reshape(wout, 2); o += 2 * wout
        kh = t[o:o + wout * ksh].reshape(wout, ksh); o += wout * ksh
        bv = t[o:o + 2 * hout].reshape(hout, 2); o += 2 * hout
        kv = t[o:o + hout * ksv].reshape(hout, ksv)
        tmp = np.zeros((a.shape[0], wout, 3), np.uint8)
        for xx in range(wout):
            x0, n = bh[xx]
            ss = (a[:, x0:x0 + n].astype(np.int64) * kh[xx, :n, None].astype(np.int64)).sum(1) + (1 << 21)
            tmp[:, xx] = np.clip(ss >> 22, 0, 255)
        out = np.zeros((hout, wout, 3), np.uint8)
        for yy in range(hout):
            y0, n = bv[yy]
            ss = (tmp[y0:y0 + n].astype(np.int64) * kv[yy, :n, None, None].astype(np.int64)).sum(0) + (1 << 21)
            out[yy] = np.clip(ss >> 22, 0, 255)
        return out
    for im, (tw, th), filt, pf in zip(_images(), [(700, 525), (448, 597), (768, 576), (500, 375), (333, 448), (448, 448)],
                                      ["bicubic"] * 5 + ["bilinear"], [PIL.Image.BICUBIC] * 5 + [PIL.Image.BILINEAR]):
        a = np.asarray(im)
        t = D.resample_tables(a.shape[1], a.shape[0], tw, th, filt)
        assert np.array_equal(apply(a, t, tw, th), np.asarray(im.resize((tw, th), resample=pf))), (im.size, tw, th)


@pytest.mark.gpu
@pytest.mark.parametrize("augment", [False, True])
def test_device_resize_is_bit_exact_with_pil(augment):
    """RandomResizeLong on the device (mx_resample) against PIL.Image.resize on the host: same seeds, the staged tensors
    must be identical - alone, and followed by the device ColorJitter (whose contrast step needs the resized image's mean)."""
    from muscle_amd import data as D
    dev = torch.device("cuda:0")
    ims = _images()
    st = D.InputStager(dev, batch=len(ims))
    for rep in range(4):
        torch.manual_seed(300 + rep); random.seed(400 + rep)
        host = [D.plan_item(im, augment=augment, device_jitter=False, device_resize=False) for im in ims]
        torch.manual_seed(300 + rep); random.seed(400 + rep)
        devp = [D.plan_item(im, augment=augment, device_jitter=augment, device_resize=True) for im in ims]
        assert all(d.resize_to is not None and d.img_u8.shape[:2] == (im.size[1], im.size[0]) for d, im in zip(devp, ims))
        a = {k: v.clone() for k, v in st(host).items()}
        b = st(devp)
        for k in ("img", "view1", "view2"):
            assert torch.equal(a[k], b[k]), (rep, k, int((a[k] != b[k]).sum()))


@pytest.mark.gpu
def test_msf_list_on_the_device_equals_the_reference_dataset_arithmetic():
    """MSFStager against VOC12ClsDatasetMSF.__getitem__'s arithmetic done on the host (PIL bicubic resize, color_norm in
    float64, HWC_to_CHW, np.flip) followed by infer_mcl.py's .float(): bit for bit, every scale, odd sizes, both flips."""
    import PIL.Image
    from muscle_amd import data as D
    dev = torch.device("cuda:0")
    ms = D.MSFStager(dev, max_side=1400)
    mean = np.array([[[0.485, 0.456, 0.406]]]); std = np.array([[[0.229, 0.224, 0.225]]])
    for im in _images()[:4]:
        got = ms(im, scales=(0.5, 1.0, 1.5, 2.0))
        assert len(got) == 8
        W, H = im.size
        for i, s in enumerate((0.5, 1.0, 1.5, 2.0)):
            target = (round(W * s), round(H * s))
            arr = np.asarray(im.resize(target, resample=PIL.Image.BICUBIC))
            x = np.transpose((arr / 255 - mean) / std, (2, 0, 1))                 # color_norm (float64) + HWC_to_CHW
            want = torch.from_numpy(x.copy()).float()[None]
            wantf = torch.from_numpy(np.flip(x, -1).copy()).float()[None]
            assert torch.equal(got[2 * i].cpu(), want), (im.size, s)
            assert torch.equal(got[2 * i + 1].cpu(), wantf), (im.size, s, "flip")
