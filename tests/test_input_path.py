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
        p = D.plan_item(im)
        assert np.array_equal(np.array([p.coord1, p.coord2, p.ori_coord]), G[f"coords{i}"]), i
        assert p.view1_u8.shape == (224, 224, 3) and p.view1_u8.dtype == np.uint8
        assert p.img_u8.shape[0] <= 448 and p.img_u8.shape[1] <= 448


@pytest.mark.gpu
def test_device_stage_bit_exact_with_reference():
    from muscle_amd import data as D
    _seed()
    plans = [D.plan_item(im) for im in _images()]
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
    p = D.plan_item(im, resize_long=(300, 400))
    b = stager([p])
    assert np.array_equal(b["img"][0].cpu().numpy(), want[0]) and float((b["img"][0] == 0).float().mean()) > 0.1
    assert np.array_equal(b["view2"][0].cpu().numpy(), want[2].astype(np.float32))
