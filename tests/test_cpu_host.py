"""CPU-only checks: the C ABI library loads and exports every symbol of include/muscle_hip.h, the module's
state_dict matches the reference key contract, host-side logic (arch tables, crop planning, gating) and the
data-parallel hook over gloo with two ranks.  No kernel is launched here."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_util as gu
from muscle_amd import _lib, arch, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_library_exports_every_declared_symbol():
    sigs = _lib.parse_header()
    assert len(sigs) >= 40
    if not os.path.exists(_lib.LIB_PATH):
        from muscle_amd import _build
        _build.build(verbose=False)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in list(sigs) + ["mx_last_error"]:
        assert hasattr(L, name), name
    L.mx_version.restype = ctypes.c_int
    assert L.mx_version() >= 100
    # argument errors are reported before any launch (no GPU needed)
    LL = _lib.lib()
    rc = LL.mx_colstats(None, 0, 0, None, None)
    assert rc < 0 and b"colstats" in LL.mx_last_error()


def test_abi_hash_guards_against_a_stale_library(monkeypatch):
    """The library carries the hash of the header it was built from; _lib.lib() refuses a library built from another one
    (ctypes signatures come from the header on disk, so a stale .so would be called with a shifted argument layout)."""
    from muscle_amd import _build
    L = _lib.lib()
    assert L.mx_abi_hash() == _build.abi_hash() != 0
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_build, "abi_hash", lambda path=None: 12345)
    with pytest.raises(_lib.MuscleHipError, match="rebuild"):
        _lib.lib()


def test_fused_adam_state_dict_round_trip():
    """Checkpointing keeps the moments and step counts (torch.optim.Adam layout), not the arenas."""
    import muscle_amd
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
    opt = muscle_amd.FusedAdam(ps, lr=1e-3, weight_decay=5e-5)
    g = opt.param_groups[0]
    opt._flatten(g)                                          # the arena bookkeeping itself is device independent
    g["_m"].copy_(torch.arange(g["_m"].numel(), dtype=torch.float32))
    g["_v"].copy_(torch.arange(g["_v"].numel(), dtype=torch.float32) * 2)
    g["_steps"][:] = [3, 0, 5]
    sd = opt.state_dict()
    assert sorted(sd["state"]) == [0, 2] and sd["state"][2]["step"] == 5
    assert not any(k.startswith("_") for k in sd["param_groups"][0]) and sd["param_groups"][0]["params"] == [0, 1, 2]
    assert sd["state"][0]["exp_avg"].shape == (5, 3)
    import io
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    sd2 = torch.load(buf)
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = muscle_amd.FusedAdam(ps2, lr=1.0)
    opt2.load_state_dict(sd2)
    g2 = opt2.param_groups[0]
    assert g2["lr"] == 1e-3 and g2["weight_decay"] == 5e-5
    opt2._flatten(g2)
    opt2._apply_pending(g2)
    assert g2["_steps"] == [3, 0, 5]
    o0, o2 = g2["_offs"][0], g2["_offs"][2]
    assert torch.equal(g2["_m"][o0:o0 + 15], g["_m"][g["_offs"][0]:g["_offs"][0] + 15])
    assert torch.equal(g2["_v"][o2:o2 + 4], g["_v"][g["_offs"][2]:g["_offs"][2] + 4])
    o1 = g2["_offs"][1]
    assert float(g2["_m"][o1:o1 + 7].abs().sum()) == 0.0     # the parameter that never stepped has no state


def test_header_cites_reference_lines():
    text = open(_lib.HEADER_PATH).read()
    for cite in ("model.py", "utils.py", "MuSCLe.py", "loss_multilabel.py", "train_mcl.py", "torchutils.py"):
        assert cite in text


@pytest.mark.parametrize("name", ["efficientnet-b0", "efficientnet-b3", "efficientnet-b7"])
def test_state_dict_contract(name):
    import muscle_amd
    cfg = arch.net_cfg(name, False)
    m = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False)
    spec = synth.state_dict_spec(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    for k, shp in spec.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state_dict(cfg, 1).items()}, strict=True)
    live = {id(p) for p in m.live_parameters("cam")}
    dead = [k for k, p in m.named_parameters() if id(p) not in live]
    assert dead == ["backbone._conv_head.weight", "backbone._bn1.weight", "backbone._bn1.bias", "backbone._fc.weight",
                    "backbone._fc.bias", "fuse_dec.weight", "fuse_dec.bias"]
    assert all(id(p) in live for p in m.live_parameters("pix"))
    assert id(m.fc.weight) not in {id(p) for p in m.live_parameters("pix")}


def test_state_dict_contract_decoder():
    import muscle_amd
    cfg = arch.net_cfg("efficientnet-b7", True)
    m = muscle_amd.MuSCLe(21, "efficientnet-b7", layers=3, last_pooling=True, mode="dec")
    spec = synth.state_dict_spec(cfg, mode="dec", layers=3)
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in spec.items())


def test_no_cpu_fallback():
    import muscle_amd
    m = muscle_amd.MuSCLe(21, "efficientnet-b0", last_pooling=False)
    with pytest.raises(_lib.MuscleHipError):
        m(torch.zeros(1, 3, 32, 32))
    with pytest.raises(_lib.MuscleHipError):
        muscle_amd.FocalLoss()(torch.rand(2, 20), torch.zeros(2, 20))
    d = muscle_amd.MuSCLe(21, "efficientnet-b3", layers=3, last_pooling=True, mode="dec")
    with pytest.raises(_lib.MuscleHipError):
        d(torch.zeros(1, 3, 32, 32), cam="seg")
    with pytest.raises(AttributeError):
        d(torch.zeros(1, 3, 32, 32), cam="cam")


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "muscle_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in src and "from oracle" not in src and "oracle." not in src, f


def test_arch_tables_match_reference_fixture():
    # shapes of every block output as the reference produced them (forward_b0/b3 fixtures)
    for fname, size in (("forward_b0.npz", 64), ("forward_b3.npz", 64)):
        G = gu.load(fname)
        cfg = arch.net_cfg(str(G["name"]), False)
        h = cfg.stem_out_size(size)
        for b, shp in zip(cfg.blocks, G["feat_shapes"]):
            h = b.out_size(h)
            assert [b.cout, h, h] == list(shp[1:]), b.index
    cfg7 = arch.net_cfg("efficientnet-b7", False)
    assert abs(arch.forward_macs(cfg7, 448)["total"] / 1e9 - 43.66) < 0.01
    assert abs(arch.min_materialisation_bytes(cfg7, 448) / 1e9 - 2.152) < 0.001


def test_crop_planning_matches_reference_draws():
    from muscle_amd import phase2
    U = gu.load("units.npz")
    c1, c2 = torch.from_numpy(U["dc_coord1"]), torch.from_numpy(U["dc_coord2"])
    np.random.seed(9)
    plan = phase2._plan_crops(c1, c2, None)
    assert plan.bidx == U["dc_bidx"].tolist()
    assert [[i, h, w] for i, s in enumerate(plan.per1) for (_, h, w) in s] == U["dc_shapes1"].tolist()
    assert [[i, h, w] for i, s in enumerate(plan.per2) for (_, h, w) in s] == U["dc_shapes2"].tolist()
    plan2 = phase2._plan_crops(c1, c2, gu.geometry_from_draws(U["dc_coord1"], U["dc_draws"]))
    assert plan2.pairs == plan.pairs and len(plan.pairs) == sum(len(a) * len(b) for a, b in zip(plan.per1, plan.per2))


def test_synth_is_deterministic():
    a = synth.synth_batch(4, 16, 8, 3)
    b = synth.synth_batch(4, 16, 8, 3)
    for k in a:
        assert np.array_equal(a[k], b[k])
    lab = synth.synth_labels(32, 0)
    assert lab.sum() <= 105 and (lab.sum(1) >= 1).all() and np.array_equal(lab[0], lab[1])


_DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from muscle_amd.dist import GradAverager, broadcast_parameters
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
r = dist.get_rank()
class Sink: pass
class Model(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.w = torch.nn.Parameter(torch.full((5,), float(r + 1))); self.register_buffer("b", torch.full((2,), float(r)))
m = Model(); broadcast_parameters(m)
assert torch.equal(m.w.data, torch.full((5,), 1.0)) and torch.equal(m.b, torch.zeros(2))
m.last_grad_sink = Sink(); m.last_grad_sink.arena = torch.arange(8, dtype=torch.float32) * (r + 1)
h = GradAverager(); h(m, 1)
assert torch.allclose(m.last_grad_sink.arena, torch.arange(8, dtype=torch.float32) * 1.5), m.last_grad_sink.arena
assert h.bytes_reduced == 32
dist.destroy_process_group(); print("ok", r)
'''


def test_dp_hook_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_chunked_exchange_equals_single_allreduce(tmp_path):
    """muscle_amd.dist.GradAverager: the ~25 MB chunks launched as backward reports progress give exactly the arena one
    all-reduce gives (two gloo ranks, CPU tensors)."""
    worker = os.path.join(ROOT, "tests", "dp_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    outs = [str(tmp_path / f"a{r}.npz") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, ROOT, "cpu-arena", outs[r]], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    a0, a1 = np.load(outs[0]), np.load(outs[1])
    want = (np.arange(1003, dtype=np.float32) * 1 + 0.0 + np.arange(1003, dtype=np.float32) * 2 + 0.25) / 2
    for tag in ("single", "chunked", "chunked_late"):
        assert np.array_equal(a0[tag], a1[tag]), tag
        assert np.array_equal(a0[tag], a0["single"]), tag
    np.testing.assert_allclose(a0["single"], want, rtol=1e-6)


def test_grad_averager_behind_the_side_stream_issues_every_chunk_once(monkeypatch):
    """engine._WgradLane.progress (the weight-gradient side stream's hand-off to the gradient exchange) with a mock stream:
    the exchange must be issued INSIDE the side-stream context, only after that stream has been made to wait for the main
    one (so the collective is ordered behind the kernels of both), pending weight-gradient GEMMs must have been flushed
    first, and over a whole backward + the optimizer hook every chunk of the arena goes out exactly once."""
    import torch
    from muscle_amd import engine, ops
    from muscle_amd import dist as mdist

    log = []

    class FakeStream:
        def __init__(self, name):
            self.name = name

        def wait_stream(self, other):
            log.append(("wait", self.name, other.name))

        def record_event(self):
            return None

    main, side = FakeStream("main"), FakeStream("side")
    active = [main]

    class _Ctx:
        def __init__(self, s):
            self.s = s

        def __enter__(self):
            active.append(self.s)

        def __exit__(self, *a):
            active.pop()

    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: active[-1])
    monkeypatch.setattr(torch.cuda, "stream", lambda s: _Ctx(s))
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(engine, "WGRAD_SIDE_STREAM", True)
    monkeypatch.setattr(engine, "WGRAD_CUS", 0)
    monkeypatch.setitem(engine._side_streams, (0, 0), side)
    monkeypatch.setattr(ops, "pw_wgrad", lambda G, X, dW, **kw: log.append(("wgrad", active[-1].name, dW)))

    n, chunk = 1000, 64
    launched = []

    class Sink:
        arena = torch.zeros(n)

    h = mdist.GradAverager(chunk_bytes=4 * chunk)
    h.world = 2
    monkeypatch.setattr(h, "_nccl", lambda: False)
    monkeypatch.setattr(mdist.dist, "all_reduce", lambda buf, **kw: launched.append((active[-1].name, buf.data_ptr(), buf.numel())))
    sink = Sink()

    lane = engine._WgradLane(torch.device("cpu"))
    assert lane.s is side
    # a backward that reports progress back to front, with weight-gradient GEMMs queued in between
    for step, lo in enumerate((900, 641, 640, 300, 130)):
        lane.wgrad("G", "X", f"dW{step}")
        before = len(log)
        lane.progress(lambda m, lo=lo: h.on_ready(sink, lo), None)
        new = log[before:]
        assert ("wgrad", "side", f"dW{step}") in new                     # flushed, on the side stream
        assert new[-1] == ("wait", "side", "main") or ("wait", "side", "main") in new
    assert all(where == "side" for where, _, _ in launched)             # early chunks: issued from the side stream
    early = len(launched)
    assert early == h.launched_early > 0
    lane.join()

    class Model:
        last_grad_sink = sink

    h(Model(), 1)                                                       # the hook before the optimizer: the rest, incl. chunk 0
    base = sink.arena.data_ptr()
    spans = sorted(((p - base) // 4, cnt) for _, p, cnt in launched)
    assert spans[0][0] == 0 and sum(c for _, c in spans) == n
    for (a, ca), (b, _) in zip(spans, spans[1:]):
        assert a + ca == b                                              # contiguous, no overlap: every element exactly once
    assert h.bytes_reduced == 4 * n


def test_infer_oracle_semantics_and_file_format(tmp_path):
    """infer_mcl.py:107-182 restated in the oracle: dict keys = positive labels, float32 [H,W] maps, the min-max rule
    with its 'below min + 1e-6 -> 0' quirk, un-flipping of odd passes, and the .npy dict layout evaluation.py reads."""
    from oracle import mcl_oracle as O
    from muscle_amd import synth, infer
    from muscle_amd.arch import net_cfg
    rng = np.random.default_rng(0)
    stack = [rng.standard_normal((20, 6, 5)).astype(np.float32) for _ in range(4)]
    n = O.infer_norm([s.copy() for s in stack])
    s = np.sum(stack, axis=0)
    s[s < 0] = 0
    assert n.dtype == np.float32 and np.isclose(n.max(), 1.0, atol=1e-5)
    for c in range(20):                                   # channel minimum is 0 here -> zeroed pixels sit at -1e-6/den
        assert np.isclose(n[c].min(), (0 - s[c].min() - 1e-6) / (s[c].max() - s[c].min() + 1e-6), rtol=1e-4)
    cfg = net_cfg("efficientnet-b0", False)
    net = O.OracleNet("efficientnet-b0", synth.synth_state_dict(cfg, 3))
    img = torch.from_numpy(synth.normal(3, "img", (1, 3, 40, 56)).astype(np.float32))
    label = torch.zeros(1, 20)
    label[0, [1, 19]] = 1
    cam, sgc, score = O.infer_cam(net, [img, torch.flip(img, dims=[3])], label, 33, 47)
    assert sorted(cam) == sorted(sgc) == [1, 19] and cam[1].shape == (33, 47) and cam[1].dtype == np.float32
    assert score.shape == (20,) and float(score.min()) > 0 and float(score.max()) < 1
    p = str(tmp_path / "a.npy")
    infer.save_cam_dict(p, sgc)
    back = infer.load_cam_dict(p)
    assert sorted(back) == [1, 19] and np.array_equal(back[19], sgc[19])


def test_eval_oracle_matches_hand_count():
    """src/evaluation.py:27-68 restated in the oracle, on a 2x3 image counted by hand."""
    from oracle import mcl_oracle as O
    pd = {i: np.zeros((2, 3), np.half) for i in range(20)}
    pd[0][:] = [[0.9, 0.1, 0.3], [0.3, 0.0, 0.6]]          # class 1
    pd[4][:] = [[0.2, 0.4, 0.3], [0.1, 0.0, 0.6]]          # class 5
    gt = np.array([[1, 5, 0], [255, 0, 1]], np.uint8)
    TP, P, T_ = O.eval_compare(pd, gt, 0.3)
    # fp16(0.3) = 0.30005 > 0.3, so at (0,2) class 1 beats the threshold channel and (first maximum) class 5:
    # predict = [[1,5,1],[x,0,1]]
    assert (TP[0], TP[1], TP[5]) == (1, 2, 1) and (P[0], P[1], P[5]) == (1, 3, 1) and (T_[0], T_[1], T_[5]) == (2, 2, 1)
    m, per = O.eval_miou(TP, P, T_)
    assert abs(per[0] - 50.0) < 1e-6 and abs(per[1] - 200.0 / 3) < 1e-6 and abs(per[5] - 100.0) < 1e-6
    assert abs(m - (50.0 + 200.0 / 3 + 100.0) / 21) < 1e-6


def test_irn_search_paths_table():
    """The path table the HIP affinity kernel consumes (muscle_amd.indexing.search_paths) lists exactly the oracle's /
    the reference's search paths (src/indexing.py:13-47): same directions, same pixels, farthest pixel first."""
    from oracle import mcl_oracle as O
    from muscle_amd import indexing
    for radius in (3, 5):
        mine = [tuple(map(tuple, p)) for p in indexing.search_paths(radius)]
        ref = [tuple(map(tuple, p)) for g in O.irn_search_paths(radius) for p in g]
        assert mine == ref and len(mine) == len(set(p[0] for p in mine))      # one path per destination
        for p in mine:
            assert p[-1] == (0, 0) and abs(p[0][0]) + abs(p[0][1]) == max(abs(y) + abs(x) for y, x in p)
    assert len(indexing.search_paths(5)) == 34      # 4 + 9 + 9 + 7 + 5 directions inside the radius-5 half disc


def test_build_refuses_lab_switches(monkeypatch):
    """Timing-only -DMX_LAB_* switches (they change results) must never reach the shipped library through MUSCLE_EXTRA_FLAGS."""
    from muscle_amd import _build
    monkeypatch.setenv("MUSCLE_EXTRA_FLAGS", "-DMX_LAB_NOSPLIT")
    with pytest.raises(RuntimeError, match="lab-only"):
        _build.build(verbose=False)
    src = "".join(open(os.path.join(ROOT, "muscle_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "muscle_amd", "csrc")))
    assert "MX_LAB_" not in src
