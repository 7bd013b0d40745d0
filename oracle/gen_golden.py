"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    python oracle/gen_golden.py            # needs /root/reference; never runs on the GPU box

The reference is pure Python on torch; it is imported from /root/reference and executed on
CPU.  Nothing from it is copied: the fixtures hold inputs' seeds and the reference's
*outputs* only (weights and batches are regenerated bit-identically from
muscle_amd.synth on either side).

Accommodations, all confined to this script:
  1. `import src` pulls in third-party modules that are not installed here and that no
     hot-path function touches (torchvision, cv2, qpth, imageio, skimage); empty placeholder
     modules are registered for those names before the import.
  2. MuSCLe.__init__ unconditionally fetches ImageNet weights from a URL
     (src/MuSCLe.py:165 -> efficientnet_pytorch/utils.py:326); the name
     `load_pretrained_weights` is rebound to a no-op and synthetic weights are loaded with
     load_state_dict(strict=True).
  3. The reference MuSCLe has no EfficientNet-B0 table (src/MuSCLe.py:167-178): for B0 a B1
     model is built (same widths), its backbone replaced by EfficientNet.from_name('b0') and
     p*_seq set by the same last-block-of-stage rule.
  4. train_mcl.py cannot be imported (top-level CUDA/VOC/tensorboard dependencies); its
     helper functions (lines 21-36) and its loop body (lines 154-229) are taken from the
     file's AST and executed unmodified in a namespace that supplies model / optimizer /
     criteria / the batch.
"""
from __future__ import annotations

import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("MUSCLE_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from muscle_amd import synth  # noqa: E402
from muscle_amd.arch import net_cfg  # noqa: E402


# ---------------------------------------------------------------------------
# reference loading
# ---------------------------------------------------------------------------
def load_reference():
    class _Absent:
        def __init__(self, *a, **k):
            pass

    def placeholder(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = placeholder("torchvision")
    tv.transforms = placeholder("torchvision.transforms", Compose=_Absent, ColorJitter=_Absent,
                                RandomErasing=_Absent)
    tv.transforms.functional = placeholder("torchvision.transforms.functional")
    placeholder("cv2")
    placeholder("qpth").qp = placeholder("qpth.qp", QPFunction=_Absent)
    placeholder("imageio")
    placeholder("skimage").transform = placeholder("skimage.transform", resize=None)
    sys.path.insert(0, REF)
    import src  # noqa
    import src.efficientnet_pytorch.model as ref_model
    ref_model.load_pretrained_weights = lambda *a, **k: None
    return src


def train_script_ast():
    with open(os.path.join(REF, "train_mcl.py")) as f:
        return ast.parse(f.read())


def script_functions(tree, names):
    ns = {"torch": torch}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), "train_mcl.py", "exec"), ns)
    return {n: ns[n] for n in names}


def loop_body(tree):
    """Statements of `for iter, pack in enumerate(train_data_loader)` up to line 229."""
    main = [n for n in tree.body if isinstance(n, ast.If)][-1]
    ep_loop = [n for n in main.body if isinstance(n, ast.For) and getattr(n.target, "id", "") == "ep"][0]
    it_loop = ep_loop.body[0]
    stmts = [s for s in it_loop.body if s.end_lineno <= 229]
    return compile(ast.Module(stmts, []), "train_mcl.py", "exec")


def build_model(src, name, sd_np, last_pooling=False):
    import src.efficientnet_pytorch as effnet
    if name == "efficientnet-b0":
        m = src.MuSCLe(num_classes=21, pretrained="efficientnet-b1", layers=3, MemoryEfficient=True,
                       last_pooling=last_pooling)
        m.backbone = effnet.EfficientNet.from_name("efficientnet-b0", override_params={"num_classes": 21},
                                                   last_pooling=last_pooling)
        (m.p1_seq, m.p2_seq, m.p3_seq, m.p4_seq, m.p5_seq, m.p6_seq, m.p7_seq) = net_cfg(name, last_pooling).taps
    else:
        m = src.MuSCLe(num_classes=21, pretrained=name, layers=3, MemoryEfficient=True, last_pooling=last_pooling)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    return m


# ---------------------------------------------------------------------------
# summaries (shared with the tests through tests/golden_util.py conventions)
# ---------------------------------------------------------------------------
def tensor_summary(named, seed=123):
    """[l2, probe-dot] per tensor; probe = synth.normal(seed, key)."""
    rows = []
    for k, v in named:
        if v is None:
            rows.append([np.nan, np.nan])
            continue
        a = v.detach().double().numpy().ravel()
        pr = synth.normal(seed, k, a.shape)
        rows.append([float(np.sqrt((a * a).sum())), float((a * pr).sum())])
    return np.array(rows, dtype=np.float64)


def bn_summary(model):
    rows = []
    for k, v in model.state_dict().items():
        if k.endswith("running_mean"):
            rv = model.state_dict()[k.replace("running_mean", "running_var")]
            rows.append([float(v.double().sum()), float(rv.double().sum())])
    return np.array(rows, dtype=np.float64)


def calibrate_bn(model, x):
    """One train-mode pass with momentum 1.0 so running stats equal batch stats
    (eval-mode forward of a random net is degenerate otherwise)."""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    old = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    torch.manual_seed(7)
    with torch.no_grad():
        model(x, cam="pix")
    for m, o in zip(bns, old):
        m.momentum = o


# ---------------------------------------------------------------------------
# step goldens
# ---------------------------------------------------------------------------
class RecordingAdam:
    """Proxy handed to the reference loop body in place of `optimizer`; records gradient and
    parameter summaries around each real step."""

    def __init__(self, model, opt):
        self.model, self.opt, self.records = model, opt, []
        self.param_groups = opt.param_groups

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self):
        named = list(self.model.named_parameters())
        before = {k: p.detach().clone() for k, p in named}
        g = tensor_summary([(k, p.grad) for k, p in named])
        self.opt.step()
        d = tensor_summary([(k, p.detach() - before[k]) for k, p in named])
        self.records.append((g, d))


def gen_step(src, tree, name, n, size, view, ep, seed, fname, torch_seed=11, lr=1e-4, cam_stride=4):
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    batch = synth.synth_batch(n, size, view, seed)
    model = build_model(src, name, sd)
    if ep >= 8:
        calibrate_bn(model, torch.from_numpy(batch["view1"]))
    fns = script_functions(tree, ["cam_maxnorm", "cam_softmaxnorm"])
    opt = torch.optim.Adam(params=model.parameters(), lr=lr, weight_decay=5e-5)
    rec = RecordingAdam(model, opt)

    # drop_connect draws: replay the generator to record them (utils.py:88 order)
    torch.manual_seed(torch_seed)
    drop_u = [torch.rand([n, 1, 1, 1]).view(-1).numpy().copy() for b in cfg.blocks if b.skip and b.drop_rate]
    drop_idx = [b.index for b in cfg.blocks if b.skip and b.drop_rate]

    # crop geometry draws: wrap the reference's np.random.randint to log them
    import src.torchutils as ref_tu
    geom_log = []
    real_randint = np.random.randint

    def logging_randint(*a, **k):
        v = real_randint(*a, **k)
        geom_log.append(int(v))
        return v

    ns = dict(fns)
    ns.update(torch=torch, F=torch.nn.functional, np=np, model=model, optimizer=rec, ep=ep,
              criterion1=src.FocalLoss(), criterion2=src.Log_Sum_Exp_Pairwise_Loss,
              criterion3=torch.nn.MultiLabelSoftMarginLoss(), criterion4=src.EMD(),
              criterion5=src.image_level_contrast, PixPro=src.PixPro, torchutils=ref_tu,
              pack=("name", torch.from_numpy(batch["img"]), torch.from_numpy(batch["label"]),
                    torch.from_numpy(batch["view1"]).double(), torch.from_numpy(batch["view2"]).double(),
                    torch.from_numpy(batch["coord1"]), torch.from_numpy(batch["coord2"]), None))
    # train_mcl.py:161-165 casts to float only under cuda; replicate the cast for CPU
    ns["pack"] = tuple(t.float() if torch.is_tensor(t) and t.dtype == torch.float64 else t for t in ns["pack"])
    np.random.seed(5)
    np.random.randint = logging_randint
    torch.manual_seed(torch_seed)
    try:
        exec(loop_body(tree), ns)
    finally:
        np.random.randint = real_randint

    def f(x):
        return float(x.detach()) if torch.is_tensor(x) else float(x)

    out = {
        "meta": np.array([n, size, view, ep, seed, torch_seed], dtype=np.int64),
        "name": np.array(name),
        "lr": np.array(lr),
        "drop_idx": np.array(drop_idx, dtype=np.int64),
        "drop_u": np.array(drop_u, dtype=np.float32).reshape(len(drop_idx), n),
        "crop_draws": np.array(geom_log, dtype=np.int64),
        "losses": np.array([f(ns[k]) for k in ("loss_focal", "loss_softmargin", "loss_pair", "loss_er",
                                                "loss_imc", "loss_pixpro", "loss_emd")], dtype=np.float64),
        "loss_is_tensor": np.array([torch.is_tensor(ns[k]) for k in
                                    ("loss_imc", "loss_pixpro", "loss_emd")]),
        "emb": ns["emb"].detach().numpy(),
        "logits": ns["logits"].detach().numpy(),
        "cam_stride": np.array(cam_stride, dtype=np.int64),
        "raw_cams_s4": ns["raw_cams"].detach().numpy()[:, :, ::cam_stride, ::cam_stride].copy(),
        "raw_sgcs_s4": ns["raw_sgcs"].detach().numpy()[:, :, ::cam_stride, ::cam_stride].copy(),
        "raw_cams_stats": np.array([float(ns["raw_cams"].double().sum()), float(ns["raw_cams"].double().pow(2).sum())]),
        "raw_sgcs_stats": np.array([float(ns["raw_sgcs"].double().sum()), float(ns["raw_sgcs"].double().pow(2).sum())]),
        "param_keys": np.array([k for k, _ in model.named_parameters()]),
        "bn_after": bn_summary(model),
    }
    for i, (g, d) in enumerate(rec.records):
        out[f"grad{i + 1}"] = g
        out[f"delta{i + 1}"] = d
    if ep >= 8:
        out["sgcs_vw1_s4"] = ns["sgcs_vw1"].detach().numpy()[:, :, ::cam_stride, ::cam_stride].copy()
        out["cams_vw2_s4"] = ns["cams_vw2"].detach().numpy()[:, :, ::cam_stride, ::cam_stride].copy()
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "losses", out["losses"], "steps", len(rec.records), "crop draws", len(geom_log))


# ---------------------------------------------------------------------------
# unit goldens (one per §8(a) row that is a separable function)
# ---------------------------------------------------------------------------
def gen_units(src, tree, fname="units.npz", seed=3):
    import src.efficientnet_pytorch.model as ref_model
    import src.efficientnet_pytorch.utils as ref_utils
    import src.torchutils as ref_tu
    fns = script_functions(tree, ["cam_maxnorm", "cam_softmaxnorm"])
    out = {}
    T = lambda a: torch.from_numpy(np.asarray(a))  # noqa: E731

    # a4 swish fwd/bwd
    x = T(synth.normal(seed, "swish.x", (257,)).astype(np.float32) * 3).requires_grad_()
    y = ref_utils.MemoryEfficientSwish()(x)
    y.backward(T(synth.normal(seed, "swish.g", (257,)).astype(np.float32)))
    out["swish_y"], out["swish_dx"] = y.detach().numpy(), x.grad.numpy()

    # a2/a3/a5: MBConv blocks of B7 geometry (k3s1 e1, k3s2 e6, k5s1 e6 skip+drop, k5s2) at small spatial size
    cfg7 = net_cfg("efficientnet-b7", False)
    for bi, hw in ((0, 12), (1, 12), (4, 13), (11, 13), (12, 10), (38, 6)):
        b = cfg7.blocks[bi]
        ba = ref_utils.BlockArgs(kernel_size=b.kernel, num_repeat=1, input_filters=b.cin, output_filters=b.cout,
                                 expand_ratio=(b.cexp // b.cin), id_skip=True,
                                 # first block of a stage keeps the decoder's list stride (utils.py:205), repeats get
                                 # the int 1 (model.py:148) -- only the latter satisfies `stride == 1` at model.py:90
                                 stride=(b.stride if b.skip else [b.stride]), se_ratio=0.25)
        gp = ref_utils.GlobalParams(batch_norm_momentum=0.99, batch_norm_epsilon=1e-3, image_size=600)
        blk = ref_model.MBConvBlock(ba, gp)
        keys = [k for k in synth.state_dict_spec(cfg7) if k.startswith(f"backbone._blocks.{bi}.")]
        sd = synth.synth_state_dict(cfg7, seed)
        blk.load_state_dict({k.split(f"_blocks.{bi}.")[1]: T(sd[k]) for k in keys}, strict=True)
        blk.train()
        n = 3
        xin = T(synth.normal(seed, f"mb{bi}.x", (n, b.cin, hw, hw)).astype(np.float32)).requires_grad_()
        torch.manual_seed(21)
        u = torch.rand([n, 1, 1, 1]).view(-1).numpy().copy()
        torch.manual_seed(21)
        yb = blk(xin, drop_connect_rate=b.drop_rate)
        gy = T(synth.normal(seed, f"mb{bi}.g", tuple(yb.shape)).astype(np.float32))
        yb.backward(gy)
        out[f"mb{bi}_u"] = u
        out[f"mb{bi}_y"] = yb.detach().numpy()
        out[f"mb{bi}_dx"] = xin.grad.numpy()
        out[f"mb{bi}_dw"] = tensor_summary([(f"backbone._blocks.{bi}." + k, p.grad) for k, p in blk.named_parameters()])
        out[f"mb{bi}_bn"] = np.array([[float(m.running_mean.sum()), float(m.running_var.sum())]
                                      for m in blk.modules() if isinstance(m, torch.nn.BatchNorm2d)])

    # a8/a14 CAM normalisations
    cam = T(synth.normal(seed, "cam", (2, 21, 9, 11)).astype(np.float32))
    out["cam_softmaxnorm"] = fns["cam_softmaxnorm"](cam).numpy()
    out["cam_maxnorm"] = fns["cam_maxnorm"](cam).numpy()

    # a10-a12 classification losses
    lab = T(synth.synth_labels(6, seed))
    logit = T(synth.normal(seed, "logit", (6, 20)).astype(np.float32) * 2).requires_grad_()
    p = torch.sigmoid(logit)
    l1 = src.FocalLoss()(p, lab)
    l2 = torch.nn.MultiLabelSoftMarginLoss()(logit, lab)
    l3 = src.Log_Sum_Exp_Pairwise_Loss(p, lab)
    (l1 + l2 + l3.mean()).backward()
    out["cls_losses"] = np.array([float(l1), float(l2)])
    out["cls_pair"] = l3.detach().numpy()
    out["cls_dlogit"] = logit.grad.numpy()

    # a13 IMC: regular, and the all-skip case that returns Python 0.0
    emb = T(synth.normal(seed, "imc.emb", (8, 48)).astype(np.float32)).requires_grad_()
    lab8 = T(synth.synth_labels(8, seed + 1))
    li = src.image_level_contrast(emb, lab8)
    out["imc_is_tensor"] = np.array(torch.is_tensor(li))
    if torch.is_tensor(li):
        li.backward()
        out["imc_demb"] = emb.grad.numpy()
    out["imc"] = np.array(float(li))
    same = torch.ones(4, 20)
    l0 = src.image_level_contrast(T(synth.normal(seed, "imc.emb0", (4, 48)).astype(np.float32)), same)
    out["imc_degenerate"] = np.array([float(l0), float(torch.is_tensor(l0))])

    # a9 ER expression (train_mcl.py:185-188) on small maps
    a = T(synth.uniform(seed, "er.a", (3, 21, 8, 8)).astype(np.float32))
    s = T(synth.uniform(seed, "er.s", (3, 21, 8, 8)).astype(np.float32)).requires_grad_()
    lab3 = T(synth.synth_labels(3, seed + 2))
    lwb = torch.cat((torch.ones(3, 1), lab3), dim=1)
    vc = int(lab3.sum())
    ca, sg = a * lwb.unsqueeze(2).unsqueeze(3), s * lwb.unsqueeze(2).unsqueeze(3)
    ler = torch.topk(torch.flatten(torch.abs(ca.detach() - sg), start_dim=1), k=int(0.2 * vc * 8 * 8), dim=-1)[0].mean()
    ler.backward()
    out["er"], out["er_ds"] = np.array(float(ler)), s.grad.numpy()

    # a15 PixPro
    f1 = T(synth.uniform(seed, "pp.1", (3, 21, 20, 20)).astype(np.float32)).requires_grad_()
    f2 = T(synth.uniform(seed, "pp.2", (3, 21, 20, 20)).astype(np.float32))
    f2[0, :, 3:5, 3:5] = 0     # exercise the eps clamp of cosine_similarity
    c1, c2, _ = synth.synth_coords(3, 20, 40, seed)
    lp = src.PixPro(f1, f2, T(c1), T(c2))
    lp.backward()
    out["pixpro"], out["pixpro_d1"] = np.array(float(lp)), f1.grad.numpy()

    # a16 dynamic crops + a17 EMD
    v = 64
    x1 = torch.nn.functional.normalize(T(synth.uniform(seed, "dc.1", (3, 21, v, v)).astype(np.float32)), dim=1).requires_grad_()
    x2 = torch.nn.functional.normalize(T(synth.uniform(seed, "dc.2", (3, 21, v, v)).astype(np.float32)), dim=1)
    c1, c2, _ = synth.synth_coords(3, v, 2 * v, seed + 4)
    c1[2] = (0, 0, 10, 40)
    c2[2] = (5, 3, 10, 40)       # h < 15 -> skipped sample (torchutils.py:240)
    draws = []
    real = np.random.randint

    def lr(*a_, **k_):
        r = real(*a_, **k_)
        draws.append(int(r))
        return r

    np.random.seed(9)
    np.random.randint = lr
    try:
        cr1, cr2, bidx = ref_tu.get_dynamic_crops(x1, T(c1), x2, T(c2))
    finally:
        np.random.randint = real
    out["dc_coord1"], out["dc_coord2"] = c1, c2
    out["dc_draws"] = np.array(draws, dtype=np.int64)
    out["dc_bidx"] = np.array(bidx, dtype=np.int64)
    out["dc_shapes1"] = np.array([[i, *c.shape[2:]] for i, bc in enumerate(cr1) for c in bc], dtype=np.int64)
    out["dc_shapes2"] = np.array([[i, *c.shape[2:]] for i, bc in enumerate(cr2) for c in bc], dtype=np.int64)
    out["dc_sums1"] = np.array([float(c.double().sum()) for bc in cr1 for c in bc])
    out["dc_sums2"] = np.array([float(c.double().sum()) for bc in cr2 for c in bc])
    le = src.EMD()(cr1, cr2, mode="dynamic")
    le.backward()
    out["emd"], out["emd_dx1"] = np.array(float(le)), x1.grad.numpy()

    # a18 Adam (torch.optim.Adam with weight_decay): three steps on a small vector, one skipped grad
    w = T(synth.normal(seed, "adam.w", (33,)).astype(np.float32)).requires_grad_()
    w2 = T(synth.normal(seed, "adam.w2", (5,)).astype(np.float32)).requires_grad_()
    o = torch.optim.Adam([w, w2], lr=1e-4, weight_decay=5e-5)
    traj = []
    for stp in range(3):
        o.zero_grad()
        w.grad = T(synth.normal(seed, f"adam.g{stp}", (33,)).astype(np.float32))
        w2.grad = None if stp == 1 else T(synth.normal(seed, f"adam.h{stp}", (5,)).astype(np.float32))
        o.step()
        traj.append(np.concatenate([w.detach().numpy(), w2.detach().numpy()]))
    out["adam_traj"] = np.array(traj)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, {k: np.asarray(v).shape for k, v in out.items() if not k.startswith("mb")})


def gen_forward(src, name, n, size, seed, fname):
    """a1/a6/a7: MuSCLe.forward(cam='cam') in train mode and (after BN calibration) eval 'pix'."""
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    model = build_model(src, name, sd)
    x = torch.from_numpy(synth.normal(seed, "fwd.x", (n, 3, size, size)).astype(np.float32))
    torch.manual_seed(31)
    drop_u = [torch.rand([n, 1, 1, 1]).view(-1).numpy().copy() for b in cfg.blocks if b.skip and b.drop_rate]
    model.train()
    torch.manual_seed(31)
    feats = model.backbone(x)
    model2 = build_model(src, name, sd)
    model2.train()
    torch.manual_seed(31)
    cams, sgc, emb, logits = model2(x, cam="cam")
    calibrate_bn(model2, x)
    model2.eval()
    with torch.no_grad():
        cams_e, sgc_e = model2(x, cam="pix")
    out = {
        "meta": np.array([n, size, seed], dtype=np.int64), "name": np.array(name),
        "drop_idx": np.array([b.index for b in cfg.blocks if b.skip and b.drop_rate], dtype=np.int64),
        "drop_u": np.array(drop_u, dtype=np.float32),
        "feat_stats": np.array([[float(f.double().sum()), float(f.double().pow(2).sum())] for f in feats]),
        "feat_shapes": np.array([list(f.shape) for f in feats], dtype=np.int64),
        "p7": feats[cfg.taps[6]].detach().numpy(),
        "cams_s4": cams.detach().numpy()[:, :, ::4, ::4].copy(), "sgc_s4": sgc.detach().numpy()[:, :, ::4, ::4].copy(),
        "cams_stats": np.array([float(cams.double().sum()), float(cams.double().pow(2).sum())]),
        "sgc_stats": np.array([float(sgc.double().sum()), float(sgc.double().pow(2).sum())]),
        "emb": emb.detach().numpy(), "logits": logits.detach().numpy(),
        "cams_eval_s4": cams_e.numpy()[:, :, ::4, ::4].copy(), "sgc_eval_s4": sgc_e.numpy()[:, :, ::4, ::4].copy(),
        "bn_calibrated": bn_summary(model2),
    }
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "emb", out["emb"].shape, "cams max", float(cams.max()), "eval cams max", float(cams_e.max()))


# ---------------------------------------------------------------------------
# config 4: decoder mode + BEACON FieldLoss (train_muscle.py:171-203)
# ---------------------------------------------------------------------------
def muscle_loop_body():
    with open(os.path.join(REF, "train_muscle.py")) as f:
        tree = ast.parse(f.read())
    main = [n for n in tree.body if isinstance(n, ast.If)][-1]
    ep_loop = [n for n in main.body if isinstance(n, ast.For) and getattr(n.target, "id", "") == "ep"][0]
    it_loop = [n for n in ep_loop.body if isinstance(n, ast.For)][0]
    stmts = [s_ for s_ in it_loop.body if s_.end_lineno <= 203]
    return compile(ast.Module(stmts, []), "train_muscle.py", "exec")


def smooth_field(seed, name, shape, passes=3):
    """Deterministic smooth random maps (box-blurred noise) so class boundaries are clean curves."""
    x = torch.from_numpy(synth.normal(seed, name, shape).astype(np.float32))
    k = torch.ones(shape[1], 1, 5, 5) / 25.0
    for _ in range(passes):
        x = torch.nn.functional.conv2d(x, k, padding=2, groups=shape[1])
    return (x / x.std()).contiguous()


def gen_field_units(src, fname="units_field.npz", seed=4):
    import random
    import src.edge as ref_edge
    n, c, ch, hw, kk, step = 2, 21, 16, 64, 8, 3
    seg = smooth_field(seed, "fl.seg", (n, c, hw, hw)) * 0.05
    ft = smooth_field(seed, "fl.ft", (n, ch, hw, hw), 1).requires_grad_()
    lab = synth.synth_labels(n, seed)
    lab[:, :] = 0
    lab[0, [2, 7]] = 1
    lab[1, [7, 11, 14]] = 1
    mask = torch.from_numpy(synth.synth_soft_mask(lab, hw, seed))
    lwb = torch.cat((torch.ones(n, 1), torch.from_numpy(lab)), 1)
    crit = ref_edge.FieldLoss(sobel_size=5, beta=1e2, k=kk)
    random.seed(77)
    loss, edge_fg = crit(seg, ft, mask, lwb, step)
    out = {"meta": np.array([n, c, ch, hw, kk, step, seed]), "label": lab,
           "is_tensor": np.array(torch.is_tensor(loss)), "edge_fg": edge_fg.numpy()}
    if torch.is_tensor(loss):
        loss.backward()
        out["loss"], out["dft"] = np.array(float(loss)), ft.grad.numpy()
    # degenerate: no foreground label -> no boundary pixel survives the label mask -> returns False (edge.py:376-378)
    l0, _ = crit(seg, ft.detach(), mask, torch.cat((torch.ones(n, 1), torch.zeros(n, c - 1)), 1), step)
    out["nolabel_returns_false"] = np.array(l0 is False)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "loss", out.get("loss"), "grad nnz", int((out.get("dft", np.zeros(1)) != 0).sum()))


def gen_muscle_step(src, name, n, size, seed, fname, lamb, k, step, lr=1e-5, torch_seed=13):
    import random
    import src.edge as ref_edge
    cfg = net_cfg(name, True)
    sd = synth.synth_state_dict(cfg, seed, mode="dec", layers=3)
    model = src.MuSCLe(num_classes=21, pretrained=name, layers=3, MemoryEfficient=True, last_pooling=True, mode="dec")
    model.load_state_dict({k_: torch.from_numpy(np.asarray(v)) for k_, v in sd.items()}, strict=True)
    lab = synth.synth_labels(n, seed)
    img = torch.from_numpy(synth.normal(seed, "img", (n, 3, size, size)).astype(np.float32))
    mask = torch.from_numpy(synth.synth_soft_mask(lab, size, seed))
    opt = torch.optim.Adam(params=model.parameters(), lr=lr, weight_decay=1e-5)
    rec = RecordingAdam(model, opt)
    torch.manual_seed(torch_seed)
    drop_u = [torch.rand([n, 1, 1, 1]).view(-1).numpy().copy() for b in cfg.blocks if b.skip and b.drop_rate]
    drop_idx = [b.index for b in cfg.blocks if b.skip and b.drop_rate]

    class Args:
        pass
    args = Args()
    args.lamb, args.step = lamb, step
    ns = dict(torch=torch, model=model, optimizer=rec, args=args, criterion1=torch.nn.CrossEntropyLoss(),
              criterion2=ref_edge.FieldLoss(sobel_size=5, beta=1e2, k=k),
              pack=("name", img, torch.from_numpy(lab), mask))
    model.train()
    random.seed(78)
    torch.manual_seed(torch_seed)
    # Which boundary points enter the BEACON term depends on thresholding a softmax(100*seg) edge map: 1e-6 of round-off in
    # seg moves pixels in and out of the point lists, so two fp32 implementations draw different samples from the same
    # random stream.  The reference's own choices are recorded (in_out_div's lists through random.sample's picks) so that
    # the loss and its gradient can be compared on identical points.
    div_log, smp_log = [], []
    real_div, real_sample = ref_edge.FieldLoss.in_out_div, random.sample

    def logging_div(self, *a_, **k_):
        o, i = real_div(self, *a_, **k_)
        div_log.append((o.clone(), i.clone()))
        return o, i

    def logging_sample(pop, kk):
        r = real_sample(pop, kk)
        smp_log.append(list(r))
        return r
    ref_edge.FieldLoss.in_out_div = logging_div
    random.sample = logging_sample
    clip_log = []
    real_clip = torch.nn.utils.clip_grad_norm_

    def logging_clip(params, max_norm, norm_type=2):
        tn = real_clip(params, max_norm, norm_type=norm_type)
        clip_log.append(float(tn))
        return tn
    torch.nn.utils.clip_grad_norm_ = logging_clip
    try:
        exec(muscle_loop_body(), ns)
    finally:
        torch.nn.utils.clip_grad_norm_ = real_clip
        ref_edge.FieldLoss.in_out_div = real_div
        random.sample = real_sample
    # reassemble the sampled points per qualifying (sample, class) slot, in the loss's own order (edge.py:279-299)
    lab_t = torch.from_numpy(lab)
    slots = [(b_, c_) for b_ in range(n) for c_ in range(20) if lab_t[b_, c_] != 0]
    rp_b, rp_out, rp_in, j = [], [], [], 0
    if lamb > 0:
        assert len(div_log) == len(slots), (len(div_log), len(slots))
        for (b_, c_), (o_, i_) in zip(slots, div_log):
            if i_.numel() > k and o_.numel() > k:
                r_out, r_in = smp_log[j], smp_log[j + 1]
                j += 2
                rp_b.append(b_)
                rp_out.append(o_[r_out].numpy())
                rp_in.append(i_[r_in].numpy())
        assert j == len(smp_log)
    out = {
        "replay_b": np.array(rp_b, dtype=np.int64), "replay_out": np.array(rp_out, dtype=np.int64).reshape(len(rp_b), k),
        "replay_in": np.array(rp_in, dtype=np.int64).reshape(len(rp_b), k),
        "meta": np.array([n, size, seed, torch_seed, k, step], dtype=np.int64), "name": np.array(name),
        "lamb": np.array(lamb), "lr": np.array(lr),
        "drop_idx": np.array(drop_idx, dtype=np.int64), "drop_u": np.array(drop_u, dtype=np.float32).reshape(len(drop_idx), n),
        "losses": np.array([float(ns["l1"]), float(ns["l2"])]), "l2_is_tensor": np.array(torch.is_tensor(ns["l2"])),
        "grad_norm": np.array(clip_log),
        "seg_map_s4": ns["seg_map"].detach().numpy()[:, :, ::4, ::4].copy() if "seg_map" in ns else np.zeros(1),
        "param_keys": np.array([k_ for k_, _ in model.named_parameters()]),
        "grad1": rec.records[0][0], "delta1": rec.records[0][1], "bn_after": bn_summary(model),
    }
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "losses", out["losses"], "grad norm", clip_log)


def gen_seg_forward(src, name, n, size, seed, fname):
    cfg = net_cfg(name, True)
    sd = synth.synth_state_dict(cfg, seed, mode="dec", layers=3)
    model = src.MuSCLe(num_classes=21, pretrained=name, layers=3, MemoryEfficient=True, last_pooling=True, mode="dec")
    model.load_state_dict({k_: torch.from_numpy(np.asarray(v)) for k_, v in sd.items()}, strict=True)
    x = torch.from_numpy(synth.normal(seed, "fwd.x", (n, 3, size, size)).astype(np.float32))
    torch.manual_seed(31)
    drop_u = [torch.rand([n, 1, 1, 1]).view(-1).numpy().copy() for b in cfg.blocks if b.skip and b.drop_rate]
    model.train()
    torch.manual_seed(31)
    seg, ft = model(x, cam="seg")
    out = {"meta": np.array([n, size, seed], dtype=np.int64), "name": np.array(name),
           "drop_idx": np.array([b.index for b in cfg.blocks if b.skip and b.drop_rate], dtype=np.int64),
           "drop_u": np.array(drop_u, dtype=np.float32),
           "seg_s4": seg.detach().numpy()[:, :, ::4, ::4].copy(), "ft_s8": ft.detach().numpy()[:, :, ::8, ::8].copy(),
           "seg_stats": np.array([float(seg.double().sum()), float(seg.double().pow(2).sum())]),
           "ft_stats": np.array([float(ft.double().sum()), float(ft.double().pow(2).sum())]),
           "bn_after": bn_summary(model)}
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "seg absmax", float(seg.abs().max()))


def gen_irn_units(src, fname="irn_rw.npz"):
    """src/indexing.py::propagate_to_edge (the IRN random walk, infer_irn.py:76) run as is on CPU: its three hard-wired
    `.cuda()` calls (indexing.py:84, :110) are made no-ops for the duration of the call, nothing else is touched."""
    import src.indexing as RI
    out = {}
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for tag, (h, w, radius, beta, times, seed) in {"a": (13, 17, 5, 10, 8, 0), "b": (9, 22, 5, 8, 4, 1), "c": (16, 12, 3, 10, 6, 2)}.items():
            x = torch.from_numpy(synth.uniform(seed, f"irn_x_{tag}", (1, 20, h, w)).astype(np.float32))
            edge = torch.from_numpy(synth.uniform(seed, f"irn_e_{tag}", (1, h, w)).astype(np.float32)) ** 2
            rw = RI.propagate_to_edge(x.clone(), edge.clone(), radius=radius, beta=beta, exp_times=times)
            out[f"{tag}_x"], out[f"{tag}_edge"], out[f"{tag}_rw"] = x.numpy(), edge.numpy(), rw.numpy()
            out[f"{tag}_params"] = np.array([radius, beta, times], np.int64)
    finally:
        torch.Tensor.cuda = orig
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_eval_units(src, fname="eval_rapid.npz"):
    """src/evaluation.py::do_python_eval run as is (its 8 worker processes, files and all) on synthetic prediction dicts and
    ground-truth pngs written to a temporary directory; the fixture keeps the inputs and the returned loglists."""
    import tempfile
    from PIL import Image
    from src.evaluation import do_python_eval
    rng = np.random.default_rng(12)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        pred_dir, gt_dir = os.path.join(td, "pred"), os.path.join(td, "gt")
        os.makedirs(pred_dir); os.makedirs(gt_dir)
        names = []
        for i in range(5):
            H, W = 30 + 7 * i, 41 + 5 * i
            present = sorted(rng.choice(20, 3, replace=False).tolist())
            d = {c: (rng.random((H, W)).astype(np.float32) * (1.0 if c in present else 0.0)).astype(np.half) for c in range(20)}
            gt = rng.integers(0, 21, size=(H // 6 + 1, W // 6 + 1)).astype(np.uint8)
            gt = np.kron(gt, np.ones((6, 6), np.uint8))[:H, :W].copy()
            gt[rng.random((H, W)) < 0.07] = 255
            name = f"img{i}"
            names.append(name)
            np.save(os.path.join(pred_dir, name + ".npy"), d)
            Image.fromarray(gt).save(os.path.join(gt_dir, name + ".png"))
            out[f"pred{i}"] = np.stack([d[c] for c in range(20)])
            out[f"gt{i}"] = gt
        thr = [t / 100.0 for t in range(20, 52, 2)]
        rows = []
        for t in thr:
            ll = do_python_eval(pred_dir, gt_dir, names, 21, "npy", t, printlog=False)
            rows.append([ll[k] for k in list(ll.keys())])          # 21 category IoUs (percent) + mIoU
        out["thresholds"] = np.array(thr, np.float64)
        out["loglists"] = np.array(rows, np.float64)
    np.savez_compressed(os.path.join(OUT, fname), **out)


def main_config4(src):
    gen_field_units(src)
    gen_seg_forward(src, "efficientnet-b3", 2, 96, 21, "seg_forward_b3.npz")
    gen_muscle_step(src, "efficientnet-b3", 2, 96, 22, "muscle_step_b3_ce.npz", lamb=0.0, k=8, step=3)
    gen_muscle_step(src, "efficientnet-b3", 2, 128, 23, "muscle_step_b3_beacon.npz", lamb=0.05, k=8, step=3)


def gen_forward_eval(src, name, n, H, W, seed, fname, stride=16):
    """config 5 (infer_mcl.py:107-125): eval-mode forwards of a BN-calibrated model at a non-square, odd size.
    'cam' (what infer_mcl.py calls), 'pix', 'logits'."""
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    model = build_model(src, name, sd)
    x = torch.from_numpy(synth.normal(seed, "fwd.x", (n, 3, H, W)).astype(np.float32))
    calibrate_bn(model, x)
    model.eval()
    with torch.no_grad():
        cams, sgc, emb, logits = model(x, cam="cam")
        cams_p, sgc_p = model(x, cam="pix")
        emb_l, logits_l = model(x, cam="logits")
    assert torch.equal(cams, cams_p) and torch.equal(emb, emb_l)

    def st(t):
        return np.array([float(t.double().sum()), float(t.double().pow(2).sum()), float(t.abs().max())])
    out = {"meta": np.array([n, H, W, seed, stride], dtype=np.int64), "name": np.array(name),
           "cams_s": cams.numpy()[:, :, ::stride, ::stride].copy(), "sgc_s": sgc.numpy()[:, :, ::stride, ::stride].copy(),
           "cams_stats": st(cams), "sgc_stats": st(sgc), "emb": emb.numpy(), "logits": logits.numpy(),
           "bn_calibrated": bn_summary(model)}
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "cams max", float(cams.max()), "sgc max", float(sgc.max()), "emb", tuple(emb.shape))


def synth_photo(seed, h, w):
    """Deterministic uint8 RGB test image with structure at several scales (smooth gradients + blocks + noise)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    u = synth.uniform(seed, "photo", (h // 16 + 2, w // 16 + 2, 3))
    blocks = np.kron(u, np.ones((16, 16, 1)))[:h, :w]
    img = 0.45 * blocks + 0.35 * np.stack([np.sin(xx / 23.0 + c) * np.cos(yy / 17.0 - c) * 0.5 + 0.5 for c in range(3)], -1) \
        + 0.2 * synth.uniform(seed, "photo.noise", (h, w, 3))
    return np.clip(img * 255.0, 0, 255).astype(np.uint8)


def gen_input_units(src, fname="input_views.npz"):
    """the reference's own VOC12ImageViews.__getitem__ (src/data.py:306-315) with the train transform of
    train_mcl.py:104-115 (written out here: that script cannot be imported), on synthetic JPEGs in a temporary VOC tree.
    torchvision is not installed: its pieces on this path are supplied as what they do for PIL inputs -
    Compose = call in order; F.hflip = transpose(FLIP_LEFT_RIGHT); F.crop(img,i,j,h,w) = img.crop((j,i,j+w,i+h));
    F.resize(img,(448,448)) = img.resize((448,448), BILINEAR); ColorJitter and RandomErasing = IDENTITY (not pinned,
    not reproduced by the build).  Everything else (get_views, get_inter, RandomResizeLong, color_norm, RandomCrop,
    HWC_to_CHW) is the reference's code, run as is (Image.CUBIC, removed in Pillow 10, is re-aliased to Image.BICUBIC)."""
    import random
    import tempfile
    import zlib
    import PIL.Image
    import src.data as ref_data
    import src.imutils as ref_im

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Identity:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    tvt = sys.modules["torchvision.transforms"]
    tvt.Compose, tvt.ColorJitter, tvt.RandomErasing = Compose, Identity, Identity
    F_ = sys.modules["torchvision.transforms.functional"]
    F_.hflip = lambda im: im.transpose(PIL.Image.FLIP_LEFT_RIGHT)
    F_.crop = lambda im, i, j, h, w: im.crop((j, i, j + w, i + h))
    F_.resize = lambda im, size: im.resize(size[::-1], PIL.Image.BILINEAR)
    ref_data.transforms, ref_data.F = tvt, F_
    if not hasattr(PIL.Image, "CUBIC"):
        PIL.Image.CUBIC = PIL.Image.BICUBIC       # src/imutils.py:136 uses the alias Pillow 10 removed (same filter, value 3)
    out = {}
    sizes = [(375, 500), (500, 333), (300, 400), (480, 640), (460, 449), (448, 448)]
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "JPEGImages"))
        names = []
        for i, (h, w) in enumerate(sizes):
            name = f"2007_{i:06d}"
            PIL.Image.fromarray(synth_photo(40 + i, h, w)).save(os.path.join(td, "JPEGImages", name + ".jpg"), quality=92)
            names.append(name)
        lst = os.path.join(td, "list.txt")
        open(lst, "w").write("\n".join(names) + "\n")
        ds = ref_data.VOC12ImageViews(lst, td, transform=Compose([
            ref_im.RandomResizeLong(448, 768), Identity(), np.asarray, ref_im.color_norm, ref_im.RandomCrop(448),
            ref_im.HWC_to_CHW, torch.from_numpy, Identity()]), output_size=(224, 224))
        torch.manual_seed(1234)
        random.seed(4321)
        for i, name in enumerate(names):
            jpg = open(os.path.join(td, "JPEGImages", name + ".jpg"), "rb").read()
            nm, img, v1, v2, c1, c2, oc = ds[i]
            img, v1, v2 = img.numpy(), v1.numpy(), v2.numpy()
            assert img.dtype == np.float32 and v1.dtype == np.float64
            out[f"jpg{i}"] = np.frombuffer(jpg, dtype=np.uint8)
            out[f"coords{i}"] = np.array([c1, c2, oc], dtype=np.int64)
            # the full tensors would be 4 MB per item: bit-exact identity is pinned by CRC-32 of the fp32 bytes, values
            # by strided samples and sums
            v1f, v2f = v1.astype(np.float32), v2.astype(np.float32)              # train_mcl.py:163-165 .float()
            out[f"crc{i}"] = np.array([zlib.crc32(img.tobytes()), zlib.crc32(v1f.tobytes()), zlib.crc32(v2f.tobytes())], dtype=np.int64)
            out[f"img_s{i}"] = img[:, ::16, ::16].copy()
            out[f"v1_s{i}"] = v1f[:, ::16, ::16].copy()
            out[f"sums{i}"] = np.array([img.astype(np.float64).sum(), v1.sum(), v2.sum()])
    out["seeds"] = np.array([1234, 4321], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, {k: v.shape for k, v in out.items() if k.startswith("coords")}, out["coords0"].tolist())


def main_fullsize(src, tree):
    """Reference-pinned fixtures at the headline size (B7 / 448x448 scalars + summaries)."""
    gen_step(src, tree, "efficientnet-b7", 4, 448, 224, 4, 31, "step_b7_448_ep4.npz", cam_stride=16)
    gen_step(src, tree, "efficientnet-b7", 4, 448, 224, 12, 31, "step_b7_448_ep12_lr0.npz", lr=0.0, cam_stride=16)
    gen_forward_eval(src, "efficientnet-b7", 2, 563, 750, 32, "forward_b7_eval_563x750.npz")   # 375x500 x 1.5
    gen_forward_eval(src, "efficientnet-b7", 2, 448, 448, 33, "forward_b7_eval_448.npz")


def main_config2(src, tree):
    """BASELINE.json configs[1]'s workload shape (B0 at 448x448; the batch is 4 of the 16 rows so the fixture stays small):
    the reference's own loop body, epoch-4 gates."""
    gen_step(src, tree, "efficientnet-b0", 4, 448, 224, 4, 5, "step_b0_448_ep4.npz", cam_stride=16)


def main_config1(src, tree):
    """BASELINE.json configs[0]: train_mcl.py on EfficientNet-B0, 2 synthetic 224x224 images, 21 classes, one step - the
    reference's own loop body at exactly that shape (epoch-4 gates; with two images IMC has no qualifying anchor row and
    returns the Python float 0.0, train_mcl.py:194 skips it)."""
    gen_step(src, tree, "efficientnet-b0", 2, 224, 112, 4, 7, "step_b0_224_n2_ep4.npz", cam_stride=8)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    src = load_reference()
    tree = train_script_ast()
    gen_units(src, tree)
    gen_forward(src, "efficientnet-b0", 4, 64, 1, "forward_b0.npz")
    gen_forward(src, "efficientnet-b3", 4, 64, 2, "forward_b3.npz")
    gen_step(src, tree, "efficientnet-b0", 4, 96, 64, 12, 0, "step_b0_ep12.npz")
    gen_step(src, tree, "efficientnet-b0", 4, 64, 32, 4, 4, "step_b0_ep4_imc0.npz")   # IMC returns Python 0.0
    gen_step(src, tree, "efficientnet-b0", 4, 64, 32, 4, 5, "step_b0_ep4.npz")
    gen_step(src, tree, "efficientnet-b0", 4, 64, 32, 0, 5, "step_b0_ep0.npz")
    gen_step(src, tree, "efficientnet-b3", 4, 64, 64, 12, 6, "step_b3_ep12.npz")
    # lr = 0 variants: Adam's first update is sign(g)*lr, which turns the round-off gradients of
    # BN-cancelled parameters (e.g. _bn2.bias ahead of a train-mode BN) into +-lr steps that any two
    # fp32 implementations take differently; with lr = 0 phase 2 is evaluated on identical weights,
    # so its losses and gradients can be held to a tight tolerance.
    gen_step(src, tree, "efficientnet-b0", 4, 96, 64, 12, 0, "step_b0_ep12_lr0.npz", lr=0.0)
    gen_step(src, tree, "efficientnet-b3", 4, 64, 64, 12, 6, "step_b3_ep12_lr0.npz", lr=0.0)
    gen_step(src, tree, "efficientnet-b7", 4, 64, 32, 4, 8, "step_b7_ep4.npz")
    main_config4(src)
    gen_irn_units(src)
    gen_eval_units(src)
    main_fullsize(src, tree)
    main_config2(src, tree)
    main_config1(src, tree)
    gen_input_units(src)


if __name__ == "__main__":
    if "--input" in sys.argv:
        gen_input_units(load_reference())
    elif "--config2" in sys.argv:
        torch.set_num_threads(8)
        main_config2(load_reference(), train_script_ast())
    elif "--config1" in sys.argv:
        torch.set_num_threads(8)
        main_config1(load_reference(), train_script_ast())
    elif "--fullsize" in sys.argv:
        torch.set_num_threads(8)
        main_fullsize(load_reference(), train_script_ast())
    else:
        main()
