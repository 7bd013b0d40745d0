"""Forward / backward schedule of the EfficientNet MBConv chain on the HIP kernels.

This is the host-side orchestration of the path the reference runs through autograd
(src/efficientnet_pytorch/model.py:67-94 per block, :171-188 for the chain): it decides what is
materialised in HBM and what is recomputed in a consumer's prologue.

Per block, forward (NHWC, M = N*H*W rows in, M' rows out):
    x --pw_fwd--> e_raw (+BN0 stats) --dwconv_fwd[BN0+SiLU on load]--> d_raw (+BN1 stats)
      --pool_sum[BN1+SiLU on load]--> squeeze --se_fwd--> gate
    d_raw --pw_fwd[BN1+SiLU+gate on load]--> p_raw (+BN2 stats) --bn_apply[+drop_connect, +skip]--> out
    (project convs wider than MATERIALISE_ABOVE channels read a materialised a = swish(bn1(d_raw))*gate instead: with
    several N tiles the prologue recompute costs more than the extra pass, measured)
e_raw, d_raw, p_raw, out (and `a` where materialised) are what exists in HBM; they are also exactly what backward needs.

Per block, backward:
    g_out --BN2 bwd (reduce, finalise, apply)--> dp --pw_wgrad / pw_dgrad--> dA
    (dA, d_raw) --se_bn1_pool (one pass: SE gate gradient + BN1 backward sums per sample)--> se_bwd --> gh --bn1_coeffs--> add, c1..c3
    stride 1: dwconv_bwd_fused[BN1 data gradient on the fly; dW, dX, *swish'(bn0), BN0 sums] --> gz
    stride 2: bn_bwd_apply --> dwconv_bwd_weight, dwconv_bwd_data --> ge --BN0 bwd reduce
    gz --BN0 finalise + apply--> de --pw_wgrad / pw_dgrad (+skip gradient)--> g_in
With save=False (no backward will follow) every tensor only backward would read is released as the forward goes.
"""
from __future__ import annotations

import functools
import os

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import ops
from .arch import BlockCfg, NetCfg
from .ops import BNState


@dataclass
class BlockTape:
    cfg: BlockCfg
    H: int
    W: int
    Ho: int
    Wo: int
    x: torch.Tensor                       # block input [N,H,W,Cin] (materialised) or the stem's raw output for block 0
    x_st: Optional[BNState]               # BN+SiLU still to be applied to x (block 0 only)
    e_raw: Optional[torch.Tensor] = None
    bn0: Optional[BNState] = None
    d_raw: Optional[torch.Tensor] = None
    bn1: Optional[BNState] = None
    s: Optional[torch.Tensor] = None
    h: Optional[torch.Tensor] = None
    gate: Optional[torch.Tensor] = None
    a: Optional[torch.Tensor] = None     # materialised swish(bn1(d_raw))*gate, only where the project GEMM has > 1 N tile
    p_raw: Optional[torch.Tensor] = None
    bn2: Optional[BNState] = None
    row_scale: Optional[torch.Tensor] = None
    out: Optional[torch.Tensor] = None


@dataclass
class Tape:
    training: bool
    N: int
    cols: Optional[torch.Tensor] = None       # stem im2col [R,28]
    stem_raw: Optional[torch.Tensor] = None   # [N,H0,W0,C0]
    stem_bn: Optional[BNState] = None
    H0: int = 0
    W0: int = 0
    blocks: List[BlockTape] = field(default_factory=list)
    wt: Dict[int, torch.Tensor] = field(default_factory=dict)     # id(conv weight) -> transposed copy for the data gradient
    wt_event: Optional["torch.cuda.Event"] = None                 # ... valid on the main stream after this event
    wp: Dict[int, int] = field(default_factory=dict)              # id(conv weight) -> pre-split image of W (forward GEMMs)
    wtp: Dict[int, int] = field(default_factory=dict)             # id(conv weight) -> pre-split image of W^T (data gradients)


_keep_cache: Dict[tuple, torch.Tensor] = {}
MATERIALISE_ABOVE = int(os.environ.get("MUSCLE_MATERIALISE_ABOVE", "128"))      # project convs with Cout above this use a materialised activated input
# Inference reads the activated input once (no weight gradient shares it): the operand prologue of the project GEMM beats the
# extra pass at every width - batch 64, 448 / 512 / 768 px: 61.1 / 76.6 / 172.2 ms with the training threshold, 58.4 / 73.0 / 163.7 without
MATERIALISE_ABOVE_EVAL = int(os.environ.get("MUSCLE_MATERIALISE_ABOVE_EVAL", str(1 << 30)))
# ... where the GEMM has rows to spread the prologue's sigmoid over: the batch-1 multi-scale passes of infer_mcl.py (a few thousand
# rows in the late stages) keep the materialised input (one 500x375 image, 8 passes: 40.4 ms against 42.4)
EVAL_PROLOGUE_MIN_ROWS = int(os.environ.get("MUSCLE_EVAL_PROLOGUE_MIN_ROWS", "16384"))


def _blk(backbone, i):
    return backbone._blocks[i]


def stem_weight28(backbone):
    w = backbone._conv_stem.weight
    w28 = torch.zeros(w.shape[0], 28, dtype=torch.float32, device=w.device)
    w28[:, :27] = w.reshape(w.shape[0], 27)
    return w28


def _fold_sources(backbone, cfg: NetCfg):
    """Every tensor the inference fold is computed from."""
    for b in cfg.blocks:
        m = _blk(backbone, b.index)
        if b.expand:
            yield m._expand_conv.weight
            yield from (m._bn0.weight, m._bn0.bias, m._bn0.running_mean, m._bn0.running_var)
        yield from (m._bn1.weight, m._bn1.bias, m._bn1.running_mean, m._bn1.running_var)
        yield m._project_conv.weight
        yield from (m._bn2.weight, m._bn2.bias, m._bn2.running_mean, m._bn2.running_var)


def fold_fingerprint(backbone, cfg: NetCfg):
    """(storage address, in-place version) of every source tensor: moves when weights are loaded through ANY module
    (load_state_dict copies in place), edited in place, or moved to another device."""
    return tuple((t.data_ptr(), t._version) for t in _fold_sources(backbone, cfg))


def fold_eval_bn(backbone, cfg: NetCfg):
    """Inference-only constants of every block (infer_mcl.py:107-125 runs the model in eval mode): BN0 folded into the
    expand weight (+ bias), BN2 folded into the project weight (+ bias), BN1 as scale / shift for the depthwise kernel's
    SE squeeze and the project GEMM's operand prologue.  One launch per block (mx_fold_block): done once per model load by
    MuSCLe.fold_eval_bn(), or per forward when no cache is installed (always correct; the no-grad forward of view 2 in
    the training loop, train_mcl.py:205-206, is that case: the weights have just been stepped).

    The returned dict ALIASES the backbone's fold arena: the next fold overwrites the same tensors in place and returns a dict
    with a new 'fingerprint' - a dict kept from an earlier call then holds the NEW values under its OLD fingerprint.  Holders
    must re-fold when `fold_fingerprint` moves (backbone_forward does) or clone what they keep."""
    dev = backbone._conv_stem.weight.device
    # the folded tensors live with the backbone and are overwritten by every fold (the per-step refold of the training loop's
    # no-grad forward allocates nothing), so the pre-split images of the folded weights (ops.PlanesPlan: one table, one launch)
    # can be built once as well
    arena = backbone.__dict__.get("_fold_arena")
    key = (str(dev), tuple(b.index for b in cfg.blocks))
    if arena is not None and (arena["key"] != key or arena["ptrs"] != tuple(t.data_ptr() for _, _, t in arena["mats"])):
        # another device / block list, or a copy of the module (copy.deepcopy clones the tensors, the plan's table would still
        # name the original's): start over; the old buffers stay alive (a captured graph may still use them)
        backbone.__dict__.setdefault("_fold_arenas_retired", []).append(arena)
        arena = None
    fresh = arena is None
    if fresh:
        arena = {"key": key, "blocks": {}}
    for b in cfg.blocks:
        m = _blk(backbone, b.index)
        arena["blocks"][b.index] = ops.fold_block(m._expand_conv.weight.view(b.cexp, b.cin) if b.expand else None, m._bn0 if b.expand else None,
                                                  m._bn1, m._project_conv.weight.view(b.cout, b.cexp), m._bn2, out=arena["blocks"].get(b.index))
    if fresh:
        cmax = max(b.cexp for b in cfg.blocks)
        arena["one"] = torch.ones(cmax, dtype=torch.float32, device=dev)
        arena["zero"] = torch.zeros(cmax, dtype=torch.float32, device=dev)
        mats = []
        for b in cfg.blocks:
            f = arena["blocks"][b.index]
            if b.expand:
                mats.append((b.index, "We_planes", f["We"]))
            mats.append((b.index, "Wp_planes", f["Wp"]))
        arena["mats"] = mats
        arena["ptrs"] = tuple(t.data_ptr() for _, _, t in mats)
        arena["plan"] = None if torch.cuda.is_current_stream_capturing() else ops.PlanesPlan([t for _, _, t in mats])
        backbone.__dict__["_fold_arena"] = arena
    if arena["plan"] is None and not torch.cuda.is_current_stream_capturing():
        arena["plan"] = ops.PlanesPlan([t for _, _, t in arena["mats"]])      # first built under capture: the plan is made on the next eager fold
    if arena["plan"] is not None:
        for (bi, name, _), im in zip(arena["mats"], arena["plan"].run()):
            arena["blocks"][bi][name] = im
    out = dict(arena["blocks"])
    out["one"], out["zero"] = arena["one"], arena["zero"]
    out["fingerprint"] = fold_fingerprint(backbone, cfg)
    return out


def _block_forward_eval(m, b: BlockCfg, f, ident, x, x_st, N, h, w, ho, wo):
    """One MBConv block in inference form: no statistics, no p_raw / separate BN2 pass, the SE squeeze leaves the
    depthwise kernel, BN2 + skip ride in the project GEMM's epilogue.  HBM: in + 2*exp + 2*dw + out (+ skip)."""
    M, Mo = N * h * w, N * ho * wo
    if b.expand:
        e = ops.pw_fwd(x.view(M, b.cin), f["We"], b.cexp, bias=f["be"], planes=f.get("We_planes")).view(N, h, w, b.cexp)      # = bn0(conv(x))
        dw_in, dw_st = e, BNState(ident[0][:b.cexp], ident[1][:b.cexp], None, None)                 # swish on load
    else:
        dw_in, dw_st = x, x_st
    d, pooled = ops.dwconv_fwd(dw_in, m._depthwise_conv.weight, b.kernel, b.stride, b.pad_lo, ho, wo, st=dw_st, pool=f["bn1"][:2])
    _, _, gate = ops.se_fwd(pooled, 1.0 / (ho * wo), m._se_reduce.weight.view(b.se, b.cexp), m._se_reduce.bias,
                            m._se_expand.weight.view(b.cexp, b.se), m._se_expand.bias)
    d2 = d.view(Mo, b.cexp)
    res = x.view(M, b.cin) if b.skip else None
    if b.cout > MATERIALISE_ABOVE_EVAL or (Mo < EVAL_PROLOGUE_MIN_ROWS and b.cout > MATERIALISE_ABOVE):
        a = ops.bn_apply(d2, f["bn1"], gate=gate, rows_per_sample=ho * wo, act=True)
        out = ops.pw_fwd(a, f["Wp"], b.cout, bias=f["bp"], residual=res, planes=f.get("Wp_planes"))
    else:
        out = ops.pw_fwd(d2, f["Wp"], b.cout, a_mode=ops.BNACT, a_scale=f["bn1"].scale, a_shift=f["bn1"].shift, a_gate=gate,
                         rows_per_sample=ho * wo, bias=f["bp"], residual=res)
    return out.view(N, ho, wo, b.cout)


EVAL_FOLD = os.environ.get("MUSCLE_EVAL_FOLD", "1") == "1"


def _prepare_weights(backbone, cfg: NetCfg, tape: Tape, dev, backward: bool):
    """Per-step derivatives of the 1x1 conv weights (ops.WeightPlan).  The pre-split images of W feed the forward GEMMs of the
    second-generation split kernel: one launch on the main stream, ahead of the stem.  With `backward`, the data-gradient GEMMs
    run as forward GEMMs against W^T (ops.pw_dgrad): the ~110 transposes (one launch) and the images of W^T (one launch)
    depend on nothing but the weights, so they go to the side stream at the start of the forward, where they fill the gaps of
    the main stream's kernel chain instead of sitting on the backward's critical path."""
    ws = []
    for b in cfg.blocks:
        m = _blk(backbone, b.index)
        if b.expand and b.cexp % 4 == 0:
            ws.append((m._expand_conv.weight, m._expand_conv.weight.view(b.cexp, b.cin)))
        if b.cout % 4 == 0:
            ws.append((m._project_conv.weight, m._project_conv.weight.view(b.cout, b.cexp)))
    if not ws:
        return
    # buffers and tables live with the backbone (the previous step's backward, the only other reader, is behind the stream order);
    # rebuilt when a weight has moved (load on another device, re-created parameters)
    plan = getattr(backbone, "_wt_plan", None)
    views = [v for _, v in ws]
    if plan is None or not plan.matches(views):
        # (building the plan uploads tables: not under stream capture - a captured step whose eager warm-up did not come
        # through here transposes weight by weight, into the graph's own pool, and runs the first-generation split kernel)
        if plan is not None:                       # a captured graph may still write into its buffers: retired, not freed
            backbone.__dict__.setdefault("_wt_plans_retired", []).append(plan)
        plan = None if torch.cuda.is_current_stream_capturing() else ops.WeightPlan(views)
        backbone._wt_plan = plan
    if plan is not None and ops.get_gemm_mode() != 0:
        for (w, _), im in zip(ws, plan.run_forward()):
            if im is not None:
                tape.wp[id(w)] = im
    if not (backward and ops.DGRAD_AS_FORWARD):
        return
    side = _WgradLane(dev).s
    if side is None:
        if plan is not None:
            wts, ims = plan.run_backward()
            for (w, _), wt, im in zip(ws, wts, ims):
                tape.wt[id(w)] = wt
                if im is not None and ops.get_gemm_mode() != 0:
                    tape.wtp[id(w)] = im
        return
    side.wait_stream(torch.cuda.current_stream())            # the optimizer step that produced these weights
    with torch.cuda.stream(side):
        if plan is not None:
            wts, ims = plan.run_backward()
            for (w, _), wt, im in zip(ws, wts, ims):
                tape.wt[id(w)] = wt
                if im is not None and ops.get_gemm_mode() != 0:
                    tape.wtp[id(w)] = im
        else:
            for w, v in ws:
                tape.wt[id(w)] = ops.transpose(v)
        tape.wt_event = side.record_event()


def backbone_forward(backbone, cfg: NetCfg, img: torch.Tensor, training: bool,
                     drop_u: Optional[Dict[int, torch.Tensor]] = None, save: bool = True) -> Tape:
    """img: NCHW fp32 CUDA.  Returns the tape; block outputs are tape.blocks[i].out (NHWC).
    save=False (no backward will follow, e.g. under torch.no_grad()): every tensor that only backward would read is
    released as soon as its consumer has been enqueued, and only the tapped block outputs stay alive, so inference at
    batch 64 / 768x768 does not carry the training footprint."""
    N, _, H, W = img.shape
    dev = img.device
    tape = Tape(training=training, N=N)
    lo, hi = cfg.stem_pad
    H0, W0 = (H + lo + hi - 3) // 2 + 1, (W + lo + hi - 3) // 2 + 1
    tape.H0, tape.W0 = H0, W0
    # save: a backward will follow (train mode, or phase 2's eval-mode forward of view 1)
    if save or training or not EVAL_FOLD:
        _prepare_weights(backbone, cfg, tape, dev, backward=save)
    # stem: im2col + MFMA GEMM (K = 27 padded to 28), BN statistics in the GEMM epilogue
    tape.cols = ops.stem_im2col(img, H0, W0, lo)
    C0 = cfg.stem_out
    raw, stats = ops.pw_fwd(tape.cols, stem_weight28(backbone), C0, want_stats=True) if training else \
        (ops.pw_fwd(tape.cols, stem_weight28(backbone), C0), None)
    tape.stem_raw = raw.view(N, H0, W0, C0)
    tape.stem_bn = ops.bn_finalize(stats, N * H0 * W0, backbone._bn0, training)

    drop_scales: Dict[int, torch.Tensor] = {}
    if training and drop_u is None:
        # drop_connect (utils.py:50-60): every block's Bernoulli(keep)/keep row scale in one go - per block it was a rand,
        # an add, a floor and a div kernel, ~190 five-microsecond launches per B7 step
        idx = [b.index for b in cfg.blocks if b.skip and b.drop_rate]
        if idx:
            rates = tuple(1.0 - cfg.blocks[i].drop_rate for i in idx)
            key = (rates, str(dev))                     # by value: a cfg object's id can be reused after it is collected
            if key not in _keep_cache:
                _keep_cache[key] = torch.tensor(rates, dtype=torch.float32).to(dev).view(-1, 1)
            keep = _keep_cache[key]
            rs = torch.floor(keep + torch.rand(len(idx), N, device=dev)) / keep
            drop_scales = {i: rs[j] for j, i in enumerate(idx)}
    x, x_st, h, w = tape.stem_raw, tape.stem_bn, H0, W0
    fold = None
    if not training and not save and EVAL_FOLD:
        fold = getattr(backbone, "_eval_fold", None)
        if fold is not None and fold["fingerprint"] != fold_fingerprint(backbone, cfg):
            # the cached fold is of other weights (a submodule's load_state_dict, an in-place edit, .to(device)): redo it
            fold = backbone._eval_fold = fold_eval_bn(backbone, cfg)
        elif fold is None:
            fold = fold_eval_bn(backbone, cfg)
    for b in cfg.blocks:
        m = _blk(backbone, b.index)
        ho, wo = b.out_size(h), b.out_size(w)
        t = BlockTape(cfg=b, H=h, W=w, Ho=ho, Wo=wo, x=x, x_st=x_st)
        if fold is not None:
            # inference-only path (BASELINE.json configs[4]): nothing is kept for a backward
            t.out = _block_forward_eval(m, b, fold[b.index], (fold["one"], fold["zero"]), x, x_st, N, h, w, ho, wo)
            t.x = None
            tape.blocks.append(t)
            if b.index == 0:
                tape.cols = tape.stem_raw = None
            if len(tape.blocks) >= 2 and tape.blocks[-2].cfg.index not in cfg.taps:
                tape.blocks[-2].out = None
            x, x_st, h, w = t.out, None, ho, wo
            continue
        M, Mo = N * h * w, N * ho * wo
        if b.expand:
            assert x_st is None
            st0 = None
            t.e_raw = ops.pw_fwd(x.view(M, b.cin), m._expand_conv.weight.view(b.cexp, b.cin), b.cexp, want_stats=training,
                                 planes=tape.wp.get(id(m._expand_conv.weight)))
            if training:
                t.e_raw, st0 = t.e_raw
            t.e_raw = t.e_raw.view(N, h, w, b.cexp)
            t.bn0 = ops.bn_finalize(st0, M, m._bn0, training)
            dw_in, dw_st = t.e_raw, t.bn0
        else:
            dw_in, dw_st = x, x_st
        st1 = None
        t.d_raw = ops.dwconv_fwd(dw_in, m._depthwise_conv.weight, b.kernel, b.stride, b.pad_lo, ho, wo, st=dw_st,
                                 want_stats=training)
        if training:
            t.d_raw, st1 = t.d_raw
        t.bn1 = ops.bn_finalize(st1, Mo, m._bn1, training)
        d2 = t.d_raw.view(Mo, b.cexp)
        pooled = ops.pool_sum(d2, ho * wo, st=t.bn1, act=True)
        t.s, t.h, t.gate = ops.se_fwd(pooled, 1.0 / (ho * wo), m._se_reduce.weight.view(b.se, b.cexp), m._se_reduce.bias,
                                      m._se_expand.weight.view(b.cexp, b.se), m._se_expand.bias)
        st2 = None
        if b.cout > MATERIALISE_ABOVE:
            # With several N tiles the BN+SiLU+gate prologue would be recomputed (and its scale/shift/gate re-loaded
            # through the per-CU load path) once per tile: measured 100 -> 63 TFLOP/s at K=3840, N=640.  One streaming
            # pass writes the activated tensor instead; it is kept for the weight gradient.
            t.a = ops.bn_apply(d2, t.bn1, gate=t.gate, rows_per_sample=ho * wo, act=True)
            t.p_raw = ops.pw_fwd(t.a, m._project_conv.weight.view(b.cout, b.cexp), b.cout, want_stats=training,
                                 planes=tape.wp.get(id(m._project_conv.weight)))
        else:
            t.p_raw = ops.pw_fwd(d2, m._project_conv.weight.view(b.cout, b.cexp), b.cout, a_mode=ops.BNACT,
                                 a_scale=t.bn1.scale, a_shift=t.bn1.shift, a_gate=t.gate, rows_per_sample=ho * wo,
                                 want_stats=training, planes=tape.wp.get(id(m._project_conv.weight)))
        if training:
            t.p_raw, st2 = t.p_raw
        t.bn2 = ops.bn_finalize(st2, Mo, m._bn2, training)
        if b.skip and training and b.drop_rate:
            if drop_u is not None:                      # recorded draws (parity tests)
                keep = 1.0 - b.drop_rate
                t.row_scale = torch.floor(keep + drop_u[b.index].to(dev, torch.float32)) / keep
            else:
                t.row_scale = drop_scales[b.index]
        res = None
        if b.skip:
            # the skip input is the activated block input; for the (never skipping) block 0 it would be raw
            assert x_st is None
            res = x.view(M, b.cin)
        t.out = ops.bn_apply(t.p_raw, t.bn2, row_scale=t.row_scale, residual=res, rows_per_sample=ho * wo).view(N, ho, wo, b.cout)
        tape.blocks.append(t)
        x, x_st, h, w = t.out, None, ho, wo
        if not save:
            t.x = t.e_raw = t.d_raw = t.p_raw = t.a = None
            if b.index == 0:
                tape.cols = tape.stem_raw = None
            if len(tape.blocks) >= 2 and tape.blocks[-2].cfg.index not in cfg.taps:
                tape.blocks[-2].out = None
    ops.flush_batch_counters()
    return tape


class GradSink:
    """Where parameter gradients go: maps a parameter to its (accumulating) fp32 gradient buffer."""

    def __init__(self):
        self.bufs: Dict[int, torch.Tensor] = {}

    def of(self, p: torch.nn.Parameter) -> torch.Tensor:
        g = self.bufs.get(id(p))
        if g is None:
            g = torch.zeros_like(p, memory_format=torch.contiguous_format)
            self.bufs[id(p)] = g
        return g


FUSED_DW_BACKWARD = True
# BN0 backward apply inside the expand data-gradient GEMM's operand load (the GEMM also writes dz for the weight gradient):
# one pass over the Cexp-wide tensors less on paper, measured on MI355X (B7/448/bs32, A/B/A/B on one box) 130.2 / 130.5 ms
# per step without against 133.0 / 133.2 with it - the second operand stream and the dz stores of the N-tile-0 workgroups
# cost the MFMA-bound GEMMs more than the 8.5 ms pass gives back a third of.  Off; kept as an option.
FOLD_BN0_APPLY = os.environ.get("MUSCLE_FOLD_BN0", "0") == "1"
# Round 4: dZ = c1*G + c2*X + c3 formed in the operand loads of BOTH consumers (the second-generation split data gradient, whose
# activations go straight to registers, and the split weight gradient's loader threads), dZ never written: the 2R + 1W pass over
# the Cexp-wide tensors of the 28 x 28 stages disappears (-2.3 ms of bn_bwd_apply per step, -13 GB of traffic).  Measured
# A/B/A/B on one box (profiles/r04_fold_bn0_both_ab.txt): 107.4 / 107.7 ms per step without, 109.9 / 110.0 with - the folded GEMMs
# lose more (+5.3 ms: 192 registers take the data gradient from 3 to 2 workgroups per CU, 222 the weight gradient from 3 to 2 waves
# per SIMD) than the pass costs.  Off; MUSCLE_FOLD_BN0_BOTH=1 turns it on (kernels and tests stay: tests/test_gpu_split.py).
FOLD_BN0_BOTH = os.environ.get("MUSCLE_FOLD_BN0_BOTH", "0") == "1"
# ... but in stages 1-2 (expand convs 32 -> 192, 48 -> 288 at 0.4-1.6 M rows) both consumers are HBM-bound, and there the same fold
# (data gradient through the planes kernel, weight gradient through wgrad_small_kernel<..., GBN>) trades a 3-pass kernel for one
# more operand stream in two kernels that wait on memory anyway.  MUSCLE_FOLD_BN0_EARLY=0 restores the pass.
FOLD_BN0_EARLY = os.environ.get("MUSCLE_FOLD_BN0_EARLY", "1") == "1"
# Round 5: the fold on the weight-gradient side only.  The expand convolution's weight gradient runs FIRST, on the main stream; the
# loader waves of wgrad_split_ws_kernel<true> form dZ and store it, the data gradient reads the stored dZ as before: bn_bwd_apply is gone
# for the split-arithmetic layers (stages 4-7), neither GEMM's matrix waves carry a second operand.  Measured (profiles/r05_knob_ab.txt):
# bn_bwd_apply 7.00 -> 2.94 ms per step, but the 42 folded weight gradients 6.6 -> 10.4 ms (their two G loader waves carry twice the
# requests, the stores and the FMAs; the kernel is sensitive to its L2 traffic, which grows 1.75x) - step 96.50 -> 96.16 ms A/B/A/B on one box, 96.77 -> 97.09 on another.
# Too little for a second dZ buffer per block: off by default (MUSCLE_FOLD_BN0_WGRAD=1); kernel and test stay.
FOLD_BN0_WGRAD = os.environ.get("MUSCLE_FOLD_BN0_WGRAD", "0") == "1"
# Weight-gradient GEMMs on a second HIP stream (MUSCLE_WGRAD_STREAM=0 turns it off; `engine.WGRAD_SIDE_STREAM` can be
# flipped at run time).  Nothing in the backward chain consumes them (only the optimizer and the gradient exchange do),
# they are MFMA-bound, and the chain between two of them (BN backward, SE, depthwise) is HBM-bound.  Measured on
# MI355X, B7/448/bs32: round 1 162.0 -> 156.3 ms/step (+3.6 %), whether the weight gradient starts with or after its
# data-gradient twin - both kernels fill every CU's wave slots, so the second one only trickles in.  Under overlap the
# duration of a single launch says nothing about the kernel: bench.py takes its per-launch HIP events with this off.
WGRAD_SIDE_STREAM = os.environ.get("MUSCLE_WGRAD_STREAM", "1") == "1"
_side_streams: Dict[tuple, "torch.cuda.Stream"] = {}


def _masked_stream(device, n_cus: int):
    """A HIP stream confined to `n_cus` of the chip's compute units (hipExtStreamCreateWithCUMask), wrapped for torch.
    Stream creation is plumbing, so it goes straight to the HIP runtime torch has already loaded."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    total = torch.cuda.get_device_properties(device).multi_processor_count
    n_cus = max(1, min(int(n_cus), total))
    words = (total + 31) // 32
    # every (256 // n)-th CU rather than the first n: the mask's bit order walks the XCDs / shader engines round-robin on
    # some runtimes and CU-major on others; an evenly spaced pattern takes the same share of every XCD under both
    mask = (ctypes.c_uint32 * words)()
    for i in range(n_cus):
        cu = (i * total) // n_cus
        mask[cu // 32] |= 1 << (cu % 32)
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), mask)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed (hipError {rc})")
    return torch.cuda.ExternalStream(handle.value, device=device)


# Weight-gradient side stream confined to this many CUs (0 = the whole chip).  The side stream's 180 us GEMM workgroups
# otherwise take wave slots on every CU and the small latency-bound kernels of the main chain queue behind them
# (profiles/r02_b_timeline.txt: se_bwd_b 134.9 us instead of 25.7 under overlap).
WGRAD_CUS = int(os.environ.get("MUSCLE_WGRAD_CUS", "0"))
# MUSCLE_FUSED_BN0_FINALIZE=1: the fused depthwise backward finishes the BatchNorm-0 backward statistics itself (its last workgroup per
# channel chunk; mx_dwconv_bwd_fused_bn0, same bits) instead of a bn_reduce_finalize launch between it and the BN0 apply / the folded
# GEMMs: 51 launches fewer per step and no time gained (alternating on one box 96.15 / 96.21, 96.74 / 96.60, 96.47 / 96.17 ms) - the
# backward is bound by the sum of its kernels' work, not by the chain's launches.  Off; built and tested (tests/test_gpu_dwfused.py).
FUSED_BN0_FINALIZE = os.environ.get("MUSCLE_FUSED_BN0_FINALIZE", "0") != "0"
# SE excitation parameter gradients (se_bwd_b) on the side stream instead of the data-gradient chain
SE_PARAMS_ASIDE = os.environ.get("MUSCLE_SE_PARAMS_ASIDE", "0") != "0"     # measured 0.6 ms per step SLOWER (profiles/r04_knob_sweep.txt)
# Priority of the weight-gradient side stream: 0 = as the main stream, -1 = higher, 1 = LOWER (a raw HIP stream: torch only offers -1 / 0)
WGRAD_PRIO = int(os.environ.get("MUSCLE_WGRAD_PRIO", "0"))


def _low_priority_stream(device, prio: int):
    """A HIP stream of priority `prio` > 0 (LOWER than the default streams; torch.cuda.Stream clamps to [-1, 0]), wrapped for torch.
    hipDeviceGetStreamPriorityRange on MI355X: least = 1, greatest = -1."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(handle), ctypes.c_uint(1), ctypes.c_int(int(prio)))      # hipStreamNonBlocking
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipStreamCreateWithPriority failed (hipError {rc})")
    return torch.cuda.ExternalStream(handle.value, device=device)


class _WgradLane:
    def __init__(self, device):
        self.s = None
        self.pending = []
        self.keep = []
        if WGRAD_SIDE_STREAM:
            key = (device.index if device.index is not None else torch.cuda.current_device(), WGRAD_CUS)
            if key not in _side_streams:
                if WGRAD_CUS > 0:
                    _side_streams[key] = _masked_stream(device, WGRAD_CUS)
                elif WGRAD_PRIO > 0:
                    _side_streams[key] = _low_priority_stream(device, WGRAD_PRIO)
                else:
                    _side_streams[key] = torch.cuda.Stream(device=device, priority=WGRAD_PRIO)
            self.s = _side_streams[key]

    def wgrad_bnbwd(self, G, G2, coef, X, dW):
        """Queue dW += (c1*G + c2*G2 + c3)^T X (ops.pw_wgrad_bnbwd)."""
        if self.s is None:
            return ops.pw_wgrad_bnbwd(G, G2, coef, X, dW)
        self.pending.append((G, X, dW, {"_bnbwd": (G2, coef)}))

    def dw_reduce(self, scratch, dW):
        """Queue the addition of a depthwise weight gradient's partial rows (ops.dw_parts_reduce): no consumer before Adam."""
        if self.s is None:
            return ops.dw_parts_reduce(scratch, dW)
        self.pending.append((scratch, None, dW, {"_dwreduce": True}))

    def defer(self, fn, *tensors):
        """Queue any launch whose results nothing reads before the optimizer (`tensors` = its operands, kept alive until join())."""
        if self.s is None:
            return fn()
        self.pending.append((tensors, None, None, {"_fn": fn}))

    def wgrad(self, G, X, dW, **kw):
        """Queue dW += G^T X'.  It is launched by the next flush(), i.e. right after the data-gradient GEMM of the same
        conv has been enqueued on the main stream: two MFMA-bound GEMMs side by side gain nothing, a weight-gradient GEMM
        next to the HBM-bound kernels that follow the data gradient does."""
        if self.s is None:
            return ops.pw_wgrad(G, X, dW, **kw)
        self.pending.append((G, X, dW, kw))

    def flush(self):
        if self.s is None or not self.pending:
            return
        self.s.wait_stream(torch.cuda.current_stream())     # operands ready, the data-gradient GEMM done
        with torch.cuda.stream(self.s):
            for G, X, dW, kw in self.pending:
                if "_fn" in kw:
                    kw["_fn"]()
                elif "_dwreduce" in kw:
                    ops.dw_parts_reduce(G, dW)
                elif "_bnbwd" in kw:
                    ops.pw_wgrad_bnbwd(G, kw["_bnbwd"][0], kw["_bnbwd"][1], X, dW)
                else:
                    ops.pw_wgrad(G, X, dW, **kw)
        # the caller drops G and X before the side stream has read them: they stay referenced until join() (one dz per
        # block, ~10 GB for B7/448/bs32 of 288; no record_stream, so the same code can be captured into a hipGraph)
        self.keep.extend((G, X, kw.get("_bnbwd")) for G, X, _, kw in self.pending)
        self.pending = []

    def progress(self, done, module):
        """Data parallelism: report `module`'s (and every later block's) gradients complete.  The exchange it may start
        must come after the kernels of BOTH streams, so it is issued from the side stream once that has caught up with
        the main one."""
        if self.s is None:
            return done(module)
        self.flush()
        self.s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.s):
            done(module)

    def join(self):
        self.flush()
        if self.s is not None:
            torch.cuda.current_stream().wait_stream(self.s)
        self.keep = []


def backbone_backward(backbone, cfg: NetCfg, tape: Tape, tap_grads: Dict[int, torch.Tensor], sink: GradSink):
    """tap_grads: {block index: dL/d out [N,Ho,Wo,Cout]} for the tapped features.  Parameter gradients are
    accumulated into `sink`.  The image gets no gradient (the reference never asks for one)."""
    N, training = tape.N, tape.training
    g_out: Optional[torch.Tensor] = None
    lane = _WgradLane(tape.blocks[0].out.device)
    if tape.wt_event is not None:
        torch.cuda.current_stream().wait_event(tape.wt_event)
    for t in reversed(tape.blocks):
        b, m = t.cfg, _blk(backbone, t.cfg.index)
        tg = tap_grads.get(b.index)
        if tg is not None:
            g_out = tg if g_out is None else g_out + tg
        if g_out is None:
            continue                                   # blocks after the last tap get no gradient
        M, Mo = N * t.H * t.W, N * t.Ho * t.Wo
        hw = t.Ho * t.Wo
        g2 = g_out.reshape(Mo, b.cout)
        # BN2 backward (upstream gradient carries the drop_connect scale of the sample)
        dp = ops.bn_backward(g2, t.p_raw, m._bn2, t.bn2, sink.of(m._bn2.weight), sink.of(m._bn2.bias), training,
                             row_scale=t.row_scale, rows_per_sample=hw)
        # project conv: weight gradient against the recomputed activated+gated input, then data gradient
        d2 = t.d_raw.view(Mo, b.cexp)
        if t.a is not None:
            lane.wgrad(dp, t.a, sink.of(m._project_conv.weight).view(b.cout, b.cexp))
            t.a = None
        else:
            lane.wgrad(dp, d2, sink.of(m._project_conv.weight).view(b.cout, b.cexp), x_mode=ops.BNACT, x_scale=t.bn1.scale,
                       x_shift=t.bn1.shift, x_gate=t.gate, rows_per_sample=hw)
        ga = ops.pw_dgrad(dp, m._project_conv.weight.view(b.cout, b.cexp), b.cexp, wt=tape.wt.get(id(m._project_conv.weight)),
                          planes=tape.wtp.get(id(m._project_conv.weight)))                    # dL/d(act*gate) [Mo,Cexp]
        lane.flush()
        del dp
        # One pass over (ga, d_raw) yields the SE gate gradient sum_hw ga*act AND the per-sample pieces of the BN1 backward
        # sums; the excitation backward then gives the pooled-path term `add`, and the BN1 sums follow without
        # touching the big tensors again.
        pooled5 = ops.se_bn1_pool(ga, d2, t.bn1, hw)
        W2 = m._se_expand.weight.view(b.cexp, b.se)
        if SE_PARAMS_ASIDE:
            # the excitation's parameter gradients have no consumer on this chain: beside the weight-gradient GEMMs
            gh = ops.se_bwd_gh(pooled5[0], t.gate, t.h, W2)
            lane.defer(functools.partial(ops.se_bwd_params, pooled5[0], t.gate, t.s, t.h, gh,
                                         sink.of(m._se_reduce.weight).view(b.se, b.cexp), sink.of(m._se_reduce.bias),
                                         sink.of(m._se_expand.weight).view(b.cexp, b.se), sink.of(m._se_expand.bias)),
                       pooled5, t.gate, t.s, t.h, gh)
        else:
            gh = ops.se_bwd(pooled5[0], t.gate, t.s, t.h, W2,
                            sink.of(m._se_reduce.weight).view(b.se, b.cexp), sink.of(m._se_reduce.bias),
                            sink.of(m._se_expand.weight).view(b.cexp, b.se), sink.of(m._se_expand.bias))
        # pooled-path gradient `add`, the BN1 backward sums and their finalisation: one launch, no second pass over the tensors
        c1, add = ops.bn1_coeffs(pooled5, t.gate, gh, m._se_reduce.weight.view(b.se, b.cexp), 1.0 / hw, Mo, m._bn1, t.bn1,
                                 sink.of(m._bn1.weight), sink.of(m._bn1.bias), training)
        dw_in, dw_st = (t.e_raw, t.bn0) if b.expand else (t.x, t.x_st)
        skip_res = g_out if b.skip else None            # d out / d x through the identity branch
        fused = FUSED_DW_BACKWARD and b.stride == 1 and b.pad_lo == (b.kernel - 1) // 2
        if fused:
            # stride 1: BN1 data gradient, depthwise weight + data gradients and the BN0 backward sums in one kernel
            bn_mod0 = m._bn0 if b.expand else backbone._bn0
            fin0 = (bn_mod0, sink.of(bn_mod0.weight), sink.of(bn_mod0.bias), training) if (FUSED_BN0_FINALIZE and dw_st is not None) else None
            res = ops.dwconv_bwd_fused(ga.view(N, t.Ho, t.Wo, b.cexp), t.d_raw, t.gate, add, t.bn1, c1, dw_in, dw_st,
                                       m._depthwise_conv.weight, sink.of(m._depthwise_conv.weight), b.kernel, b.pad_lo,
                                       residual=None if dw_st is not None else skip_res, defer=lane.dw_reduce, bn0=fin0)
            gx, part0, c0_fin = res if fin0 is not None else (res[0], res[1], None)
            del ga
        else:
            # BN1 backward with g = (ga*gate + add) * swish'(bn1(d_raw)), in place over ga
            dd = ops.bn_backward_from_coeffs(ga, d2, t.bn1, c1, gate=t.gate, gate_add=add, rows_per_sample=hw,
                                             out=ga).view(N, t.Ho, t.Wo, b.cexp)
            ops.dwconv_bwd_weight(dw_in, dd, sink.of(m._depthwise_conv.weight), b.kernel, b.stride, b.pad_lo, st=dw_st)
            gx = ops.dwconv_bwd_data(dd, m._depthwise_conv.weight, b.kernel, b.stride, b.pad_lo, t.H, t.W,
                                     residual=None if dw_st is not None else skip_res)
            del dd, ga
        if dw_st is not None:
            # the depthwise input sits behind a BatchNorm + SiLU (BN0 of an expand block, or the stem's BN for block 0)
            bn_mod = m._bn0 if b.expand else backbone._bn0
            raw2 = dw_in.view(M, dw_in.shape[3])
            gx2 = gx.view(M, dw_in.shape[3])
            fold = fused and b.expand and FOLD_BN0_APPLY and ops.DGRAD_AS_FORWARD and b.cexp % 4 == 0
            # round 4: the apply folded into BOTH consumers (dZ never written) where both run in split arithmetic
            wtp = tape.wtp.get(id(m._expand_conv.weight)) if b.expand else None
            kind = ops.bnbwd_fold_takes(M, b.cexp, b.cin) if (fused and b.expand and not fold and wtp is not None and ops.DGRAD_AS_FORWARD) else None
            fold2 = (kind == "tile" and FOLD_BN0_BOTH) or (kind == "small" and FOLD_BN0_EARLY)
            fold3 = (fused and b.expand and not fold and not fold2 and FOLD_BN0_WGRAD and gx2.is_contiguous() and raw2.is_contiguous()
                     and ops.wgrad_bnbwd_dz_takes(M, b.cexp, b.cin))
            if fused:
                c0 = c0_fin if c0_fin is not None else ops.bn_bwd_coeffs(part0, M, bn_mod, dw_st, sink.of(bn_mod.weight), sink.of(bn_mod.bias), training)
                if fold3:
                    dz = ops.pw_wgrad_bnbwd_dz(gx2, raw2, c0, t.x.view(M, b.cin), sink.of(m._expand_conv.weight).view(b.cexp, b.cin))
                elif not fold and not fold2:
                    dz = ops.bn_bwd_apply_plain(gx2, raw2, c0, gx2)
            else:
                dz = ops.bn_backward(gx2, raw2, bn_mod, dw_st, sink.of(bn_mod.weight), sink.of(bn_mod.bias), training,
                                     act=dw_st, out=gx2)
            if fold2:
                lane.wgrad_bnbwd(gx2, raw2, c0, t.x.view(M, b.cin), sink.of(m._expand_conv.weight).view(b.cexp, b.cin))
                g_in = ops.pw_dgrad_bnbwd_planes(gx2, raw2, c0, wtp, b.cin,
                                                 residual=skip_res.reshape(M, b.cin) if skip_res is not None else None)
                lane.flush()
                g_out = g_in.view(N, t.H, t.W, b.cin)
            elif fold:
                # the BN0 backward apply rides in the data-gradient GEMM's operand load (which also writes dz for the weight
                # gradient): one pass over the Cexp-wide tensors less than bn_bwd_apply + GEMM
                g_in, dz = ops.pw_dgrad_bnbwd(gx2, raw2, c0, m._expand_conv.weight.view(b.cexp, b.cin), b.cin,
                                              residual=skip_res.reshape(M, b.cin) if skip_res is not None else None,
                                              wt=tape.wt.get(id(m._expand_conv.weight)))
                lane.wgrad(dz, t.x.view(M, b.cin), sink.of(m._expand_conv.weight).view(b.cexp, b.cin))
                lane.flush()
                g_out = g_in.view(N, t.H, t.W, b.cin)
            elif b.expand:
                if not fold3:
                    lane.wgrad(dz, t.x.view(M, b.cin), sink.of(m._expand_conv.weight).view(b.cexp, b.cin))
                g_in = ops.pw_dgrad(dz, m._expand_conv.weight.view(b.cexp, b.cin), b.cin, wt=tape.wt.get(id(m._expand_conv.weight)),
                                    planes=tape.wtp.get(id(m._expand_conv.weight)),
                                    residual=skip_res.reshape(M, b.cin) if skip_res is not None else None)
                lane.flush()
                g_out = g_in.view(N, t.H, t.W, b.cin)
            else:
                # block 0: the input is the stem's raw output -> stem weight gradient
                dw28 = torch.zeros(cfg.stem_out, 28, dtype=torch.float32, device=dz.device)
                ops.pw_wgrad(dz, tape.cols, dw28)
                sink.of(backbone._conv_stem.weight).view(cfg.stem_out, 27).add_(dw28[:, :27])
                g_out = None
        else:
            g_out = gx
        done = getattr(sink, "block_done", None)
        if done is not None and getattr(sink, "on_ready", None) is not None:
            lane.progress(done, m)   # this block's (and every later block's) parameter gradients are enqueued
    lane.join()
    return
