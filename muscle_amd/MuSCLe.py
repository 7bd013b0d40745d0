"""MuSCLe model (CAM encoder mode) — drop-in for the reference's `src.MuSCLe.MuSCLe`.

Same constructor, same `forward(x, cam=...)` return tuples, same `state_dict()` keys
(src/MuSCLe.py:156-298); the computation is the HIP path of muscle_amd.engine
(backbone) plus the head below.  Differences a caller can see:
  * the constructor never downloads weights (`weights=` takes a state-dict path instead of the
    reference's unconditional `EfficientNet.from_pretrained`, src/MuSCLe.py:165);
  * EfficientNet-B0 is accepted (taps by the same last-block-of-stage rule);
  * it only runs on a ROCm GPU: on CPU tensors forward() raises (there is no fallback path);
  * parameter gradients are written to `p.grad` by the module's own backward (they are views of
    one flat fp32 arena, which is what the DP all-reduce and the fused Adam consume).
`mode='dec'` builds the BiFPN decoder of train_muscle.py (muscle_amd/dec.py): forward(x, cam='seg'|'vis').
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn

from . import dec, engine, ops
from ._lib import MuscleHipError
from .arch import net_cfg
from .efficientnet import EfficientNet

_CPAD = 24      # class dimension (21) padded to a multiple of 4 for 16-byte rows


class _HeadTape:
    __slots__ = ("emb", "cam", "fs", "f", "fn", "nrm", "aff", "T", "hw", "hwp", "h", "w", "H", "W", "fcw", "mode")


_one_cache = {}


def _one_hot_bias(K: int, dev):
    """[_CPAD] bias with a 1 at column K; cached per device (a host scalar written per forward would be a blocking
    copy, and is not allowed while a stream is capturing)."""
    key = (K, str(dev))
    if key not in _one_cache:
        v = torch.zeros(_CPAD, dtype=torch.float32)
        v[K] = 1.0
        _one_cache[key] = v.to(dev)
    return _one_cache[key]


class _ArenaSink(engine.GradSink):
    """Gradient sink whose buffers are consecutive views of one flat, zero-filled fp32 arena."""

    def __init__(self, params: List[nn.Parameter]):
        super().__init__()
        self.touched = set()
        total = sum((p.numel() + 3) // 4 * 4 for p in params)
        self.arena = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        self.params = params
        self.off = {}
        self.on_ready = None          # callable(sink, lo): arena[lo:] is complete on the current stream (dist.GradAverager)
        off = 0
        for p in params:
            self.bufs[id(p)] = self.arena[off:off + p.numel()].view(p.shape)
            self.off[id(p)] = off
            off += (p.numel() + 3) // 4 * 4

    def of(self, p):
        self.touched.add(id(p))
        return self.bufs[id(p)]

    def ready_from(self, p):
        """Everything from parameter p to the end of the arena has been written (parameters are laid out in forward order
        and backward runs back to front): lets the data-parallel exchange start on that part while backward goes on."""
        if self.on_ready is not None and id(p) in self.off:
            self.on_ready(self, self.off[id(p)])

    def block_done(self, block_module):
        self.ready_from(next(block_module.parameters()))

    def publish(self, touched_only: bool = True):
        """Make p.grad a view of this arena for every parameter that received a gradient.  A gradient left from an earlier
        backward (no zero_grad in between) is added INTO the arena first, so that p.grad always aliases the arena the
        data-parallel average (dist.GradAverager), the clip (edge.clip_grad_norm_) and FusedAdam work on."""
        for p in self.params:
            if touched_only and id(p) not in self.touched:
                continue
            g = self.bufs[id(p)]
            if p.grad is not None and p.grad.data_ptr() != g.data_ptr():
                g.add_(p.grad)
            p.grad = g

    def check_aliases(self, model):
        """Raise if some parameter's gradient lives outside this arena (it would silently miss the all-reduce / clip)."""
        lo = self.arena.data_ptr()
        hi = lo + 4 * self.arena.numel()
        for k, p in model.named_parameters():
            if p.grad is not None and not (lo <= p.grad.data_ptr() < hi):
                raise MuscleHipError(
                    f"gradient of {k} is not a view of model.last_grad_sink.arena (a backward of another forward mode "
                    "left it behind, or it was assigned by hand): call zero_grad() before the backward whose gradients "
                    "are averaged / clipped")


class _Forward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, model, mode, drop_u):
        # grad mode is always off inside Function.forward and needs_input_grad ignores torch.no_grad(), so the module's
        # forward() records whether a backward can follow before it calls apply()
        need_grad = model._save_for_backward
        outs, tape, head = model._run_forward(x, mode, drop_u)
        ops.flush_batch_counters()               # BatchNorms of the decoder head
        if need_grad:
            ctx.model, ctx.tape, ctx.head, ctx.mode = model, tape, head, mode
        ctx.set_materialize_grads(False)      # unused outputs arrive as None instead of full-size zeros
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        model, tape, head, mode = ctx.model, ctx.tape, ctx.head, ctx.mode
        ctx.tape = ctx.head = None
        model._run_backward(tape, head, mode, gouts)
        return None, None, None, None, None


class MuSCLe(nn.Module):
    def __init__(self, num_classes, pretrained="efficientnet-b1", layers=1, MemoryEfficient=True, bifpn_channels=256,
                 last_pooling=True, mode="enc", weights: Optional[str] = None):
        super().__init__()
        self.classes = num_classes
        if num_classes > _CPAD - 1:
            raise ValueError(f"num_classes <= {_CPAD - 1} supported")
        self.cfg = net_cfg(pretrained, last_pooling)
        self.backbone = EfficientNet(self.cfg, num_classes)
        tc = self.cfg.tap_channels
        (self.p1_seq, self.p2_seq, self.p3_seq, self.p4_seq, self.p5_seq, self.p6_seq, self.p7_seq) = self.cfg.taps
        self.mode = mode
        if mode == "enc":
            self.fuse = nn.Conv2d(tc[0] + tc[2] + tc[4], 128, 1, bias=True)
            self.pool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(tc[6], num_classes, bias=False)
        else:
            self.layers = layers
            self.BIFPN = dec.BIFPN(tc, layers, bifpn_channels)
        self.fuse_dec = nn.Conv2d(bifpn_channels, num_classes, 1)
        self.logits = None
        self.grad_ready_callback = None     # dist.GradAverager.attach(): called as backward completes the arena back to front
        self._anchor = torch.zeros(1, requires_grad=True)      # makes autograd call our backward; not a parameter
        self.last_grad_sink: Optional[_ArenaSink] = None
        if weights is not None:
            sd = torch.load(weights, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=False)

    # ---- inference constants ------------------------------------------------------------------------------------------
    def fold_eval_bn(self):
        """Fold the eval-mode BatchNorms into the 1x1 convolution weights ONCE (CAM generation, infer_mcl.py:107-125: the
        model is loaded and only ever run in eval mode).  Without this call the no-grad eval forward folds per forward.
        The cache is dropped by train() and carries a fingerprint (storage address + in-place version of every source
        tensor): a forward that finds other weights underneath - loaded through a submodule, edited in place, moved to
        another device - refolds by itself."""
        self.backbone._eval_fold = engine.fold_eval_bn(self.backbone, self.cfg)
        return self

    def train(self, mode: bool = True):
        if mode and getattr(self.backbone, "_eval_fold", None) is not None:
            self.backbone._eval_fold = None
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        if getattr(self.backbone, "_eval_fold", None) is not None:
            self.backbone._eval_fold = None          # folded copies of the old weights
        return super().load_state_dict(*args, **kwargs)

    # ---- which parameters receive gradients (SURVEY.md §7 "DDP with unused parameters") ----------
    def live_parameters(self, cam_mode: str = "cam") -> List[nn.Parameter]:
        ps: List[nn.Parameter] = [self.backbone._conv_stem.weight, self.backbone._bn0.weight, self.backbone._bn0.bias]
        last = self.cfg.taps[6]
        for blk in self.backbone._blocks[:last + 1]:
            ps += list(blk.parameters())
        if self.mode != "enc":
            # candidates; the dead branches of the last BiFPN layer (out4..out7) never get touched and keep grad None
            return ps + list(self.BIFPN.parameters()) + [self.fuse_dec.weight, self.fuse_dec.bias]
        ps += [self.fuse.weight, self.fuse.bias]
        if cam_mode in ("cam", "logits", "cam_lr"):
            ps.append(self.fc.weight)
        if cam_mode == "logits":
            ps = [p for p in ps if p is not self.fuse.weight and p is not self.fuse.bias]
        return ps

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, x, cam="cam", drop_u: Optional[Dict[int, torch.Tensor]] = None):
        if cam in ("seg", "vis", "seg_p3"):
            if self.mode == "enc":
                raise AttributeError("forward(cam='seg') needs MuSCLe(mode='dec') (the encoder model has no BIFPN)")
        elif cam in ("logits", "cam", "pix", "cam_lr"):
            if self.mode != "enc":
                raise AttributeError(f"forward(cam={cam!r}) needs MuSCLe(mode='enc') (the decoder model has no fc / fuse)")
        else:
            raise ValueError(f"unknown cam mode {cam!r}")
        if not x.is_cuda:
            raise MuscleHipError("MuSCLe.forward runs on the HIP kernels only: move the model and input to a ROCm GPU")
        x = x.contiguous().float()
        self._save_for_backward = torch.is_grad_enabled() and cam != "vis"
        if cam == "vis":                                   # MuSCLe.py:290-298: no_grad seg forward, returns (seg_map, p7)
            with torch.no_grad():
                seg_map, p7 = _Forward.apply(x, self._anchor, self, "vis", drop_u)
            return seg_map, p7
        outs = _Forward.apply(x, self._anchor, self, cam, drop_u)
        if cam == "logits":
            self.logits = outs[1]
            return outs[0], outs[1]
        if cam in ("cam", "cam_lr"):
            self.logits = outs[3]
            return outs[0], outs[1], outs[2], outs[3]
        return outs[0], outs[1]

    # ---- decoder mode (MuSCLe.py:281-298) ------------------------------------------------------------
    def _fuse_dec_padded(self, dev):
        C = self.fuse_dec.in_channels
        w = torch.zeros(_CPAD, C, dtype=torch.float32, device=dev)
        w[:self.classes] = self.fuse_dec.weight.detach().view(self.classes, C)
        b = torch.zeros(_CPAD, dtype=torch.float32, device=dev)
        b[:self.classes] = self.fuse_dec.bias.detach()
        return w, b

    def _run_forward_dec(self, x, mode, drop_u):
        cfg, K = self.cfg, self.classes
        N, _, H, W = x.shape
        tape = engine.backbone_forward(self.backbone, cfg, x, self.training, drop_u, save=getattr(self, "_save_for_backward", True))
        t = cfg.taps
        feats = [tape.blocks[i].out for i in t[2:7]]
        tp = dec.Tape(None, self.training)
        p3 = dec.bifpn_forward(tp, self.BIFPN, feats, cfg.last_pooling)
        _, h, w, C = p3.shape
        w24, b24 = self._fuse_dec_padded(x.device)
        # fuse_dec is linear and the bilinear weights sum to one, so conv(upsample(p3)) == upsample(conv(p3)):
        # the 1x1 conv runs at 1/8 resolution and only its 21-channel result is upsampled
        seg_lr = ops.pw_fwd(p3.view(N * h * w, C), w24, _CPAD, bias=b24).view(N, h, w, _CPAD)
        ht = _HeadTape()
        ht.h, ht.w, ht.H, ht.W, ht.mode, ht.fcw = h, w, H, W, mode, w24
        ht.f, ht.T, ht.fs = tp, seg_lr, feats           # reuse slots: BiFPN tape, low-res logits, tap tensors
        ht.cam = p3
        seg_map = ops.upsample_to_nchw(seg_lr, K, H, W)
        if mode == "vis":
            return (seg_map, tape.blocks[t[6]].out.permute(0, 3, 1, 2).contiguous()), tape, ht
        if mode == "seg_p3":
            return (seg_map, p3), tape, ht
        dense_ft = ops.upsample_to_nchw(p3, C, H, W)
        return (seg_map, dense_ft), tape, ht

    def _run_backward_dec(self, tape, ht, mode, gouts):
        cfg, K = self.cfg, self.classes
        N = tape.N
        g_seg, g_ft = gouts
        tp, seg_lr, feats, p3 = ht.f, ht.T, ht.fs, ht.cam
        _, h, w, C = p3.shape
        M3 = N * h * w
        dev = p3.device
        sink = _ArenaSink(self.live_parameters(mode))
        sink.on_ready = getattr(self, "grad_ready_callback", None)
        tp.sink = sink
        g_p3 = None
        if g_seg is not None:
            g_lr = torch.zeros(N, h, w, _CPAD, dtype=torch.float32, device=dev)
            ops.upsample_to_nchw_bwd(g_seg.contiguous(), g_lr)
            g2 = g_lr.view(M3, _CPAD)
            dW = torch.zeros(_CPAD, C, dtype=torch.float32, device=dev)
            ops.pw_wgrad(g2, p3.view(M3, C), dW)
            sink.of(self.fuse_dec.weight).view(K, C).add_(dW[:K])
            sink.of(self.fuse_dec.bias).add_(ops.pool_sum(g2, M3).view(_CPAD)[:K])
            g_p3 = ops.pw_dgrad(g2, ht.fcw, C).view(N, h, w, C)
        if g_ft is not None:
            if mode == "seg_p3":
                gp = g_ft.contiguous()
            else:
                gp = torch.zeros(N, h, w, C, dtype=torch.float32, device=dev)
                ops.upsample_to_nchw_bwd(g_ft.contiguous(), gp)
            g_p3 = gp if g_p3 is None else ops.ew(1, g_p3, gp, alpha=1.0)
        if g_p3 is not None:
            tp.add_grad(p3, g_p3)
            tp.run_backward()
            tap_grads = {}
            for i, f in zip(cfg.taps[2:7], feats):
                g = tp.grad(f)
                if g is not None:
                    tap_grads[i] = g
            # backbone parameters are always touched once a gradient reaches the chain
            sink.ready_from(next(self.BIFPN.parameters()))                                  # decoder + fuse_dec gradients are in
            engine.backbone_backward(self.backbone, cfg, tape, tap_grads, sink)
        sink.publish()
        self.last_grad_sink = sink

    def _run_forward(self, x, mode, drop_u):
        if mode in ("seg", "vis", "seg_p3"):
            return self._run_forward_dec(x, mode, drop_u)
        cfg, K = self.cfg, self.classes
        N, _, H, W = x.shape
        dev = x.device
        tape = engine.backbone_forward(self.backbone, cfg, x, self.training, drop_u, save=getattr(self, "_save_for_backward", True))
        t = cfg.taps
        p1, p3, p5, p7 = (tape.blocks[i].out for i in (t[0], t[2], t[4], t[6]))
        _, h, w, C7 = p7.shape
        hw, M7 = h * w, N * h * w
        ht = _HeadTape()
        ht.h, ht.w, ht.hw, ht.hwp, ht.H, ht.W, ht.mode = h, w, hw, (hw + 3) // 4 * 4, H, W, mode
        fcw = torch.zeros(_CPAD, C7, dtype=torch.float32, device=dev)
        fcw[:K] = self.fc.weight.detach()
        ht.fcw = fcw
        p7m = p7.view(M7, C7)
        emb = logits = None
        if mode in ("cam", "logits", "cam_lr"):
            emb = ops.ew(0, ops.pool_sum(p7m, hw), alpha=1.0 / hw)                      # GAP, MuSCLe.py:240
            logits = ops.pw_fwd(emb, fcw, K)                                             # fc (no bias), :241
            ht.emb = emb
            if mode == "logits":
                return (emb, logits), tape, ht
        # CAM = relu(1x1 conv of p7 with the detached fc weight), :243-247.  Column K of the padded class
        # dimension is forced to 1 through the bias so that aff @ [cam|1] also yields the affinity row sums.
        one = _one_hot_bias(K, dev)
        cam = torch.zeros(M7 + 4, _CPAD, dtype=torch.float32, device=dev)               # +4 rows: padded-K reads
        ops.pw_fwd(p7m, fcw, _CPAD, bias=one, relu=True, out=cam, ldc=_CPAD)
        ht.cam = cam
        # fs = cat(relu(resize(p1)), relu(resize(p3)), relu(p5)) under no_grad, :248-252
        c1, c3, c5 = p1.shape[3], p3.shape[3], p5.shape[3]
        fs = torch.empty(N, h, w, c1 + c3 + c5, dtype=torch.float32, device=dev)
        ops.resize_nhwc(p1, fs, 0)
        ops.resize_nhwc(p3, fs, c1)
        ops.resize_nhwc(p5, fs, c1 + c3)
        ht.fs = fs
        # PCM, :213-223
        Cf = c1 + c3 + c5
        f = ops.pw_fwd(fs.view(M7, Cf), self.fuse.weight.view(128, Cf), 128, bias=self.fuse.bias)
        fnb = torch.zeros(M7 + 4, 128, dtype=torch.float32, device=dev)
        fn = fnb[:M7]
        nrm = torch.empty(M7, dtype=torch.float32, device=dev)
        ops.call("mx_row_l2norm", ops.ptr(f), ops.ptr(fn), ops.ptr(nrm), M7, 128, 1e-5, ops.stream())
        ht.f, ht.fn, ht.nrm = f, fnb, nrm
        aff = torch.zeros(N, hw, ht.hwp, dtype=torch.float32, device=dev)
        fn3 = fn.view(N, hw, 128)
        ops.bgemm(0, fn3, fn3, aff, hw, hw, 128, relu=True)                               # relu(f^T f)
        T = torch.empty(N, hw, _CPAD, dtype=torch.float32, device=dev)
        ops.bgemm(1, aff, cam[:M7].view(N, hw, _CPAD), T, hw, _CPAD, ht.hwp)              # aff @ [cam | 1]
        ht.aff, ht.T = aff, T
        rv = ops.pcm_norm(T, torch.empty_like(T), K, 1e-5)                                # column-normalised aff applied
        if mode == "cam_lr":
            # internal mode of mcl_step: the low-resolution NHWC maps [N,h,w,24]; their bilinear upsampling
            # (:256-257) is folded into the fused ER kernel, so the 2 x [N,21,H,W] tensors are never written
            return (cam[:M7].view(N, h, w, _CPAD), rv.view(N, h, w, _CPAD), emb, logits), tape, ht
        cams = ops.upsample_to_nchw(cam[:M7].view(N, h, w, _CPAD), K, H, W)               # :256
        sgc = ops.upsample_to_nchw(rv.view(N, h, w, _CPAD), K, H, W)                      # :257
        if mode == "pix":
            return (cams, sgc), tape, ht
        return (cams, sgc, emb, logits), tape, ht

    # ---- backward -----------------------------------------------------------------------------------
    def _run_backward(self, tape, ht, mode, gouts):
        if mode in ("seg", "seg_p3"):
            return self._run_backward_dec(tape, ht, mode, gouts)
        cfg, K = self.cfg, self.classes
        N = tape.N
        p7 = tape.blocks[cfg.taps[6]].out
        _, h, w, C7 = p7.shape
        hw, M7 = ht.hw, N * ht.hw
        dev = p7.device
        sink = _ArenaSink(self.live_parameters(mode))
        sink.on_ready = getattr(self, "grad_ready_callback", None)
        if mode == "logits":
            g_cams = g_sgc = None
            g_emb, g_logits = gouts
        elif mode in ("cam", "cam_lr"):
            g_cams, g_sgc, g_emb, g_logits = gouts
        else:
            (g_cams, g_sgc), g_emb, g_logits = gouts, None, None
        g_p7 = None
        if mode != "logits":
            g_cam = torch.zeros(M7 + 4, _CPAD, dtype=torch.float32, device=dev)          # dL/d cam (low res), pre-relu mask
            have = False
            if g_cams is not None:
                if mode == "cam_lr":
                    g_cam[:M7].copy_(g_cams.reshape(M7, _CPAD))
                else:
                    ops.upsample_to_nchw_bwd(g_cams.contiguous(), g_cam[:M7].view(N, h, w, _CPAD))
                have = True
            if g_sgc is not None:
                if mode == "cam_lr":
                    g_rv = g_sgc.contiguous().view(N, hw, _CPAD)
                else:
                    g_rv = torch.zeros(N, hw, _CPAD, dtype=torch.float32, device=dev)
                    ops.upsample_to_nchw_bwd(g_sgc.contiguous(), g_rv.view(N, h, w, _CPAD))
                gTb = torch.zeros(M7 + 4, _CPAD, dtype=torch.float32, device=dev)
                gT = gTb[:M7].view(N, hw, _CPAD)
                ops.pcm_norm(ht.T, gT, K, 1e-5, grv=g_rv)
                cam3 = ht.cam[:M7].view(N, hw, _CPAD)
                g_aff = torch.zeros_like(ht.aff)
                ops.bgemm(0, gT, cam3, g_aff, hw, hw, _CPAD)                                # dT camx^T
                g_camx = torch.empty(N, hw, _CPAD, dtype=torch.float32, device=dev)
                ops.bgemm(1, ht.aff, gT, g_camx, hw, _CPAD, ht.hwp)                         # aff^T dT (aff symmetric)
                ops.ew(1, g_cam[:M7], g_camx.view(M7, _CPAD), alpha=1.0, out=g_cam[:M7])
                have = True
                G2 = ops.sym_relu_grad(g_aff, ht.aff, hw)
                del g_aff
                g_fn = torch.empty(N, hw, 128, dtype=torch.float32, device=dev)
                ops.bgemm(1, G2, ht.fn[:M7].view(N, hw, 128), g_fn, hw, 128, ht.hwp)
                del G2
                g_f = ops.row_l2norm_bwd(ht.f, ht.nrm, g_fn.view(M7, 128), 1e-5)
                Cf = ht.fs.shape[3]
                ops.pw_wgrad(g_f, ht.fs.view(M7, Cf), sink.of(self.fuse.weight).view(128, Cf))
                sink.of(self.fuse.bias).add_(ops.pool_sum(g_f, M7).view(128))
            if have:
                g_cam_m = ops.ew(2, g_cam[:M7], y=ht.cam[:M7])                              # relu backward
                g_p7 = ops.pw_dgrad(g_cam_m, ht.fcw, C7)                                    # through the detached fc weight
        if mode in ("cam", "logits", "cam_lr") and (g_emb is not None or g_logits is not None):
            g_e = g_emb.contiguous() if g_emb is not None else None
            if g_logits is not None:
                gl = torch.zeros(N, _CPAD, dtype=torch.float32, device=dev)
                gl[:, :K] = g_logits
                dW = torch.zeros(_CPAD, C7, dtype=torch.float32, device=dev)
                ops.pw_wgrad(gl, ht.emb, dW)
                sink.of(self.fc.weight).add_(dW[:K])
                g_e = ops.pw_dgrad(gl, ht.fcw, C7, residual=g_e)
            if g_p7 is None:
                g_p7 = torch.zeros(M7, C7, dtype=torch.float32, device=dev)
            ops.bcast_add(g_p7, g_e, 1.0 / hw, hw)                                          # GAP backward
        if g_p7 is not None:
            sink.ready_from(sink.params[-1] if mode == "logits" else self.fuse.weight)      # the head gradients are in
            engine.backbone_backward(self.backbone, cfg, tape, {cfg.taps[6]: g_p7.view(N, h, w, C7)}, sink)
        # only parameters a kernel wrote a gradient for get one (a head whose output was unused keeps grad None, as under
        # autograd: torch.optim.Adam / FusedAdam skip it, no weight decay, no step count)
        sink.publish()
        self.last_grad_sink = sink
