"""Thin tensor-level wrappers over the C ABI (include/muscle_hip.h).

Every function takes contiguous CUDA fp32 tensors in the library's NHWC / [rows, C] convention,
allocates outputs with torch (device memory is torch's job, arithmetic is the library's) and
enqueues on torch's current stream.
"""
from __future__ import annotations

import os
from typing import NamedTuple, Optional

import torch

from ._lib import call, lib, ptr, stream

PLAIN, BNACT, AFFINE = 0, 1, 2


class BNState(NamedTuple):
    scale: torch.Tensor
    shift: torch.Tensor
    mean: torch.Tensor
    rstd: torch.Tensor


def _f32(*shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


# Scratch for the entry points that take `ws` (include/muscle_hip.h): one persistent, zero-initialised buffer per
# (device, stream) - the arrival counters in its first 64 KB are left zero by every launch, so it can be handed to one call
# after the other.  A buffer that has to grow is REPLACED but never freed: a captured hipGraph has the old pointer baked in.
_scratch_bufs: dict = {}
_scratch_retired: list = []


def _scratch(device, nbytes):
    """(pointer, size) of this stream's scratch, at least nbytes large; (None, 0) when nbytes == 0."""
    if nbytes <= 0:
        return None, 0
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream())
    buf = _scratch_bufs.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _scratch_retired.append(buf)
        buf = torch.zeros(max(int(nbytes), 64 << 20), dtype=torch.uint8, device=device)
        _scratch_bufs[key] = buf
    return buf.data_ptr(), buf.numel()


def _call_ws(name, *args):
    """call() for the entry points whose scratch starts with arrival counters that every launch leaves at zero: if the call fails
    (an argument error before the launch, or a launch error), the counters of this stream's scratch are re-zeroed, so a failed
    call cannot poison the ordered reductions that use the buffer next."""
    try:
        call(name, *args)
    except Exception:
        key = (torch.cuda.current_device(), stream())
        buf = _scratch_bufs.get(key)
        if buf is not None:
            buf[:65536].zero_()
        raise


# ---- GEMMs ------------------------------------------------------------------------------------
_uses_planes: dict = {}


def _planes_take(M, K, N_out):
    """Does a plain-A GEMM of this shape go to the second-generation split kernel in the current arithmetic?  (cached per mode)"""
    key = (get_gemm_mode(), M, K, N_out)
    r = _uses_planes.get(key)
    if r is None:
        r = _uses_planes[key] = bool(lib().mx_pw_fwd_uses_planes(M, K, N_out))
    return r


def _planes_act_take(M, K, N_out):
    """The same question for the project convolution's activated-input form (mx_pw_fwd_planes_act)."""
    key = ("act", get_gemm_mode(), M, K, N_out)
    r = _uses_planes.get(key)
    if r is None:
        r = _uses_planes[key] = bool(lib().mx_pw_fwd_act_uses_planes(M, K, N_out))
    return r


def pw_fwd(A, W, N_out, *, a_mode=PLAIN, a_scale=None, a_shift=None, a_gate=None, rows_per_sample=1,
           bias=None, residual=None, relu=False, want_stats=False, out=None, ldc=None, planes=None):
    """A: [M, K]; W: [N_out, K] -> [M, N_out] (and, if want_stats, the partial BN statistics [P, 2, N_out]).
    planes: the pre-split image of W (WeightPlan), used when A is plain and the shape runs in split arithmetic."""
    M, K = A.shape
    ldc = ldc or N_out
    if out is None:
        out = _f32(M, ldc, device=A.device)
    stats = None
    if want_stats:
        stats = _f32(lib().mx_pw_fwd_parts(M, N_out, K), 2, N_out, device=A.device)
    if planes is not None and a_mode == PLAIN and _planes_take(M, K, N_out):
        call("mx_pw_fwd_planes", ptr(A), planes, ptr(out), M, K, N_out, A.stride(0), ldc, ptr(bias), ptr(residual), int(relu),
             ptr(stats), stream())
    elif (planes is not None and a_mode == BNACT and a_gate is not None and bias is None and residual is None and not relu
          and _planes_act_take(M, K, N_out)):
        call("mx_pw_fwd_planes_act", ptr(A), ptr(a_scale), ptr(a_shift), ptr(a_gate), rows_per_sample, planes, ptr(out), M, K, N_out,
             A.stride(0), ldc, ptr(stats), stream())
    else:
        call("mx_pw_fwd", ptr(A), a_mode, ptr(a_scale), ptr(a_shift), ptr(a_gate), rows_per_sample, ptr(W), ptr(out),
             M, K, N_out, A.stride(0), ldc, ptr(bias), ptr(residual), int(relu), ptr(stats), stream())
    return (out, stats) if want_stats else out


def set_gemm_mode(mode: int):
    """Arithmetic of the pointwise-conv GEMMs (include/muscle_hip.h, mx_set_gemm_mode): 0 = exact-fp32 MFMA everywhere;
    1 (the library's default) = fp32 operands split exactly into three bf16 terms, six products on the bf16 matrix pipe with
    fp32 accumulation, for the MFMA-bound shapes; 2 = the same for every NT GEMM (tests)."""
    call("mx_set_gemm_mode", int(mode))


def get_gemm_mode() -> int:
    return int(lib().mx_get_gemm_mode())


def set_wgrad_kernel(kernel: int = -1, groups: int = -1):
    """Which kernel takes the plain-operand split weight gradients (include/muscle_hip.h, mx_set_wgrad_kernel): 0 = the first split
    kernel, 1 = the single-stream pipeline, 2 = the wave-specialised persistent kernel (default); `groups` > 0 fixes their row groups
    (0 = planner).  -1 leaves a setting as it is.  Tests and measurement only."""
    call("mx_set_wgrad_kernel", int(kernel), int(groups))


def get_wgrad_kernel() -> int:
    return int(lib().mx_get_wgrad_kernel())


class TransposePlan:
    """Persistent W^T buffers of a fixed set of 2-D weights and the device table that transposes them all in one launch."""

    def __init__(self, weights):
        dev = weights[0].device
        self.key = tuple((w.data_ptr(), tuple(w.shape)) for w in weights)
        self.dst = [_f32(w.shape[1], w.shape[0], device=dev) for w in weights]
        rows, tile = [], 0
        for w, d in zip(weights, self.dst):
            r, c = w.shape
            rows.append([w.data_ptr(), d.data_ptr(), r, c, tile])
            tile += ((r + 31) // 32) * ((c + 31) // 32)
        self.n, self.tiles = len(rows), tile
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        torch.cuda.current_stream(dev).synchronize()     # built once: the table is in place whichever stream runs the launch

    def matches(self, weights):
        return len(weights) == self.n and all(k[0] == w.data_ptr() and k[1] == tuple(w.shape) for k, w in zip(self.key, weights))

    def run(self):
        call("mx_transpose_batch", ptr(self.table), self.n, self.tiles, stream())
        return self.dst


class PlanesPlan:
    """Pre-split bf16 images (mx_pw_planes_batch) of a fixed set of 2-D fp32 matrices [N, K]: one buffer, one launch.
    images[i] is the device address of matrix i's image, or None where the shape has none (K % 32 != 0)."""

    def __init__(self, mats):
        dev = mats[0].device
        L = lib()
        rows, tile, off = [], 0, 0
        sizes = []
        for m in mats:
            n, k = m.shape
            nb = L.mx_pw_planes_bytes(n, k) if m.is_contiguous() else -1
            sizes.append(nb)
            if nb > 0:
                off += (nb + 1023) & ~1023
        self.buf = torch.empty(max(off, 1024), dtype=torch.uint8, device=dev)
        base, off = self.buf.data_ptr(), 0
        self.images = []
        for m, nb in zip(mats, sizes):
            if nb <= 0:
                self.images.append(None)
                continue
            n, k = m.shape
            rows.append([m.data_ptr(), base + off, n, k, tile])
            self.images.append(base + off)
            tile += L.mx_pw_planes_tiles(n, k)
            off += (nb + 1023) & ~1023
        self.n, self.tiles = len(rows), tile
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev) if rows else None
        torch.cuda.current_stream(dev).synchronize()     # built once: the table is in place whichever stream runs the launch

    def run(self):
        if self.n:
            call("mx_pw_planes_batch", ptr(self.table), self.n, self.tiles, stream())
        return self.images


class WeightPlan:
    """What the GEMMs of a step derive from the 1x1 conv weights, rebuilt once per step in three launches: W^T for the data
    gradients (TransposePlan), the pre-split images of W (forward) and of W^T (data gradient) for the second-generation
    split kernel."""

    def __init__(self, weights):
        self.t = TransposePlan(weights)
        self.key = self.t.key
        self.fwd = PlanesPlan(weights)
        self.bwd = PlanesPlan(self.t.dst)

    def matches(self, weights):
        return self.t.matches(weights)

    def run_forward(self):
        """images of W, for the forward GEMMs (enqueue on the stream that runs them)"""
        return self.fwd.run()

    def run_backward(self):
        """(W^T list, images of W^T): nothing before the backward reads them - the side stream is the place"""
        wt = self.t.run()
        return wt, self.bwd.run()


def transpose(W):
    """[rows, cols] -> contiguous [cols, rows] (weights)."""
    rows, cols = W.shape
    out = _f32(cols, rows, device=W.device)
    call("mx_transpose", ptr(W), ptr(out), rows, cols, stream())
    return out


DGRAD_AS_FORWARD = True      # dX = G W as a forward GEMM against W^T (row-major LDS images, 16-column tiles); False: NN kernel


def pw_dgrad(G, W, N_in, *, residual=None, out=None, wt=None, planes=None):
    """G: [M, K=Cout]; W: [Cout, Cin] -> dX [M, Cin].  wt: W^T [Cin, Cout] if the caller already has it, planes: its pre-split image."""
    M, K = G.shape
    if out is None:
        out = _f32(M, N_in, device=G.device)
    if DGRAD_AS_FORWARD and K % 4 == 0:
        if planes is not None and _planes_take(M, K, N_in):
            call("mx_pw_fwd_planes", ptr(G), planes, ptr(out), M, K, N_in, G.stride(0), N_in, None, ptr(residual), 0, None, stream())
            return out
        call("mx_pw_fwd", ptr(G), PLAIN, None, None, None, 1, ptr(wt if wt is not None else transpose(W)), ptr(out), M, K, N_in, G.stride(0), N_in,
             None, ptr(residual), 0, None, stream())
        return out
    call("mx_pw_dgrad", ptr(G), ptr(W), ptr(out), M, K, N_in, G.stride(0), N_in, ptr(residual), stream())
    return out


def pw_dgrad_bnbwd(G, X, coef, W, N_in, *, residual=None, wt=None):
    """The BatchNorm backward apply dZ = c1*G + c2*X + c3 (coef [3, K], as bn_bwd_coeffs returns it) folded into the data
    gradient dX = dZ W: returns (dX [M, N_in], dZ [M, K]) - dZ is materialised by the GEMM for the weight gradient."""
    M, K = G.shape
    out = _f32(M, N_in, device=G.device)
    dz = torch.empty_like(G)
    call("mx_pw_dgrad_bnbwd", ptr(G), ptr(X), ptr(coef), ptr(wt if wt is not None else transpose(W)), ptr(out), ptr(dz), M, K, N_in,
         G.stride(0), N_in, ptr(residual), stream())
    return out, dz


def pw_dgrad_bnbwd_planes(G, X, coef, planes, N_in, *, residual=None):
    """dX = dZ W with dZ = c1*G + c2*X + c3 formed in the GEMM's operand load (second-generation split kernel, W^T given as its
    pre-split image); dZ is not materialised - the weight gradient folds the same expression (pw_wgrad_bnbwd)."""
    M, K = G.shape
    out = _f32(M, N_in, device=G.device)
    call("mx_pw_dgrad_bnbwd_planes", ptr(G), ptr(X), ptr(coef), planes, ptr(out), M, K, N_in, G.stride(0), N_in, ptr(residual), stream())
    return out


def bnbwd_fold_takes(M, K, N):
    """Which weight-gradient kernel can fold the BatchNorm backward apply for dZ [M, K] (data gradient against W^T [N, K], weight
    gradient dW [K, N]) in the current arithmetic, given that the data gradient takes the planes kernel: "small" (the HBM-bound
    small-output kernel of stages 1-2), "tile" (the split-arithmetic tiled kernel) or None."""
    key = ("fold", get_gemm_mode(), M, K, N)
    r = _uses_planes.get(key, 0)
    if r == 0:
        r = None
        if _planes_take(M, K, N):
            if lib().mx_pw_wgrad_small_bnbwd_ok(M, K, N):
                r = "small"
            elif lib().mx_pw_wgrad_small_ws(M, K, N, PLAIN) <= 0 and lib().mx_pw_wgrad_tile_bnbwd_ok(M, K, N):
                r = "tile"
        _uses_planes[key] = r
    return r


def pw_wgrad_bnbwd(G, G2, coef, X, dW):
    """dW[Co, Ci] += (c1*G + c2*G2 + c3)[R, Co]^T X[R, Ci] through the kernel bnbwd_fold_takes names."""
    R, Co = G.shape
    Ci = X.shape[1]
    if lib().mx_pw_wgrad_small_bnbwd_ok(R, Co, Ci):
        need = lib().mx_pw_wgrad_small_ws(R, Co, Ci, PLAIN)
        ws = _wgrad_workspace(G.device, max(need, 16))
        call("mx_pw_wgrad_small_bnbwd", ptr(G), ptr(G2), ptr(coef), ptr(X), ptr(dW), R, Co, Ci, G.stride(0), X.stride(0), ws.data_ptr(), ws.numel(), stream())
        return
    need = lib().mx_pw_wgrad_tile_ws(R, Co, Ci, PLAIN)
    ws = _wgrad_workspace(G.device, max(need, 16))
    call("mx_pw_wgrad_tile_bnbwd", ptr(G), ptr(G2), ptr(coef), ptr(X), ptr(dW), R, Co, Ci, G.stride(0), X.stride(0), ws.data_ptr(), ws.numel(), stream())


def wgrad_bnbwd_dz_takes(R, Co, Ci) -> bool:
    """True where `pw_wgrad_bnbwd_dz` exists: the wave-specialised split weight-gradient kernel takes dW [Co, Ci] over R rows."""
    key = ("dzfold", get_gemm_mode(), get_wgrad_kernel(), R, Co, Ci)
    r = _uses_planes.get(key)
    if r is None:
        r = _uses_planes[key] = bool(lib().mx_pw_wgrad_tile_bnbwd_dz_ok(R, Co, Ci, Co, Ci))
    return r


def pw_wgrad_bnbwd_dz(G, G2, coef, X, dW, dz=None):
    """dW[Co, Ci] += dZ^T X with dZ = c1*G + c2*G2 + c3 formed AND stored by the weight-gradient kernel's loader waves (round 5):
    returns dZ [R, Co] for the data gradient that follows; mx_bn_bwd_apply is not launched."""
    R, Co = G.shape
    Ci = X.shape[1]
    if dz is None:
        dz = torch.empty_like(G)
    need = lib().mx_pw_wgrad_tile_ws(R, Co, Ci, PLAIN)
    ws = _wgrad_workspace(G.device, max(need, 16))
    call("mx_pw_wgrad_tile_bnbwd_dz", ptr(G), ptr(G2), ptr(coef), ptr(X), ptr(dW), ptr(dz), R, Co, Ci, G.stride(0), X.stride(0),
         ws.data_ptr(), ws.numel(), stream())
    return dz


_wgrad_ws: dict = {}
_wgrad_ws_retired: list = []      # outgrown workspaces stay alive: captured graphs keep writing to them
WGRAD_TILE = os.environ.get("MUSCLE_WGRAD_TILE", "1") == "1"   # large outputs: tiled deterministic kernel (wgrad.hip) instead of the atomic TN GEMM
WGRAD_SMALL = True       # small outputs / long reductions: one workgroup owns the whole output, deterministic partial sums


def _wgrad_workspace(device, nbytes):
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream())
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _wgrad_ws_retired.append(ws)
        ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _wgrad_ws[key] = ws
    return ws


def pw_wgrad(G, X, dW, *, x_mode=PLAIN, x_scale=None, x_shift=None, x_gate=None, rows_per_sample=1):
    """dW[Co, Ci] += G[R, Co]^T X'[R, Ci]."""
    R, Co = G.shape
    Ci = X.shape[1]
    if WGRAD_SMALL and dW.is_contiguous():
        need = lib().mx_pw_wgrad_small_ws(R, Co, Ci, x_mode)
        if need > 0:
            ws = _wgrad_workspace(G.device, need)
            call("mx_pw_wgrad_small", ptr(G), ptr(X), x_mode, ptr(x_scale), ptr(x_shift), ptr(x_gate), rows_per_sample, ptr(dW),
                 R, Co, Ci, G.stride(0), X.stride(0), ws.data_ptr(), ws.numel(), stream())
            return
    if WGRAD_TILE and dW.is_contiguous():
        need = lib().mx_pw_wgrad_tile_ws(R, Co, Ci, x_mode)
        if need > 0:
            ws = _wgrad_workspace(G.device, need)
            call("mx_pw_wgrad_tile", ptr(G), ptr(X), x_mode, ptr(x_scale), ptr(x_shift), ptr(x_gate), rows_per_sample, ptr(dW),
                 R, Co, Ci, G.stride(0), X.stride(0), ws.data_ptr(), ws.numel(), stream())
            return
    need = lib().mx_pw_wgrad_ws(R, Co, Ci)
    ws = _wgrad_workspace(G.device, need) if need > 0 else None
    call("mx_pw_wgrad", ptr(G), ptr(X), x_mode, ptr(x_scale), ptr(x_shift), ptr(x_gate), rows_per_sample, ptr(dW),
         R, Co, Ci, G.stride(0), X.stride(0), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, stream())


def bgemm(layout, A, B, out, M, N, K, *, relu=False):
    """Batched plain GEMM on 3-D views [b, rows, ld]; extents explicit because operands may be padded.
    layout 0: C[M,N] = A[M,K] B[N,K]^T; layout 1: C[M,N] = A[M,K] B[K,N]."""
    b = A.shape[0]
    call("mx_bgemm", layout, ptr(A), ptr(B), ptr(out), M, N, K, A.stride(1), B.stride(1), out.stride(1),
         A.stride(0), B.stride(0), out.stride(0), b, int(relu), stream())
    return out


# ---- BatchNorm / elementwise ---------------------------------------------------------------------
def colstats(X2d):
    rows, C = X2d.shape
    part = _f32(lib().mx_colreduce_parts(rows, C), 2, C, device=X2d.device)
    call("mx_colstats", ptr(X2d), rows, C, ptr(part), stream())
    return part


def bn_finalize(stats, count, bn: torch.nn.BatchNorm2d, training: bool) -> BNState:
    """stats: partial rows [P, 2, C] from a producer (None in eval mode)."""
    C = bn.num_features
    dev = bn.weight.device
    buf = _f32(4, C, device=dev)
    b0, b1, b2, b3 = buf.unbind(0)                            # (one op instead of four indexing ops: this runs ~160 times per step)
    base = buf.data_ptr()
    mom = 0.0 if bn.momentum is None else float(bn.momentum)
    call("mx_bn_finalize", ptr(stats), stats.shape[0] if stats is not None else 0, C, float(count), ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
         ptr(bn.running_var), mom, float(bn.eps), int(training), base, base + 4 * C, base + 8 * C, base + 12 * C,
         _acc_scratch(dev, 128 * C) if training else None, stream())
    if training:
        _nbt_pending.append(bn.num_batches_tracked)
    return BNState(b0, b1, b2, b3)


def fold_block(We, bn0, bn1, Wp, bn2, out=None):
    """Eval-mode BatchNorm folded into one block's 1x1 convolutions (mx_fold_block): dict We / be / bn1 / Wp / bp.
    out: a dict returned by an earlier call for the same block - its tensors are overwritten instead of allocated."""
    dev = Wp.device
    cout, cexp = Wp.shape
    cin = We.shape[1] if We is not None else cexp
    if out is not None:
        f, vec = out, out["_vec"]
    else:
        f = {}
        vec = f["_vec"] = _f32(3, cexp, device=dev)
        if We is not None:
            f["We"], f["be"] = _f32(cexp, cin, device=dev), vec[0]
        f["bn1"] = BNState(vec[1], vec[2], None, None)
        f["Wp"], f["bp"] = _f32(cout, cexp, device=dev), _f32(cout, device=dev)

    def four(bn):
        return (ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), float(bn.eps))
    none4 = (None, None, None, None, 0.0)
    call("mx_fold_block", ptr(We.detach()) if We is not None else None, *(four(bn0) if We is not None else none4), *four(bn1),
         ptr(Wp.detach()), *four(bn2), cin, cexp, cout, ptr(f.get("We")), ptr(f.get("be")), ptr(vec[1]), ptr(vec[2]),
         ptr(f["Wp"]), ptr(f["bp"]), stream())
    return f


_acc_bufs: dict = {}


def _acc_scratch(dev, n):
    """fp64 scratch acc[64][2C] of the two-level statistics reduction (only touched when a layer has more than 1024 partial
    rows): one persistent buffer per (device, stream) instead of an allocation per BatchNorm; outgrown ones are kept."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), stream())
    t = _acc_bufs.get(key)
    if t is None or t.numel() < n:
        if t is not None:
            _scratch_retired.append(t)
        t = _acc_bufs[key] = torch.empty(max(n, 128 * 4096), dtype=torch.float64, device=dev)
    return t.data_ptr()


_nbt_pending: list = []


def flush_batch_counters():
    """num_batches_tracked += 1 for every BatchNorm finalised in train mode since the last flush, as one multi-tensor
    launch (162 separate int64 add kernels per B7 forward otherwise)."""
    if _nbt_pending:
        torch._foreach_add_(_nbt_pending, 1)
        _nbt_pending.clear()


def bn_apply(P2d, st: BNState, *, row_scale=None, residual=None, gate=None, rows_per_sample=1, act=False, out=None):
    rows, C = P2d.shape
    if out is None:
        out = torch.empty_like(P2d)
    call("mx_bn_apply", ptr(P2d), ptr(st.scale), ptr(st.shift), ptr(row_scale), ptr(residual), ptr(gate), ptr(out), rows, C,
         rows_per_sample, int(act), stream())
    return out


def bn_backward(G2d, X2d, bn: torch.nn.BatchNorm2d, st: BNState, dgamma, dbeta, training: bool, *, row_scale=None,
                gate=None, gate_add=None, act: Optional[BNState] = None, rows_per_sample=1, out=None):
    """Full BatchNorm backward of y = BN(X) given the (lazily composed) upstream gradient; returns dX.
    `act` is the BNState whose affine feeds a SiLU that sits between this BN's output and G (i.e. the
    same BN: G is d/d swish(BN(X))), so the SiLU derivative is recomputed from X."""
    rows, C = X2d.shape
    P = lib().mx_colreduce_parts(rows, C)
    sums = _f32(P, 2, C, device=X2d.device)
    a_sc = act.scale if act is not None else None
    a_sh = act.shift if act is not None else None
    call("mx_bn_bwd_reduce", ptr(G2d), ptr(X2d), ptr(row_scale), ptr(gate), ptr(gate_add), ptr(a_sc), ptr(a_sh), rows, C,
         rows_per_sample, ptr(sums), stream())
    c = _f32(3, C, device=X2d.device)
    call("mx_bn_bwd_finalize", ptr(sums), P, C, float(rows), ptr(bn.weight), ptr(st.mean), ptr(st.rstd), int(training),
         ptr(dgamma), ptr(dbeta), ptr(c[0]), ptr(c[1]), ptr(c[2]),
         _acc_scratch(X2d.device, 128 * C), stream())
    if out is None:
        out = torch.empty_like(X2d)
    call("mx_bn_bwd_apply", ptr(G2d), ptr(X2d), ptr(row_scale), ptr(gate), ptr(gate_add), ptr(a_sc), ptr(a_sh),
         ptr(c[0]), ptr(c[1]), ptr(c[2]), ptr(out), rows, C, rows_per_sample, stream())
    return out


def bn_backward_from_coeffs(G2d, X2d, st: BNState, c, *, gate, gate_add, rows_per_sample, out):
    """dX of BatchNorm-1 for g = (G*gate + gate_add)*swish'(bn(X)) with the coefficients c [3,C] already known (bn1_coeffs)."""
    rows, C = X2d.shape
    cb, cs = c.data_ptr(), 4 * C
    call("mx_bn_bwd_apply", ptr(G2d), ptr(X2d), None, ptr(gate), ptr(gate_add), ptr(st.scale), ptr(st.shift),
         cb, cb + cs, cb + 2 * cs, ptr(out), rows, C, rows_per_sample, stream())
    return out


def bn_bwd_coeffs(part, rows, bn, st: BNState, dgamma, dbeta, training: bool):
    """dgamma/dbeta (+=) and the [3,C] coefficients (c1,c2,c3) of dX = c1*g + c2*X + c3 from partial rows part[P][2][C]."""
    P, _, C = part.shape
    c = _f32(3, C, device=part.device)
    base = c.data_ptr()
    call("mx_bn_bwd_finalize", ptr(part), P, C, float(rows), ptr(bn.weight), ptr(st.mean), ptr(st.rstd), int(training),
         ptr(dgamma), ptr(dbeta), base, base + 4 * C, base + 8 * C, _acc_scratch(part.device, 128 * C), stream())
    return c


def bn_bwd_apply_plain(G2d, X2d, c, out):
    """dX = c1*G + c2*X + c3 (G already carries every upstream factor)."""
    rows, C = X2d.shape
    cb, cs = c.data_ptr(), 4 * c.shape[1]
    call("mx_bn_bwd_apply", ptr(G2d), ptr(X2d), None, None, None, None, None, cb, cb + cs, cb + 2 * cs, ptr(out),
         rows, C, 1, stream())
    return out


def dw_parts_reduce(scratch, dW):
    """dW += the partial rows a deferred dwconv_bwd_fused left in `scratch` [P, C*K*K]."""
    P, n = scratch.shape
    call("mx_dw_parts_reduce", ptr(scratch), P, n, ptr(dW), stream())


def dwconv_bwd_fused(dA, D, gate, add, st1: BNState, c1, X, st0: Optional[BNState], W, dW, K, pad_lo, *, residual=None, defer=None,
                     bn0=None):
    """Stride-1 fused backward of [BN0+SiLU] -> dwconv -> BN1 -> SiLU -> gate; returns (gX, BN0 partial sums or None).
    defer: a callable taking (scratch, dW) - the addition of the weight gradient's partial rows is handed to it (e.g. queued for
    the side stream) instead of being launched behind the kernel.
    bn0 = (bn module, dgamma, dbeta, training): the kernel finishes the BatchNorm-0 backward statistics itself (its last workgroup per
    channel chunk; mx_dwconv_bwd_fused_bn0) and a third value is returned: the [3, C] coefficients bn_bwd_coeffs would have made."""
    N, H, Wd, C = X.shape
    gX = _f32(N, H, Wd, C, device=X.device)
    P = lib().mx_dwconv_bwd_fused_parts(N, H, Wd, C, K)
    part = _f32(P, 2, C, device=X.device) if st0 is not None else None
    scratch = _f32(P, C * K * K, device=X.device)
    cb, cs = c1.data_ptr(), 4 * c1.shape[1]
    if bn0 is not None and st0 is not None and residual is None:
        bn, dgamma, dbeta, training = bn0
        co = _f32(3, C, device=X.device)
        ob = co.data_ptr()
        ws, wsn = _scratch(X.device, 65536)
        _call_ws("mx_dwconv_bwd_fused_bn0", ptr(dA), ptr(D), ptr(gate), ptr(add), ptr(st1.scale), ptr(st1.shift), cb, cb + cs, cb + 2 * cs,
                 ptr(X), ptr(st0.scale), ptr(st0.shift), ptr(W), ptr(gX), None if defer is not None else ptr(dW), ptr(scratch), ptr(part),
                 N, H, Wd, C, K, pad_lo, ws, wsn, float(N * H * Wd), ptr(bn.weight), ptr(st0.mean), ptr(st0.rstd), int(training),
                 ptr(dgamma), ptr(dbeta), ob, ob + 4 * C, ob + 8 * C, stream())
        if defer is not None:
            defer(scratch, dW)
        return gX, part, co
    call("mx_dwconv_bwd_fused", ptr(dA), ptr(D), ptr(gate), ptr(add), ptr(st1.scale), ptr(st1.shift), cb, cb + cs,
         cb + 2 * cs, ptr(X), ptr(st0.scale) if st0 else None, ptr(st0.shift) if st0 else None, ptr(W), ptr(residual), ptr(gX),
         None if defer is not None else ptr(dW), ptr(scratch), ptr(part), N, H, Wd, C, K, pad_lo, stream())
    if defer is not None:
        defer(scratch, dW)
    return (gX, part, None) if bn0 is not None else (gX, part)


def se_bn1_pool(dA2d, X2d, st: BNState, rows_per_sample):
    rows, C = X2d.shape
    N = rows // rows_per_sample
    out = _f32(5, N, C, device=X2d.device)
    ws, wsn = _scratch(X2d.device, lib().mx_pool_ws(rows, C, rows_per_sample, 5))
    _call_ws("mx_se_bn1_pool", ptr(dA2d), ptr(X2d), ptr(st.scale), ptr(st.shift), rows, C, rows_per_sample, ptr(out), ws, wsn, stream())
    return out


def bn1_coeffs(pooled5, gate, gh, W1, inv_hw, rows, bn, st: BNState, dgamma, dbeta, training: bool):
    """From the pooled sums and the SE backward's gh [N,SQ]: the pooled-path gradient add [N,C], the BatchNorm-1 backward sums,
    dgamma/dbeta (+=) and the [3,C] coefficients of the BN1 data gradient, in one launch.  Returns (c, add)."""
    _, N, C = pooled5.shape
    SQ = W1.shape[0]
    c = _f32(3, C, device=gate.device)
    add = torch.empty_like(gate)
    base = c.data_ptr()
    call("mx_bn1_sums_finalize", ptr(pooled5), ptr(gate), ptr(gh), ptr(W1), float(inv_hw), ptr(add), N, C, SQ, float(rows),
         ptr(bn.weight), ptr(st.mean), ptr(st.rstd), int(training), ptr(dgamma), ptr(dbeta), base, base + 4 * C, base + 8 * C, stream())
    return c, add


def pool_sum(X2d, rows_per_sample, *, G=None, st: Optional[BNState] = None, act=False):
    rows, C = X2d.shape
    out = _f32(rows // rows_per_sample, C, device=X2d.device)
    ws, wsn = _scratch(X2d.device, lib().mx_pool_ws(rows, C, rows_per_sample, 1))
    _call_ws("mx_pool_sum", ptr(X2d), ptr(G), ptr(st.scale) if st else None, ptr(st.shift) if st else None, int(act), rows, C,
             rows_per_sample, ptr(out), ws, wsn, stream())
    return out


# ---- depthwise ------------------------------------------------------------------------------------
def dwconv_fwd(X, W, K, S, pad_lo, Ho, Wo, *, st: Optional[BNState] = None, want_stats=False, pool=None):
    """pool=(scale, shift) (inference): also returns the SE squeeze sums [N, C] of swish(scale*Y + shift)."""
    N, H, Wd, C = X.shape
    Y = _f32(N, Ho, Wo, C, device=X.device)
    stats = _f32(lib().mx_dwconv_fwd_parts(N, Ho, Wo, C, S), 2, C, device=X.device) if want_stats else None
    pooled = _f32(N, C, device=X.device) if pool is not None else None
    ws, wsn = _scratch(X.device, lib().mx_dwconv_fwd_ws(N, Ho, Wo, C, S)) if pool is not None else (None, 0)
    _call_ws("mx_dwconv_fwd", ptr(X), ptr(st.scale) if st else None, ptr(st.shift) if st else None, ptr(W), ptr(Y), ptr(stats),
             ptr(pool[0]) if pool is not None else None, ptr(pool[1]) if pool is not None else None, ptr(pooled), ws, wsn,
             N, H, Wd, C, K, S, pad_lo, Ho, Wo, stream())
    if pool is not None:
        return Y, pooled
    return (Y, stats) if want_stats else Y


def dwconv_bwd_data(dY, W, K, S, pad_lo, H, Wd, *, residual=None):
    N, Ho, Wo, C = dY.shape
    dX = _f32(N, H, Wd, C, device=dY.device)
    call("mx_dwconv_bwd_data", ptr(dY), ptr(W), ptr(residual), ptr(dX), N, H, Wd, C, K, S, pad_lo, Ho, Wo, stream())
    return dX


def dwconv_bwd_weight(X, dY, dW, K, S, pad_lo, *, st: Optional[BNState] = None):
    N, H, Wd, C = X.shape
    _, Ho, Wo, _ = dY.shape
    scratch = _f32(lib().mx_dwconv_bwd_weight_parts(N, Ho, Wo, C, S), C * K * K, device=X.device)
    call("mx_dwconv_bwd_weight", ptr(X), ptr(st.scale) if st else None, ptr(st.shift) if st else None, ptr(dY), ptr(dW), ptr(scratch),
         N, H, Wd, C, K, S, pad_lo, Ho, Wo, stream())


# ---- SE / stem ------------------------------------------------------------------------------------
def se_fwd(pooled, inv_hw, W1, b1, W2, b2):
    N, C = pooled.shape
    SQ = W1.shape[0]
    s, gate = torch.empty_like(pooled), torch.empty_like(pooled)
    h = _f32(N, SQ, device=pooled.device)
    call("mx_se_fwd", ptr(pooled), float(inv_hw), ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(s), ptr(h), ptr(gate), N, C, SQ,
         stream())
    return s, h, gate


def se_bwd(ggate, gate, s, h, W2, dW1, db1, dW2, db2):
    """SE excitation backward: parameter gradients (+=) and gh [N,SQ] = dL/d(se_reduce output), from which bn1_coeffs forms
    the pooled-path gradient."""
    N, C = ggate.shape
    SQ = W2.shape[1]
    gh = torch.empty_like(h)
    call("mx_se_bwd", ptr(ggate), ptr(gate), ptr(s), ptr(h), ptr(W2), ptr(dW1), ptr(db1), ptr(dW2), ptr(db2), ptr(gh), N, C, SQ,
         stream())
    return gh


def se_bwd_gh(ggate, gate, h, W2):
    """First half of se_bwd: gh alone (the data-gradient chain needs nothing else of the excitation backward)."""
    N, C = ggate.shape
    gh = torch.empty_like(h)
    call("mx_se_bwd_gh", ptr(ggate), ptr(gate), ptr(h), ptr(W2), ptr(gh), N, C, W2.shape[1], stream())
    return gh


def se_bwd_params(ggate, gate, s, h, gh, dW1, db1, dW2, db2):
    """Second half of se_bwd: the parameter gradients (+=), which nothing reads before the optimizer."""
    N, C = ggate.shape
    call("mx_se_bwd_params", ptr(ggate), ptr(gate), ptr(s), ptr(h), ptr(gh), ptr(dW1), ptr(db1), ptr(dW2), ptr(db2), N, C,
         h.shape[1], stream())


def stem_im2col(img, Ho, Wo, pad_lo):
    N, _, H, W = img.shape
    out = _f32(N * Ho * Wo, 28, device=img.device)
    call("mx_stem_im2col", ptr(img), ptr(out), N, H, W, Ho, Wo, pad_lo, stream())
    return out


# ---- head -----------------------------------------------------------------------------------------
def resize_nhwc(src, dst, coff, relu=True):
    N, Hs, Ws, C = src.shape
    _, Hd, Wd, ldd = dst.shape
    call("mx_resize_nhwc", ptr(src), ptr(dst), N, Hs, Ws, C, Hd, Wd, ldd, coff, int(relu), stream())


def upsample_to_nchw(src, K, Hd, Wd):
    N, Hs, Ws, lds = src.shape
    dst = _f32(N, K, Hd, Wd, device=src.device)
    call("mx_upsample_to_nchw", ptr(src), ptr(dst), N, Hs, Ws, lds, K, Hd, Wd, stream())
    return dst


def upsample_to_nchw_bwd(gdst, gsrc, accumulate=False):
    N, K, Hd, Wd = gdst.shape
    _, Hs, Ws, lds = gsrc.shape
    call("mx_upsample_to_nchw_bwd", ptr(gdst), ptr(gsrc), N, Hs, Ws, lds, K, Hd, Wd, int(accumulate), stream())


def row_l2norm(x, eps):
    R, C = x.shape
    y, nrm = torch.empty_like(x), _f32(R, device=x.device)
    call("mx_row_l2norm", ptr(x), ptr(y), ptr(nrm), R, C, float(eps), stream())
    return y, nrm


def row_l2norm_bwd(x, nrm, gy, eps):
    R, C = x.shape
    gx = torch.empty_like(x)
    call("mx_row_l2norm_bwd", ptr(x), ptr(nrm), ptr(gy), ptr(gx), R, C, float(eps), stream())
    return gx


def pcm_norm(T, out, K, eps, grv=None):
    rows, L = T.shape[0] * T.shape[1], T.shape[2]
    call("mx_pcm_norm", ptr(T), ptr(grv), ptr(out), rows, L, K, float(eps), int(grv is not None), stream())
    return out


def sym_relu_grad(gaff, aff, n):
    out = torch.empty_like(gaff)
    call("mx_sym_relu_grad", ptr(gaff), ptr(aff), ptr(out), gaff.shape[0], n, gaff.shape[2], stream())
    return out


def ew(op, a, b=None, y=None, alpha=1.0, out=None):
    if out is None:
        out = torch.empty_like(a)
    call("mx_ew", op, ptr(a), ptr(b), ptr(y), float(alpha), ptr(out), a.numel(), stream())
    return out


def bcast_add(X2d, v, alpha, rows_per_sample):
    rows, C = X2d.shape
    call("mx_bcast_add", ptr(X2d), ptr(v), float(alpha), rows, C, rows_per_sample, stream())
