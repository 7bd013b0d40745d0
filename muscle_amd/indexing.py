"""IRN random-walk propagation on the HIP path — src/indexing.py::propagate_to_edge as called by infer_irn.py:76.

The reference builds flat index tables for every search path (PathIndex), gathers the padded edge map with
index_select, max-pools along each path, assembles a sparse COO matrix on the CPU, densifies it, moves it to the GPU,
raises it to beta, normalises the columns and squares it exp_times times with torch.matmul.  Here one kernel writes the
dense symmetric affinity matrix directly from the edge map, one pass turns it into the column-stochastic matrix, and the
2^exp_times-step walk is exp_times fp32 MFMA GEMMs (`mx_bgemm`) ping-ponging between two buffers; the class maps are
propagated with one more GEMM.  For a VOC image at the IRN's 1/4 resolution (≈94x125 = 11.7k vertices) that is
8 x 2 x 11.7k^3 = 25.7 TFLOP per image, by far the dominant cost of infer_irn.py.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from . import ops
from ._lib import MuscleHipError, call, ptr, stream

_tables = {}


def search_paths(radius: int = 5) -> List[List[Tuple[int, int]]]:
    """PathIndex.get_search_paths_dst (indexing.py:13-47): the straight path (farthest pixel first) to every search
    destination, in the reference's order (grouped by path length, which fixes nothing observable in the dense matrix
    but is kept so the table can be compared with the reference's)."""
    dirs = [(0, x) for x in range(1, radius)]
    for y in range(1, radius):
        for x in range(-radius + 1, radius):
            if x * x + y * y < radius ** 2:
                dirs.append((y, x))
    by_len = {}
    for dy, dx in dirs:
        lsq = dy * dy + dx * dx
        coords = [(y, x) for y in range(min(0, dy), max(0, dy) + 1) for x in range(min(0, dx), max(0, dx) + 1)
                  if (dy * x - dx * y) ** 2 / lsq < 1]
        coords.sort(key=lambda c: -abs(c[0]) - abs(c[1]))
        by_len.setdefault(len(coords), []).append(coords)
    return [p for k in sorted(by_len) for p in by_len[k]]


def _path_table(radius: int, device):
    key = (radius, str(device))
    if key not in _tables:
        paths = search_paths(radius)
        pc = np.array([c for p in paths for c in p], np.int32).reshape(-1)
        ln = np.array([len(p) for p in paths], np.int32)
        off = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.int32)
        _tables[key] = tuple(torch.from_numpy(a).to(device) for a in (pc, off, ln)) + (len(paths),)
    return _tables[key]


def propagate_to_edge(x: torch.Tensor, edge: torch.Tensor, radius: int = 5, beta: float = 10, exp_times: int = 8) -> torch.Tensor:
    """x: [..., C, h, w] class maps (any leading 1s), edge: [1,h,w] or [h,w] boundary probability, both CUDA.
    Returns rw [C,1,h,w] like the reference."""
    if not x.is_cuda or not edge.is_cuda:
        raise MuscleHipError("propagate_to_edge runs on the HIP kernels only")
    h, w = x.shape[-2:]
    n = h * w
    n4 = (n + 3) // 4 * 4                         # GEMM operands need leading dimensions that are multiples of 4
    dev = x.device
    e = edge.reshape(h, w).contiguous().float()
    pc, off, ln, nd = _path_table(radius, dev)
    A = torch.empty(n4, n4, dtype=torch.float32, device=dev)
    B = torch.empty(n4, n4, dtype=torch.float32, device=dev)
    call("mx_irn_affinity", ptr(e), h, w, radius, ptr(pc), ptr(off), ptr(ln), nd, ptr(A), n4, n4, stream())
    colsum = torch.empty(n4, dtype=torch.float32, device=dev)
    call("mx_irn_transition", ptr(A), n4, n4, float(beta), ptr(colsum), stream())
    for _ in range(exp_times):                     # trans <- trans @ trans   (indexing.py:119-120)
        ops.bgemm(1, A.view(1, n4, n4), A.view(1, n4, n4), B.view(1, n4, n4), n4, n4, n4)
        A, B = B, A
    xm = (x.reshape(-1, h, w).float() * (1 - e)).reshape(-1, n)
    C = xm.shape[0]
    C4 = (C + 3) // 4 * 4
    xp = torch.zeros(C4, n4, dtype=torch.float32, device=dev)
    xp[:C, :n] = xm
    rw = torch.empty(C4, n4, dtype=torch.float32, device=dev)
    ops.bgemm(1, xp.view(1, C4, n4), A.view(1, n4, n4), rw.view(1, C4, n4), C4, n4, n4)
    return rw[:C, :n].reshape(C, 1, h, w)


def finish_semseg(rw: torch.Tensor, H: int, W: int, bg_thres: float, soft_output: bool = False):
    """infer_irn.py:78-94: rw [C,1,h,w] -> uint8 label map [H,W] (argmax over [bg_thres, upsampled rw / max]); with
    soft_output also the fp16 [H,W,C+1] array the script saves."""
    C, _, h, w = rw.shape
    r = rw.reshape(C, h, w).contiguous().float()
    label = torch.empty(H, W, dtype=torch.uint8, device=rw.device)
    soft = torch.empty(H, W, C + 1, dtype=torch.float16, device=rw.device) if soft_output else None
    scratch = torch.empty(1, dtype=torch.int32, device=rw.device)
    call("mx_irn_finish", ptr(r), C, h, w, H, W, float(bg_thres), ptr(scratch), ptr(label), ptr(soft), stream())
    return (label, soft) if soft_output else label
