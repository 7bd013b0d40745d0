"""Phase durations inside dw_bwd_fused_kernel from the -DMX_DW_STAMPS diagnostic build (tools/hip/build/lib_stamps.so swapped in as the
library): per tile of thread 0 of the first 64 workgroups - staging (issue + transform), barrier wait, compute."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from muscle_amd import ops
from muscle_amd._lib import lib
from muscle_amd.ops import BNState
dev = "cuda"
L = lib()
for C, H, K in [(2304, 28, 5), (480, 56, 5), (3840, 28, 3), (288, 112, 3)]:
    N = 32
    rn = lambda *s: torch.randn(*s, device=dev)
    mk = lambda: BNState(torch.rand(C, device=dev) + 0.5, rn(C) * 0.1, rn(C) * 0.1, torch.rand(C, device=dev) + 0.5)
    dA, D, X = rn(N, H, H, C), rn(N, H, H, C), rn(N, H, H, C)
    gate, add, st1, st0, c1 = torch.sigmoid(rn(N, C)), rn(N, C) * 0.1, mk(), mk(), rn(3, C) * 0.1
    W = rn(C, 1, K, K); dW = torch.zeros_like(W)
    for _ in range(5):
        ops.dwconv_bwd_fused(dA, D, gate, add, st1, c1, X, st0, W, dW, K, (K - 1) // 2)
    torch.cuda.synchronize()
    buf = np.zeros(64 * 32 * 4, dtype=np.uint64)
    rc = L.mx_dw_stamps_read(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    s = buf.reshape(64, 32, 4).astype(np.int64)
    ok = (s[:, :, 3] > s[:, :, 0]) & (s[:, :, 0] > 0)
    tiles = ok.sum(1)
    st = (s[:, :, 1] - s[:, :, 0])[ok] * 0.01; bar = (s[:, :, 2] - s[:, :, 1])[ok] * 0.01; cp = (s[:, :, 3] - s[:, :, 2])[ok] * 0.01
    nxt = (s[:, 1:, 0] - s[:, :-1, 3])[ok[:, 1:] & ok[:, :-1]] * 0.01
    print(f"k{K} C={C} H={H}: tiles/wg {tiles.min()}-{tiles.max()}  staging {np.median(st):.2f} us (p90 {np.percentile(st, 90):.2f})  "
          f"barrier {np.median(bar):.2f} (p90 {np.percentile(bar, 90):.2f})  compute {np.median(cp):.2f} (p90 {np.percentile(cp, 90):.2f})  "
          f"tile-to-tile {np.median(nxt) if len(nxt) else float('nan'):.2f}", flush=True)
