"""Weight-gradient kernels on the B7 / 448 / batch-32 layer shapes: the deterministic kernels of wgrad.hip (small-output and
tiled) against the atomic TN GEMM of gemm.hip they replace (MUSCLE_WGRAD_TILE / ops.WGRAD_SMALL toggled per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_amd import ops
dev = torch.device("cuda:0")


def t(R, Co, Ci):
    G = torch.randn(R, Co, device=dev); X = torch.randn(R, Ci, device=dev); dW = torch.zeros(Co, Ci, device=dev)
    for _ in range(3):
        ops.pw_wgrad(G, X, dW)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.pw_wgrad(G, X, dW)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


shapes = [(25088, 3840, 640), (25088, 640, 3840), (25088, 2304, 384), (25088, 384, 2304), (25088, 1344, 224), (25088, 224, 1344),
          (25088, 960, 160), (25088, 160, 960), (100352, 480, 80), (100352, 80, 480), (401408, 288, 48), (401408, 48, 288)]
for (R, Co, Ci) in shapes:
    new = t(R, Co, Ci)
    ops.WGRAD_TILE, ops.WGRAD_SMALL = False, False
    old = t(R, Co, Ci)
    ops.WGRAD_TILE, ops.WGRAD_SMALL = True, True
    fl = 2.0 * R * Co * Ci
    print(f"R={R} Co={Co} Ci={Ci}: wgrad.hip {new:7.1f} us {fl / new / 1e6:6.1f} TF | atomic TN GEMM {old:7.1f} us {fl / old / 1e6:6.1f} TF", flush=True)
