"""Is the SE squeeze slower on a tensor that the previous kernel has just written?  (round 5: 78 us in the step against 51 us standalone at C = 2304)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from muscle_amd import ops
from muscle_amd.ops import BNState
dev = torch.device("cuda:0")
N, H = 32, 28
for C in (960, 1344, 2304, 3840):
    rows = N * H * H
    X = torch.randn(rows, C, device=dev); Y = torch.randn(rows, C, device=dev); Z = torch.empty_like(X)
    st = BNState(torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    def run(writer):
        tw = tp = 0.0
        for _ in range(10):
            ev[0].record(); writer(); ev[1].record(); ops.pool_sum(X, H * H, st=st, act=True); ev[2].record()
            torch.cuda.synchronize()
            tw += ev[0].elapsed_time(ev[1]); tp += ev[1].elapsed_time(ev[2])
        return tw * 100, tp * 100
    for _ in range(3): ops.pool_sum(X, H * H, st=st, act=True)
    a = run(lambda: None)
    b = run(lambda: X.copy_(Y))                                         # X written by the previous kernel
    c = run(lambda: Z.copy_(Y))                                         # another tensor written (X pushed out of the caches)
    d = run(lambda: ops.bn_apply(Y, st, act=True, out=X) if False else X.mul_(1.0))   # X read and written in place
    print(f"C={C:5d}: pool alone {a[1]:6.1f} us | after X.copy_(Y) [{b[0]:6.1f}] {b[1]:6.1f} | after Z.copy_(Y) [{c[0]:6.1f}] {c[1]:6.1f} | after X.mul_(1) [{d[0]:6.1f}] {d[1]:6.1f}", flush=True)
