"""SE excitation kernels per B7 stage width: forward (reduce + expand) and backward (gh, parameter gradients, BN1 sums), us per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from muscle_amd import ops
dev = "cuda"
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
N = 32
for C, SQ, hw in [(288, 12, 12544), (480, 20, 3136), (960, 40, 784), (1344, 56, 784), (2304, 96, 784), (3840, 160, 784)]:
    rn = lambda *s: torch.randn(*s, device=dev)
    pooled = rn(N, C) * hw; W1, b1, W2, b2 = rn(SQ, C) * 0.05, rn(SQ), rn(C, SQ) * 0.05, rn(C)
    tf = timeit(lambda: ops.se_fwd(pooled, 1.0 / hw, W1, b1, W2, b2))
    s, h, gate = ops.se_fwd(pooled, 1.0 / hw, W1, b1, W2, b2)
    gg = rn(N, C)
    dW1, db1, dW2, db2 = torch.zeros_like(W1), torch.zeros_like(b1), torch.zeros_like(W2), torch.zeros_like(b2)
    tb = timeit(lambda: ops.se_bwd(gg, gate, s, h, W2, dW1, db1, dW2, db2))
    tg = timeit(lambda: ops.se_bwd_gh(gg, gate, h, W2))
    print(f"C={C} SQ={SQ}: se_fwd {tf:.1f} us | se_bwd (gh + params) {tb:.1f} us, gh alone {tg:.1f}", flush=True)
