"""Per-kernel times of the small-output weight gradient (run under rocprofv3 --kernel-trace --stats) on the first-stage shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_amd import ops
dev = torch.device("cuda:0")
for (R, Co, Ci) in [(401408, 288, 48), (401408, 48, 288), (1605632, 32, 32), (1605632, 192, 32), (100352, 480, 80)]:
    G = torch.randn(R, Co, device=dev); X = torch.randn(R, Ci, device=dev); dW = torch.zeros(Co, Ci, device=dev)
    for _ in range(5):
        ops.pw_wgrad(G, X, dW)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.pw_wgrad(G, X, dW)
    e1.record(); torch.cuda.synchronize()
    print(f"R={R} Co={Co} Ci={Ci}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call", flush=True)
    del G, X
