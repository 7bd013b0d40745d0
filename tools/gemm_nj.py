"""Split NT GEMM, 128 x 128 vs 128 x 64 tiles (MX_GEMM_SPLIT_NJ=2 / 1) on the project / expand-dgrad shapes of stages 5-7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M, K, N in ((25088, 2304, 384), (25088, 3840, 640), (25088, 1344, 224), (25088, 384, 2304), (25088, 640, 3840), (25088, 224, 1344), (25088, 960, 160), (25088, 160, 960)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
    out = torch.empty(M, N, device=dev)
    t = timeit(lambda: ops.pw_fwd(A, W, N, want_stats=True, out=out))
    print(f"NJ={os.environ.get('MX_GEMM_SPLIT_NJ', 'auto')} M={M} K={K:5d} N={N:5d}: {t:7.1f} us  {2*M*K*N/t/1e6:6.1f} TF")
