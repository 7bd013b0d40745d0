"""Narrow-layer forward GEMM (K <= 80, store-bound): rate by row count and epilogue (training statistics vs inference bias)."""
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


for K, N in ((48, 288), (80, 480), (32, 192)):
    for M in (200704, 401408, 802816, 1605632):
        if M * N * 4 > 3e9:
            continue
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        t_stats = timeit(lambda: ops.pw_fwd(A, W, N, want_stats=True, out=out))
        t_bias = timeit(lambda: ops.pw_fwd(A, W, N, bias=bias, out=out))
        t_plain = timeit(lambda: ops.pw_fwd(A, W, N, out=out))
        gb = M * (K + N) * 4 / 1e9
        print(f"M={M:8d} K={K:3d} N={N:4d}: stats {t_stats:7.1f} us {gb/t_stats*1e3:6.2f} TB/s | bias {t_bias:7.1f} us {gb/t_bias*1e3:6.2f} TB/s | plain {t_plain:7.1f} us {gb/t_plain*1e3:6.2f} TB/s")
