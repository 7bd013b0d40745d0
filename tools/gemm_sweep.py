import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
N = 32
names = ["128x128", "128x96", "128x32", "256x64", "128x64w", "128x128k32", "128x64", "128x64k32", "128x32k32", "128x96k32"]
shapes = [(N*224*224, 32, 32), (N*224*224, 32, 192), (N*112*112, 192, 48), (N*112*112, 48, 288), (N*112*112, 288, 48), (N*56*56, 80, 480), (N*56*56, 480, 80),
          (N*28*28, 160, 960), (N*28*28, 960, 160), (N*28*28, 224, 1344), (N*28*28, 1344, 224), (N*28*28, 384, 2304), (N*28*28, 2304, 384), (N*28*28, 640, 3840), (N*28*28, 3840, 640)]
for (M, K, Nn) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(Nn, K, device=dev) * 0.05; Gm = torch.randn(M, Nn, device=dev); dW = torch.zeros(Nn, K, device=dev)
    fl = 2.0 * M * K * Nn
    rows = []
    for cfg in range(10):
        os.environ["MX_GEMM_CFG"] = str(cfg)
        t = timeit(lambda: ops.pw_fwd(A, W, Nn, want_stats=True)); t2 = timeit(lambda: ops.pw_dgrad(Gm, W, K)); t3 = timeit(lambda: ops.pw_wgrad(Gm, A, dW))
        rows.append((fl/t/1e12, fl/t2/1e12, fl/t3/1e12))
    del os.environ["MX_GEMM_CFG"]
    t = timeit(lambda: ops.pw_fwd(A, W, Nn, want_stats=True)); t2 = timeit(lambda: ops.pw_dgrad(Gm, W, K)); t3 = timeit(lambda: ops.pw_wgrad(Gm, A, dW))
    print(f"M={M} K={K} N={Nn}  auto: fwd {fl/t/1e12:.1f} dgrad {fl/t2/1e12:.1f} wgrad {fl/t3/1e12:.1f}")
    for nm, r in zip(names, rows):
        print(f"    {nm:8s} fwd {r[0]:6.1f}  dgrad(N={K}) {r[1]:6.1f}  wgrad(out {Nn}x{K}) {r[2]:6.1f}")
