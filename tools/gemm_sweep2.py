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
cfgs = {0: "128x128", 1: "128x96", 2: "128x32", 10: "256x128", 11: "128x256"}
shapes = [(N*28*28, 384, 2304), (N*28*28, 2304, 384), (N*28*28, 640, 3840), (N*28*28, 3840, 640), (N*28*28, 224, 1344), (N*28*28, 1344, 224)]
for (M, K, Nn) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(Nn, K, device=dev) * 0.05; Gm = torch.randn(M, Nn, device=dev); dW = torch.zeros(Nn, K, device=dev)
    sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1; gate = torch.rand(N, K, device=dev)
    fl = 2.0 * M * K * Nn
    print(f"M={M} K={K} N={Nn}")
    for cfg, nm in cfgs.items():
        os.environ["MX_GEMM_CFG"] = str(cfg)
        t = timeit(lambda: ops.pw_fwd(A, W, Nn, want_stats=True)); t2 = timeit(lambda: ops.pw_dgrad(Gm, W, K)); t3 = timeit(lambda: ops.pw_wgrad(Gm, A, dW))
        t4 = timeit(lambda: ops.pw_fwd(A, W, Nn, a_mode=1, a_scale=sc, a_shift=sh, a_gate=gate, rows_per_sample=784, want_stats=True))
        print(f"    {nm:8s} fwd {fl/t/1e12:6.1f}  fwd+BNACT {fl/t4/1e12:6.1f}  dgrad {fl/t2/1e12:6.1f}  wgrad {fl/t3/1e12:6.1f}")
    del os.environ["MX_GEMM_CFG"]
