"""K = 48 / 80 forward GEMMs: the planes kernel with its half K step against the exact-fp32 kernel (us per launch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import muscle_amd
from muscle_amd import ops
dev = "cuda"
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, K, N in [(401408, 48, 288), (401408, 48, 192), (100352, 80, 480), (100352, 80, 288), (401408, 288, 48), (100352, 480, 80)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
    plan = ops.PlanesPlan([W]); (img,) = plan.run()
    muscle_amd.set_gemm_mode(1)
    row = []
    for st in (True, False):
        t0 = timeit(lambda: ops.pw_fwd(A, W, N, want_stats=st))
        t1 = timeit(lambda: ops.pw_fwd(A, W, N, want_stats=st, planes=img)) if img is not None else float("nan")
        row.append((t0, t1))
    by = 4.0 * (M * K + M * N)
    print(f"M={M} K={K} N={N}: stats fp32 {row[0][0]:.1f} planes {row[0][1]:.1f} | plain fp32 {row[1][0]:.1f} planes {row[1][1]:.1f} us   (5 TB/s: {by/5e6:.1f} us)", flush=True)
print("project forward with the BN1 + SiLU + gate prologue: exact-fp32 kernel vs the planes kernel")
for M, K, N, rps in [(401408, 288, 48, 12544), (401408, 192, 48, 12544), (100352, 480, 80, 3136), (100352, 288, 80, 3136)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
    sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1; gate = torch.rand(M // rps, K, device=dev)
    plan = ops.PlanesPlan([W]); (img,) = plan.run()
    muscle_amd.set_gemm_mode(1)
    kw = dict(a_mode=ops.BNACT, a_scale=sc, a_shift=sh, a_gate=gate, rows_per_sample=rps, want_stats=True)
    t0 = timeit(lambda: ops.pw_fwd(A, W, N, **kw))
    t1 = timeit(lambda: ops.pw_fwd(A, W, N, planes=img, **kw))
    print(f"M={M} K={K} N={N}: fp32 {t0:.1f} planes {t1:.1f} us   (5 TB/s: {4.0*(M*K+M*N)/5e6:.1f} us)", flush=True)
