"""Per-shape timing of the three pointwise GEMM kinds on the B7 layer shapes (`child` mode).  The parent mode re-runs the
child under MX_GEMM_DEBUG values; those bits only exist in throw-away builds used for the bottleneck isolation recorded
in DESIGN.md section 3 (no global loads / no LDS stores / no barrier / no epilogue) and are not in the committed kernel."""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
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
    for (M, K, N) in [(25088, 640, 3840), (25088, 3840, 640), (25088, 384, 2304), (25088, 2304, 384), (25088, 224, 1344), (25088, 1344, 224), (25088, 160, 960), (25088, 960, 160), (100352, 80, 480), (100352, 480, 80), (401408, 48, 288), (401408, 288, 48), (401408, 192, 48), (1605632, 32, 192)]:
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
        fl = 2.0 * M * K * N
        t = timeit(lambda: ops.pw_fwd(A, W, N, want_stats=True))
        G = torch.randn(M, N, device=dev)
        t2 = timeit(lambda: ops.pw_dgrad(G, W, K))
        dW = torch.zeros(N, K, device=dev)
        t3 = timeit(lambda: ops.pw_wgrad(G, A, dW))
        print(f"  M={M} K={K} N={N}: fwd {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF | dgrad {t2*1e6:7.1f} us {fl/t2/1e12:6.1f} TF | wgrad {t3*1e6:7.1f} us {fl/t3/1e12:6.1f} TF", flush=True)
else:
    for dbg in (0, 16, 8):
        print(f"MX_GEMM_DEBUG={dbg}", flush=True)
        env = dict(os.environ, MX_GEMM_DEBUG=str(dbg))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
