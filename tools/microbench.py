"""Per-kernel micro-benchmarks on the B7/448/bs32 shapes (GB/s or TFLOP/s per call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_amd import ops
from muscle_amd.ops import BNState
dev = torch.device("cuda:0")
which = sys.argv[1:] or ["reduce", "apply", "pool", "dw", "gemm"]

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

def bnstate(C):
    return BNState(torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5)

N = 32
shapes = [(32, 224), (288, 112), (480, 56), (960, 28), (1344, 28), (2304, 28), (3840, 28), (192, 224)]
for C, H in shapes:
    rows = N * H * H
    if rows * C * 4 > 6e9: continue
    G = torch.randn(rows, C, device=dev); X = torch.randn(rows, C, device=dev)
    st = bnstate(C)
    bn = torch.nn.BatchNorm2d(C).to(dev)
    gate = torch.rand(N, C, device=dev); add = torch.randn(N, C, device=dev) * 0.01
    gb = rows * C * 4 / 1e9
    if "reduce" in which:
        from muscle_amd._lib import call, ptr, stream, lib
        P = lib().mx_colreduce_parts(rows, C)
        part = torch.empty(P, 2, C, device=dev)
        t = timeit(lambda: call("mx_bn_bwd_reduce", ptr(G), ptr(X), None, ptr(gate), ptr(add), ptr(st.scale), ptr(st.shift), rows, C, H * H, ptr(part), stream()))
        t2 = timeit(lambda: call("mx_bn_bwd_reduce", ptr(G), ptr(X), None, None, None, None, None, rows, C, H * H, ptr(part), stream()))
        t3 = timeit(lambda: call("mx_colstats", ptr(X), rows, C, ptr(part), stream()))
        print(f"reduce C={C:5d} H={H:4d} P={P}: full {t*1e6:7.1f}us {2*gb/t:7.1f} GB/s | plain {t2*1e6:7.1f}us {2*gb/t2:7.1f} GB/s | colstats {t3*1e6:7.1f}us {gb/t3:7.1f} GB/s")
    if "apply" in which:
        c = torch.randn(3, C, device=dev)
        out = torch.empty_like(G)
        t = timeit(lambda: call("mx_bn_bwd_apply", ptr(G), ptr(X), None, ptr(gate), ptr(add), ptr(st.scale), ptr(st.shift), ptr(c[0]), ptr(c[1]), ptr(c[2]), ptr(out), rows, C, H * H, stream()))
        print(f"apply  C={C:5d} H={H:4d}: {t*1e6:7.1f}us {3*gb/t:7.1f} GB/s")
    if "pool" in which:
        t = timeit(lambda: ops.pool_sum(X, H * H, st=st, act=True))
        t2 = timeit(lambda: ops.pool_sum(X, H * H, G=G, st=st, act=True))
        t3 = timeit(lambda: ops.se_bn1_pool(G, X, st, H * H))
        print(f"pool   C={C:5d} H={H:4d}: fwd {t*1e6:7.1f}us {gb/t:7.1f} GB/s | bwd {t2*1e6:7.1f}us {2*gb/t2:7.1f} GB/s | se_bn1_pool {t3*1e6:7.1f}us {2*gb/t3:7.1f} GB/s")
    if "dw" in which:
        for K in (3, 5):
            W = torch.randn(C, 1, K, K, device=dev)
            X4 = X.view(N, H, H, C)
            pad = (K - 1) // 2
            t = timeit(lambda: ops.dwconv_fwd(X4, W, K, 1, pad, H, H, st=st, want_stats=True))
            dY = G.view(N, H, H, C)
            t2 = timeit(lambda: ops.dwconv_bwd_data(dY, W, K, 1, pad, H, H))
            dW = torch.zeros_like(W)
            t3 = timeit(lambda: ops.dwconv_bwd_weight(X4, dY, dW, K, 1, pad, st=st))
            print(f"dw k{K} C={C:5d} H={H:4d}: fwd {t*1e6:7.1f}us {2*gb/t:7.1f} GB/s | bwd_data {t2*1e6:7.1f}us {2*gb/t2:7.1f} GB/s | bwd_w {t3*1e6:7.1f}us {2*gb/t3:7.1f} GB/s")
    if "dwfused" in which:
        from muscle_amd._lib import call, ptr, stream, lib
        for K in ((3,) if (C, H) in [(32, 224), (192, 224), (288, 112), (3840, 28)] else (5,) if (C, H) != (960, 28) else (3, 5)):
            W = torch.randn(C, 1, K, K, device=dev); dW = torch.zeros_like(W)
            X4 = X.view(N, H, H, C); dA = G.view(N, H, H, C); D = torch.randn(N, H, H, C, device=dev)
            st0 = bnstate(C); pad = (K - 1) // 2
            c1 = torch.randn(3, C, device=dev) * 0.1
            tf = timeit(lambda: ops.dwconv_bwd_fused(dA, D, gate, add, st, c1, X4, st0, W, dW, K, pad))
            dd = torch.empty_like(G)
            def unfused():
                call("mx_bn_bwd_apply", ptr(G), ptr(D), None, ptr(gate), ptr(add), ptr(st.scale), ptr(st.shift), ptr(c1[0]), ptr(c1[1]), ptr(c1[2]), ptr(dd), rows, C, H * H, stream())
                d4 = dd.view(N, H, H, C)
                ops.dwconv_bwd_weight(X4, d4, dW, K, 1, pad, st=st0)
                ge = ops.dwconv_bwd_data(d4, W, K, 1, pad, H, H)
                P = lib().mx_colreduce_parts(rows, C)
                part = torch.empty(P, 2, C, device=dev)
                call("mx_bn_bwd_reduce", ptr(ge), ptr(X), None, None, None, ptr(st0.scale), ptr(st0.shift), rows, C, 1, ptr(part), stream())
            tu = timeit(unfused)
            print(f"dwfused k{K} C={C:5d} H={H:4d}: fused {tf*1e6:7.1f}us ({4*gb/tf:6.0f} GB/s of 4 passes) | unfused {tu*1e6:7.1f}us", flush=True)
    del G, X
if "gemm" in which:
    # (M, K, N) of expand / project fwd for representative blocks
    for (M, K, Nn) in [(N*224*224, 32, 32), (N*224*224, 32, 192), (N*112*112, 192, 48), (N*112*112, 48, 288), (N*112*112, 288, 48), (N*56*56, 80, 480), (N*56*56, 480, 80),
                       (N*28*28, 160, 960), (N*28*28, 960, 160), (N*28*28, 224, 1344), (N*28*28, 1344, 224), (N*28*28, 384, 2304), (N*28*28, 2304, 384), (N*28*28, 640, 3840), (N*28*28, 3840, 640)]:
        A = torch.randn(M, K, device=dev); W = torch.randn(Nn, K, device=dev) * 0.05
        fl = 2.0 * M * K * Nn
        t = timeit(lambda: ops.pw_fwd(A, W, Nn, want_stats=True), 5)
        Gm = torch.randn(M, Nn, device=dev)
        t2 = timeit(lambda: ops.pw_dgrad(Gm, W, K), 5)
        dW = torch.zeros(Nn, K, device=dev)
        t3 = timeit(lambda: ops.pw_wgrad(Gm, A, dW), 5)
        print(f"gemm M={M:8d} K={K:5d} N={Nn:5d}: fwd {t*1e6:8.1f}us {fl/t/1e12:6.1f} TF | dgrad {t2*1e6:8.1f}us {fl/t2/1e12:6.1f} TF | wgrad {t3*1e6:8.1f}us {fl/t3/1e12:6.1f} TF  (min-bytes {(M*K+M*Nn)*4/1e9:.2f} GB -> {(M*K+M*Nn)*4/t/1e9:.0f} GB/s fwd)")
        del A, W, Gm
