import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, muscle_amd
from muscle_amd import ops
DEV = "cuda:0"
for (M, K, N) in [(6272, 2304, 384), (25088, 960, 160), (3001, 384, 130)]:
    g = torch.Generator(device=DEV).manual_seed(M + K)
    G = torch.randn(M, K, device=DEV, generator=g); X = torch.randn(M, K, device=DEV, generator=g)
    coef = torch.stack([torch.rand(K, device=DEV, generator=g) + 0.5, torch.randn(K, device=DEV, generator=g) * 0.3,
                        torch.randn(K, device=DEV, generator=g) * 0.1]).contiguous()
    Wt = torch.randn(N, K, device=DEV, generator=g) * (K ** -0.5)
    Xin = torch.randn(M, N, device=DEV, generator=g)
    (image,) = ops.PlanesPlan([Wt]).run()
    muscle_amd.set_gemm_mode(1)
    dz = ops.bn_bwd_apply_plain(G, X, coef, torch.empty_like(G))
    for name, fn in (("dgrad fold", lambda: ops.pw_dgrad_bnbwd_planes(G, X, coef, image, N)),
                     ("dgrad plain planes", lambda: ops.pw_dgrad(dz, None, N, wt=Wt, planes=image)),
                     ("wgrad fold", lambda: (lambda dw: (ops.pw_wgrad_bnbwd(G, X, coef, Xin, dw), dw)[1])(torch.zeros(K, N, device=DEV))),
                     ("wgrad plain", lambda: (lambda dw: (ops.pw_wgrad(dz, Xin, dw), dw)[1])(torch.zeros(K, N, device=DEV)))):
        outs = [fn().clone() for _ in range(4)]
        torch.cuda.synchronize()
        d = [float((o - outs[0]).abs().max()) for o in outs[1:]]
        nz = [int((o != outs[0]).sum()) for o in outs[1:]]
        print(M, K, N, name, "maxdiff", d, "count", nz, flush=True)
        if nz[0]:
            idx = (outs[1] != outs[0]).nonzero()
            print("   first diffs at", idx[:6].tolist(), "rows range", int(idx[:, 0].min()), int(idx[:, 0].max()), "cols", int(idx[:, 1].min()), int(idx[:, 1].max()))
