"""One late-stage block's forward chain in isolation (B7 block 45 in MuSCLe, whose stage 6 keeps stride 1: 28x28, 384 -> 2304 -> 384, k5), per-kernel events.
Per-kernel times to set beside the in-step times of a rocprofv3 kernel trace (HW, CIN, CEXP from the environment)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import muscle_amd
from muscle_amd import ops
from muscle_amd.ops import BNState
dev = torch.device("cuda:0")
N, H, Cin, Cexp, K = 32, int(os.environ.get("HW", "28")), int(os.environ.get("CIN", "384")), int(os.environ.get("CEXP", "2304")), 5
M = N * H * H
x = torch.randn(M, Cin, device=dev)
We = torch.randn(Cexp, Cin, device=dev) * 0.05
Wp = torch.randn(Cin, Cexp, device=dev) * 0.02
Wd = torch.randn(Cexp, 1, K, K, device=dev) * 0.2
st0 = BNState(torch.ones(Cexp, device=dev), torch.zeros(Cexp, device=dev), None, None)
st1 = BNState(torch.ones(Cexp, device=dev), torch.zeros(Cexp, device=dev), None, None)
gate = torch.rand(N, Cexp, device=dev)
names = ["expand gemm", "dw_fwd", "pool", "bn_apply(act,gate)", "project gemm"]


def chain(ev):
    def rec():
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(e)
    rec()
    e_raw, _ = ops.pw_fwd(x, We, Cexp, want_stats=True); rec()
    d, _ = ops.dwconv_fwd(e_raw.view(N, H, H, Cexp), Wd, K, 1, 2, H, H, st=st0, want_stats=True); rec()
    p = ops.pool_sum(d.view(M, Cexp), H * H, st=st1, act=True); rec()
    a = ops.bn_apply(d.view(M, Cexp), st1, gate=gate, rows_per_sample=H * H, act=True); rec()
    o, _ = ops.pw_fwd(a, Wp, Cin, want_stats=True); rec()


for reps in (1, 30):
    allev = []
    for _ in range(3):
        chain([])
    torch.cuda.synchronize()
    for _ in range(reps):
        ev = []; chain(ev); allev.append(ev)
    torch.cuda.synchronize()
    print(f"--- {reps} chain(s) back to back, N={N} HW={H} Cin={Cin} Cexp={Cexp}")
    for i, nm in enumerate(names):
        ts = [e[i].elapsed_time(e[i + 1]) * 1e3 for e in allev]
        print(f"  {nm:20s} median {statistics.median(ts):7.1f} us  min {min(ts):7.1f}  max {max(ts):7.1f}")
