"""What the GPU box's host share looks like to a test process, and what the CPU oracle costs under it."""
import os, sys, time
import numpy as np
t0 = time.time()
import torch
print("import torch", round(time.time() - t0, 1), flush=True)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(),
      "interop", torch.get_num_interop_threads(), flush=True)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.strerror)
print({k: v for k, v in os.environ.items() if "THREADS" in k or "OMP" in k or "MKL" in k})
import golden_util as gu
from muscle_amd import synth
from muscle_amd.arch import net_cfg
from oracle import mcl_oracle as O
name, n, size, mode = "efficientnet-b7", 2, 64, "cam"
cfg = net_cfg(name, False); sd = synth.synth_state_dict(cfg, 23)
x = torch.from_numpy(synth.normal(23, "x", (n, 3, size, size)).astype(np.float32))
du = gu.drop_draws(cfg, n, 5)
for nt in (None, 16, 8, 4, 1):
    if nt:
        torch.set_num_threads(nt)
    for dt in (torch.float32, torch.float64):
        t = time.time(); net = O.OracleNet(name, sd, dtype=dt); net.train()
        outs = net.forward(x.to(dt), mode, du); tf = time.time() - t
        t = time.time(); sum((o * o).sum() for o in outs).backward()
        print("threads", torch.get_num_threads(), dt, "fwd %.2f bwd %.2f" % (tf, time.time() - t), flush=True)
