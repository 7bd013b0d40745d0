"""Quick timing of the HIP MBConv chain (forward + backward), not the contract bench."""
import argparse, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from muscle_amd import synth, engine
from muscle_amd.arch import net_cfg, forward_macs
from muscle_amd.efficientnet import EfficientNet

ap = argparse.ArgumentParser()
ap.add_argument("--name", default="efficientnet-b7"); ap.add_argument("--n", type=int, default=32)
ap.add_argument("--size", type=int, default=448); ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = net_cfg(a.name, False)
bb = EfficientNet(cfg, 21).to(dev)
x = torch.randn(a.n, 3, a.size, a.size, device=dev)
def step():
    tape = engine.backbone_forward(bb, cfg, x, True)
    t7 = tape.blocks[cfg.taps[6]].out
    sink = engine.GradSink()
    engine.backbone_backward(bb, cfg, tape, {cfg.taps[6]: torch.ones_like(t7)}, sink)
step(); torch.cuda.synchronize()
print("peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
t0 = time.time()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = (time.time() - t0) / a.steps
fl = 6 * forward_macs(cfg, a.size)["total"] * a.n
print(f"{a.name} n={a.n} size={a.size}: {dt*1e3:.1f} ms/step fwd+bwd, {a.n/dt:.1f} img/s, {fl/dt/1e12:.1f} TFLOP/s")
