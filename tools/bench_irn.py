"""infer_irn.py:76 at a real VOC size: cams [20, 94, 125] (a 375x500 image at the IRN's 1/4 resolution), radius 5, beta 10,
exp_times 8 -> 8 squarings of an 11750^2 matrix.  Reports time and fp32 TFLOP/s, plus the oracle on the host at a
bounded size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_amd import indexing
dev = torch.device("cuda:0")
for (h, w, times) in [(94, 125, 8), (64, 64, 8)]:
    x = torch.rand(1, 20, h, w, device=dev); edge = torch.rand(1, h, w, device=dev) ** 2
    indexing.propagate_to_edge(x, edge, exp_times=times); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps): indexing.propagate_to_edge(x, edge, exp_times=times)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    n = h * w
    fl = times * 2.0 * n ** 3 + 2.0 * 20 * n * n
    print(f"propagate_to_edge {h}x{w} (n={n}), exp_times={times}: {dt*1e3:8.1f} ms  {fl/dt/1e12:6.1f} TFLOP/s fp32", flush=True)
from oracle import mcl_oracle as O
torch.set_num_threads(min(16, os.cpu_count() or 1))
h, w = 40, 52
x = torch.rand(1, 20, h, w); edge = torch.rand(1, h, w) ** 2
t0 = time.perf_counter(); O.irn_propagate_to_edge(x, edge, 5, 10, 8); dt = time.perf_counter() - t0
n = h * w
print(f"oracle (torch-CPU, {torch.get_num_threads()} threads) {h}x{w} (n={n}): {dt*1e3:.1f} ms  {(8*2.0*n**3)/dt/1e12:.2f} TFLOP/s", flush=True)
