"""BASELINE.json configs[3]: train_muscle.py loop body (decoder mode + BEACON FieldLoss), EfficientNet-B7 448x448 batch 16
per GPU, synthetic soft pseudo-labels.  Not the contract bench."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import muscle_amd
from muscle_amd import synth
dev = torch.device("cuda:0")
N, S = 16, 448
torch.manual_seed(0)
model = muscle_amd.MuSCLe(21, "efficientnet-b7", layers=3, last_pooling=True, mode="dec").to(dev)
opt = muscle_amd.FusedAdam(model.live_parameters("seg") if hasattr(model, "live_parameters") else model.parameters(), lr=1e-5, weight_decay=5e-5)
label = synth.synth_labels(N, 7)
batch = {"img": torch.from_numpy(synth.normal(7, "img", (N, 3, S, S)).astype(np.float32)).to(dev),
         "label": torch.from_numpy(label).to(dev),
         "mask": torch.from_numpy(synth.synth_soft_mask(label, S, 7)).to(dev)}
for lamb in (0.05, 0.0):
    for _ in range(2): out = muscle_amd.muscle_step(model, opt, batch, lamb=lamb, step=7, k=128)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = 4
    for _ in range(steps): out = muscle_amd.muscle_step(model, opt, batch, lamb=lamb, step=7, k=128)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"muscle_step B7 dec 448x448 bs{N} lamb={lamb}: {dt*1e3:7.1f} ms/step  {N/dt:6.1f} img/s  "
          f"loss_seg {float(out['loss_seg']):.4f} beacon {float(out['loss_beacon']) if torch.is_tensor(out['loss_beacon']) else out['loss_beacon']}  peak mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
