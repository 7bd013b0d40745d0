import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, muscle_amd
sys.argv = sys.argv[:1]
import bench
dev = torch.device("cuda:0")
model = muscle_amd.MuSCLe(21, "efficientnet-b7", layers=3, last_pooling=False).to(dev)
opt = muscle_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=5e-5)
batch = bench.make_batch(32, 448, 224, 1000, dev)
vc = batch["label"].sum()
for _ in range(3): muscle_amd.mcl_step(model, opt, batch, 12, valid_channel=vc)
torch.cuda.synchronize()
for _ in range(4):
    t0 = time.perf_counter()
    muscle_amd.mcl_step(model, opt, batch, 12, valid_channel=vc)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"ep12: host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
