"""cProfile of the host side of one training step (where do the ~15 us per launch go?).  usage: host_profile.py [model] [batch]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, muscle_amd
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet-b0"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sys.argv = sys.argv[:1]
import bench
dev = torch.device("cuda:0")
model = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False).to(dev)
opt = muscle_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=5e-5)
batch = bench.make_batch(bs, 448, 224, 1000, dev)
for _ in range(5):
    muscle_amd.mcl_step(model, opt, batch, 4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    muscle_amd.mcl_step(model, opt, batch, 4)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e2*(t1-t0):.2f} ms/step, total {1e2*(t2-t0):.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    muscle_amd.mcl_step(model, opt, batch, 4)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
