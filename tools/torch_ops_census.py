"""Which ATen ops (fills, copies, tiny elementwise) the host code still launches per step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, muscle_amd
from muscle_amd import arch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet-b7"
torch.manual_seed(0)
model = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False).to(dev)
opt = muscle_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=5e-5)
batch = bench.make_batch(8, 224, 112, 1000, dev)
vc = int(batch["label"].sum().item())
for _ in range(2): muscle_amd.mcl_step(model, opt, batch, 4, valid_channel=vc)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=False, with_stack=True) as prof:
    muscle_amd.mcl_step(model, opt, batch, 4, valid_channel=vc)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=50))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=30, max_name_column_width=40, max_src_column_width=90))
