"""Worker of the data-parallel tests (test infrastructure): one rank of an N-process run of mcl_step on a shared GPU over
gloo (tests/test_gpu_dist.py), or - with argv[2] == 'cpu-arena' - of the chunked-exchange check on CPU tensors."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
mode, out_path = sys.argv[2], sys.argv[3]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("DP_BACKEND", "gloo")            # "nccl" (= RCCL): one GPU per rank, needs >= world GPUs
force1 = os.environ.get("DP_SINGLE_RANK_RCCL") == "1"   # world 1 over "nccl" with the collectives forced on: the RCCL branch on a one-GPU box
if world > 1 or force1:
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)

from muscle_amd.dist import GradAverager  # noqa: E402


class _Sink:
    pass


class _Model:
    pass


if mode == "cpu-arena":
    # the chunked, progress-driven exchange must give exactly what one all-reduce of the whole arena gives
    n = 1003
    base = torch.arange(n, dtype=torch.float32) * (rank + 1) + rank * 0.25
    res = {}
    for tag, chunk_bytes, early in (("single", 1 << 30, False), ("chunked", 4 * 64, True), ("chunked_late", 4 * 64, False)):
        m = _Model()
        m.last_grad_sink = _Sink()
        m.last_grad_sink.arena = base.clone()
        h = GradAverager(chunk_bytes=chunk_bytes)
        if early:
            for lo in (900, 640, 641, 130, 0):          # backward reports progress back to front (not on chunk borders)
                h.on_ready(m.last_grad_sink, lo)
            assert h.launched_early > 0
        h(m, 1)
        assert h.bytes_reduced == n * 4, (tag, h.bytes_reduced)
        res[tag] = m.last_grad_sink.arena.numpy().copy()
    np.savez(out_path, **res)
else:
    import muscle_amd
    from muscle_amd import synth
    from muscle_amd.arch import net_cfg
    name, n, size, view, ep, seed = "efficientnet-b0", 4, 64, 32, 4, 5
    dev = torch.device("cuda", rank if (world > 1 and backend == "nccl") else 0)
    cfg = net_cfg(name, False)
    sd = synth.synth_state_dict(cfg, seed)
    model = muscle_amd.MuSCLe(21, name, layers=3, last_pooling=False)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.to(dev)
    opt = muscle_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=5e-5)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_batch(n, size, view, seed).items()}   # SAME batch on every rank
    torch.manual_seed(3)
    du = {b.index: torch.rand(n).to(dev) for b in cfg.blocks if b.skip and b.drop_rate}
    hook = None
    if world > 1 or force1:
        from muscle_amd.dist import broadcast_parameters, sync_buffers_from_rank0
        hook = GradAverager(chunk_bytes=256 * 1024, single_rank_collectives=force1).attach(model)      # ~16 MB arena -> dozens of chunks
        if force1:
            broadcast_parameters(model, single_rank_collectives=True)                                   # (RCCL broadcast of every tensor)
    out = muscle_amd.mcl_step(model, opt, batch, ep, drop_u=du, grad_hook=hook)
    torch.cuda.synchronize()
    res = {"arena": model.last_grad_sink.arena.cpu().numpy(),
           "losses": np.array([float(out[k]) for k in ("loss_focal", "loss_softmargin", "loss_pair", "loss_er", "loss_imc")]),
           "early": np.array(hook.launched_early if hook else 0),
           "params": torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()}
    np.savez(out_path, **res)
if world > 1 or force1:
    dist.destroy_process_group()
print("ok", rank)
