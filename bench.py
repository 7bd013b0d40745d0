#!/usr/bin/env python
"""bench.py — images/sec of the MCL training step on MI355X (contract: see the task's bench section).

    python bench.py [--gpus N --steps K --warmup W] [--model efficientnet-b7 --batch 32 --size 448 --epoch 4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one iteration of the train_mcl.py loop body (muscle_amd.mcl_step) on a synthetic batch that is
already resident in HBM.  Rank 0 prints ONE JSON line.  Extra objects:
  roofline     — dominant kernel family (the pointwise-conv GEMMs), timed live with HIP events on the launch stream; every
                 launch is priced against the peak of the pipe it ran on (fp32 MFMA, or bf16 MFMA / 6 for split arithmetic)
  fp32_mfma / split_mfma — the same K steps in the OTHER GEMM arithmetic (config.arithmetic names the one `value` was measured in)
  configs      — short runs of BASELINE.json configs[1] (B0 / 448 / batch 16) and of step-full (epoch-12 gates), N=1 only
  cpu_baseline — the CPU oracle (oracle/mcl_oracle.py, a port of the reference) timed on this host, N=1 only
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from muscle_amd import arch, synth  # noqa: E402
from muscle_amd._host import cpu_share  # noqa: E402

GEMM_CALLS = ("mx_pw_fwd", "mx_pw_fwd_planes", "mx_pw_fwd_planes_act", "mx_pw_dgrad_bnbwd_planes", "mx_pw_dgrad", "mx_pw_wgrad", "mx_pw_wgrad_small", "mx_pw_wgrad_tile",
              "mx_pw_wgrad_tile_bnbwd", "mx_pw_wgrad_small_bnbwd")
MFMA_F32_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 MFMA
MFMA_BF16_PEAK_TFLOPS = 2500.0        # same guide, dense bf16 MFMA (no sparsity)
SPLIT_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0    # split arithmetic: six bf16 products per fp32 product -> 416.7 fp32-equivalent
ARITH_MODE = {"fp32": 0, "split": 1}
ARITH_TEXT = {
    "fp32": "exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32) for every GEMM",
    "split": "fp32 operands split exactly into three bf16 terms (x = h + m + l), six of the nine cross products on "
             "v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16 with fp32 accumulation, for the MFMA-bound forward / data-gradient (K >= 128) and "
             "weight-gradient GEMMs; the others on exact-fp32 MFMA; error vs fp64 equal to the fp32-MFMA kernels' (DESIGN.md section 3)"}
HBM_PEAK_GBPS = 8000.0                # same guide: HBM3E spec; 6290 GB/s is what a float4 copy achieves
# whole-step ceilings per GPU for B7 / 448x448 step-A: exact-fp32 MFMA and HBM (minimum-materialisation schedule)
STEP_CEILING_MFMA_IMGS = 600.0
STEP_CEILING_HBM_IMGS = 974.0
DEFAULT_ARITH = "split"


def make_batch(n, size, view, seed, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    c1, c2, _ = synth.synth_coords(n, view, 2 * view, seed)
    return {
        "img": torch.randn(n, 3, size, size, device=dev, generator=g),
        "view1": torch.randn(n, 3, view, view, device=dev, generator=g),
        "view2": torch.randn(n, 3, view, view, device=dev, generator=g),
        "label": torch.from_numpy(synth.synth_labels(n, seed)).to(dev),
        "coord1": torch.from_numpy(c1).to(dev),
        "coord2": torch.from_numpy(c2).to(dev),
    }


class GemmTimer:
    """HIP events around every pointwise-GEMM launch, recorded on the stream the kernel is enqueued on."""

    def __init__(self):
        self.pairs = []       # (event, event, flop of the launch, ran in split arithmetic?)
        self.shapes = []      # per pair: (entry point, M, N, K) as shape_of() names them
        self.calls = 0        # every mx_* entry point called while the timer is on
        self.dw = []          # (event, event, algorithmic bytes) of the fused depthwise backward: the slowest HBM-bound kernel
        self.on = False

    def install(self):
        from muscle_amd import _lib
        inner = _lib.call
        me = self

        def shape_of(name, a):
            """(flop, kind, M, N, K) of a GEMM entry point from its argument list (include/muscle_hip.h)."""
            if name == "mx_pw_fwd":
                M, K, N = a[8], a[9], a[10]
                return 2.0 * M * K * N, 0, M, N, K
            if name == "mx_pw_fwd_planes":
                M, K, N = a[3], a[4], a[5]
                return 2.0 * M * K * N, 2, M, N, K              # second-generation split kernel: always split
            if name == "mx_pw_fwd_planes_act":
                M, K, N = a[7], a[8], a[9]
                return 2.0 * M * K * N, 2, M, N, K
            if name == "mx_pw_dgrad_bnbwd_planes":
                M, K, N = a[5], a[6], a[7]
                return 2.0 * M * K * N, 2, M, N, K
            if name == "mx_pw_wgrad_tile_bnbwd":
                R, Co, Ci = a[5], a[6], a[7]
                return 2.0 * R * Co * Ci, 2, Co, Ci, R           # split-arithmetic tiled kernel only
            if name == "mx_pw_wgrad_small_bnbwd":
                R, Co, Ci = a[5], a[6], a[7]
                return 2.0 * R * Co * Ci, -1, Co, Ci, R          # exact-fp32 small-output kernel
            if name == "mx_pw_dgrad":
                M, K, N = a[3], a[4], a[5]
                return 2.0 * M * K * N, -1, M, N, K             # NN kernel: never split
            R, Co, Ci = a[8], a[9], a[10]
            return 2.0 * R * Co * Ci, (1 if name == "mx_pw_wgrad_tile" else -1), Co, Ci, R

        def timed(name, *a):
            if me.on:
                me.calls += 1
            if me.on and name in GEMM_CALLS:
                flop, kind, M, N, K = shape_of(name, a)
                split = kind == 2 or (kind >= 0 and bool(_lib.lib().mx_gemm_uses_split(kind, M, N, K)))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                inner(name, *a)
                e1.record()
                me.pairs.append((e0, e1, flop, split))
                me.shapes.append((name, M, N, K))
            elif me.on and name == "mx_dwconv_bwd_fused":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                inner(name, *a)
                e1.record()
                n, h, w, c, k = a[18:23]
                # algorithmic traffic: read dA, D, X once, write gX once (fp32)
                me.dw.append((e0, e1, 4.0 * 4 * n * h * w * c, k))
            else:
                inner(name, *a)

        from muscle_amd import ops, loss_multilabel, train_step
        for mod in (_lib, ops, loss_multilabel, train_step):
            if hasattr(mod, "call"):
                mod.call = timed

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b, _, _ in self.pairs), len(self.pairs)

    def blended_peak(self):
        """(peak TFLOP/s the timed launches could reach if each ran at the peak of ITS pipe, share of the flop that ran split)."""
        f_all = sum(f for _, _, f, _ in self.pairs)
        f_split = sum(f for _, _, f, sp in self.pairs if sp)
        if f_all <= 0:
            return MFMA_F32_PEAK_TFLOPS, 0.0
        t_ideal = f_split / SPLIT_PEAK_TFLOPS + (f_all - f_split) / MFMA_F32_PEAK_TFLOPS
        return f_all / t_ideal, f_split / f_all

    def table(self, steps):
        """Per (entry point, shape): launches per step, us per launch, TFLOP/s, and the time the launch's bound allows - the larger of
        flop / (peak of its pipe) and algorithmic bytes (operands read once, result written once, fp32) / 5 TB/s (what the HBM-bound
        kernels of this step reach) - so that the ratio names the shapes that are far from THEIR roofline."""
        agg = {}
        for (a, b, flop, split), key in zip(self.pairs, self.shapes):
            e = agg.setdefault(key + (split,), [0, 0.0, flop])
            e[0] += 1
            e[1] += a.elapsed_time(b)
        rows = []
        for (name, M, N, K, split), (n, ms, flop) in agg.items():
            by = 4.0 * (M * K + N * K + M * N)
            t_mfma = flop / ((SPLIT_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS) * 1e12) * 1e6
            t_hbm = by / 5e12 * 1e6
            us = ms * 1e3 / n
            rows.append({"entry": name, "M": M, "N": N, "K": K, "split": bool(split), "per_step": n / steps, "us": us,
                         "ms_per_step": ms / steps, "tflops": flop / us / 1e6, "bound": "mfma" if t_mfma > t_hbm else "hbm",
                         "bound_us": max(t_mfma, t_hbm), "x_bound": us / max(t_mfma, t_hbm)})
        rows.sort(key=lambda r: -r["ms_per_step"])
        return rows

    def dw_summary(self):
        """Per kernel size of the fused depthwise backward: launches, ms, GB/s of its algorithmic 4 passes."""
        out = {}
        for k in (3, 5):
            sel = [(a.elapsed_time(b), by) for a, b, by, kk in self.dw if kk == k]
            if sel:
                ms = sum(t for t, _ in sel)
                out[f"k{k}"] = {"launches": len(sel), "ms": ms, "GBps": sum(by for _, by in sel) / (ms * 1e-3) / 1e9}
        return out


def pointwise_flops_per_image(cfg, size):
    m = arch.forward_macs(cfg, size)
    stem_h = cfg.stem_out_size(size)
    stem = 28 * cfg.stem_out * stem_h * stem_h          # stem runs as a K=28 GEMM (27 taps + pad)
    # forward + data gradient + weight gradient, 2 flop per MAC; the stem has no data gradient
    return 6 * m["pointwise"] + 4 * stem


def _cpu_steps(model_name, size, view, ep, n, warm, timed_n, seconds_budget):
    from oracle import mcl_oracle as O
    cfg = arch.net_cfg(model_name, False)
    torch.manual_seed(0)
    net = O.OracleNet(model_name, synth.synth_state_dict(cfg, 0))
    opt = O.OracleAdam(net.parameters())
    b = {k: torch.from_numpy(v) for k, v in synth.synth_batch(n, size, view, 0).items()}
    b["label"][1] = b["label"][0]
    times = []
    t_start = time.time()
    for i in range(warm + timed_n):
        t0 = time.time()
        O.mcl_step(net, opt, b, ep)
        times.append(time.time() - t0)
        if i >= warm and time.time() - t_start > seconds_budget:
            break
    timed = times[warm:] if len(times) > warm else times[-1:]
    return sum(timed) / len(timed), len(timed)


def cpu_baseline(model_name, size, view, ep, seconds_budget=25.0):
    """The oracle (a CPU port of the reference's loop body) on this host's cores; bounded sample: 2 warm-up + up to 5
    timed steps of the headline model at batch 2, and BASELINE.json configs[0] (EfficientNet-B0, 2 images, 224x224)."""
    torch.set_num_threads(max(1, min(16, cpu_share())))     # the GPU box's CPU share for one GPU (cgroup quota, not os.cpu_count())
    n = 2
    sec, cnt = _cpu_steps(model_name, size, view, ep, n, 2, 5, seconds_budget)
    sec0, cnt0 = _cpu_steps("efficientnet-b0", 224, 112, ep, n, 2, 5, 10.0)
    return {"value": n / sec, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{cnt} timed steps (2 warm-up) of the same loop body, {model_name} {size}x{size}, batch {n}, "
                      f"epoch-{ep} semantics, oracle/mcl_oracle.py on torch-CPU fp32, {sec:.2f} s/step",
            "config1": {"value": n / sec0, "unit": "images/sec",
                        "sample": f"{cnt0} timed steps (2 warm-up), efficientnet-b0 224x224, batch {n}, {sec0:.3f} s/step"}}


def _calibrate_bn(model, x):
    """phase 2 runs in eval mode (train_mcl.py:196): give the random-init model BatchNorm running statistics of its own
    activations (one train-mode pass at momentum 1.0), as SURVEY 8(c) prescribes for eval-mode work"""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    saved = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(x, cam="pix")
    for m, mo in zip(bns, saved):
        m.momentum = mo


def short_run(model_name, batch_n, size, epoch, warm, steps, dev, seed, graph=False):
    """A short eager run of another BASELINE.json configuration on this GPU (same code path as the headline measurement),
    so that its figure is driver-timed too."""
    import muscle_amd
    torch.manual_seed(0)
    model = muscle_amd.MuSCLe(21, model_name, layers=3, last_pooling=False).to(dev)
    opt = muscle_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=5e-5)
    batch = make_batch(batch_n, size, size // 2, seed, dev)
    if epoch >= 8:
        _calibrate_bn(model, batch["view1"])
    for _ in range(warm):
        muscle_amd.mcl_step(model, opt, batch, epoch, valid_channel=batch["label"].sum())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        muscle_amd.mcl_step(model, opt, batch, epoch, valid_channel=batch["label"].sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"value": batch_n * steps / dt, "unit": "images/sec", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warm,
           "launch": "eager", "workload": f"{model_name} {size}x{size} batch {batch_n}, epoch-{epoch} gates"}
    if graph and epoch < 8:
        # A B0-sized step is ~600 launches of ~15 us: launched from Python it is as fast as the HOST is that minute (9.3-11.3 ms on the
        # boxes of round 5); replayed from a captured hipGraph (muscle_amd.GraphedStep, the product's own API) it does not depend on the
        # host but runs the weight-gradient side stream in line.  Both are timed; `value` is the better one and says which.
        gstep = muscle_amd.GraphedStep(model, opt, epoch)
        for _ in range(gstep.warmup + 2):
            gstep(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gstep(batch)
        torch.cuda.synchronize()
        dg = time.perf_counter() - t0
        gstep.close()
        res["eager"] = {"value": res["value"], "ms_per_step": res["ms_per_step"]}
        res["graph_replay"] = {"value": batch_n * steps / dg, "ms_per_step": dg / steps * 1e3}
        if dg < dt:
            res.update(value=batch_n * steps / dg, ms_per_step=dg / steps * 1e3, launch="hipGraph replay (muscle_amd.GraphedStep)")
    return res


def short_run_dec(dev, warm=2, steps=10):
    """BASELINE.json configs[3] on one GPU: train_muscle.py loop body (decoder mode + BEACON FieldLoss, lambda 0.05, k 128),
    EfficientNet-B7 448x448 batch 16, last_pooling=True, synthetic soft pseudo-labels."""
    import numpy as np
    import muscle_amd
    n, size = 16, 448
    torch.manual_seed(0)
    model = muscle_amd.MuSCLe(21, "efficientnet-b7", layers=3, last_pooling=True, mode="dec").to(dev)
    opt = muscle_amd.FusedAdam(model.live_parameters("seg") if hasattr(model, "live_parameters") else model.parameters(), lr=1e-5, weight_decay=5e-5)
    label = synth.synth_labels(n, 7)
    batch = {"img": torch.from_numpy(synth.normal(7, "img", (n, 3, size, size)).astype(np.float32)).to(dev),
             "label": torch.from_numpy(label).to(dev),
             "mask": torch.from_numpy(synth.synth_soft_mask(label, size, 7)).to(dev)}
    for _ in range(warm):
        muscle_amd.muscle_step(model, opt, batch, lamb=0.05, step=7, k=128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        muscle_amd.muscle_step(model, opt, batch, lamb=0.05, step=7, k=128)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    v = n * steps / dt
    return {"value": v, "unit": "images/sec", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warm,
            "ceiling_fp32_mfma": 1290.0, "frac_of_ceiling": v / 1290.0,
            "arithmetic": "split" if muscle_amd.get_gemm_mode() >= 1 else "fp32",
            "ceiling_split_mfma": round(1290.0 * (2500.0 / 6.0) / 157.3, 1), "frac_of_split_ceiling": v / (1290.0 * (2500.0 / 6.0) / 157.3),
            "workload": "train_muscle.py loop body (decoder + BEACON, lambda 0.05, k 128), efficientnet-b7 448x448 batch 16, last_pooling=True"}


def short_run_infer(dev, steps=3):
    """BASELINE.json configs[4]: infer_mcl.py's batched eval forward (cam_lr), EfficientNet-B7, batch 64, 448 / 512 / 768."""
    import muscle_amd
    torch.manual_seed(0)
    model = muscle_amd.MuSCLe(21, "efficientnet-b7", layers=3, last_pooling=False).to(dev).eval()
    model.fold_eval_bn()
    out = {}
    for size, ceil in ((448, 1800.0), (512, 1380.0), (768, 613.0)):
        x = torch.randn(64, 3, size, size, device=dev)
        with torch.no_grad():
            model(x, cam="cam_lr")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                model(x, cam="cam_lr")
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        # the forward runs in the library's default (split) arithmetic: its matrix ceiling is the bf16 pipe's 2500 / 6 TFLOP/s, 2.65 x the
        # exact-fp32 one; both are quoted
        split_ceil = ceil * (2500.0 / 6.0) / 157.3
        out[str(size)] = {"value": 64 / dt, "unit": "images/sec", "ms_per_batch": dt * 1e3, "steps": steps,
                          "arithmetic": "split" if muscle_amd.get_gemm_mode() >= 1 else "fp32",
                          "ceiling_fp32_mfma": ceil, "frac_of_ceiling": 64 / dt / ceil,
                          "ceiling_split_mfma": round(split_ceil, 1), "frac_of_split_ceiling": 64 / dt / split_ceil}
        del x
    out["workload"] = "infer_mcl.py eval forward (cam='cam_lr', BatchNorm folded), efficientnet-b7, batch 64, 448 / 512 / 768"
    return out


def main():
    # host threads: the GPU boxes show 256 CPUs under a cgroup quota of 16; torch's default of 128 intra-op threads makes every CPU-side
    # torch call that parallelises 10-30x slower there (profiles/r05_cpu_probe.txt).  Per rank: the quota shared by the ranks of the node.
    torch.set_num_threads(max(1, min(16, cpu_share() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="efficientnet-b7")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--epoch", type=int, default=4, help="epoch gate semantics of train_mcl.py (4: cls+ER+IMC)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arith", choices=["split", "fp32"], default=os.environ.get("MUSCLE_BENCH_ARITH", DEFAULT_ARITH),
                    help="GEMM arithmetic of the timed steps (config.arithmetic); the other one is timed as well and reported beside it")
    ap.add_argument("--no-split", "--no-other-arith", dest="no_other", action="store_true",
                    help="skip the extra K steps in the other GEMM arithmetic")
    ap.add_argument("--gemm-table", default=None, help="write the per-shape table of the pointwise GEMM launches (GemmTimer.table) here")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of configs[1] and step-full")
    ap.add_argument("--graph", action="store_true",
                    help="timed steps replay phase 1 from a captured hipGraph (muscle_amd.GraphedStep; one GPU, epoch < 8).  Not "
                         "the default: measured on MI355X the replay saves 0.5 ms of 136 on B7 but serialises the weight-gradient "
                         "side stream, which is worth 3.6 ms on B7 when launched eagerly; it is the better choice for B0-sized models")
    ap.add_argument("--eager", action="store_true", help="(the default; kept for older command lines)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    import torch.distributed as dist
    # MUSCLE_DIST_BACKEND=gloo + MUSCLE_SHARE_GPU=1: rehearsal of the N>1 flow with all ranks on one GPU (tests only)
    backend = os.environ.get("MUSCLE_DIST_BACKEND", "nccl")
    if os.environ.get("MUSCLE_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import muscle_amd
    from muscle_amd.dist import GradAverager, broadcast_parameters
    cfg = arch.net_cfg(a.model, False)
    torch.manual_seed(0)
    model = muscle_amd.MuSCLe(21, a.model, layers=3, last_pooling=False).to(dev)
    broadcast_parameters(model)
    opt = muscle_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=5e-5)
    view = a.size // 2
    batch = make_batch(a.batch, a.size, view, 1000 + rank, dev)
    hook = GradAverager().attach(model) if world > 1 else None     # chunks go out as backward fills the arena
    timer = GemmTimer()
    timer.install()
    other = "fp32" if a.arith == "split" else "split"
    muscle_amd.set_gemm_mode(ARITH_MODE[a.arith])

    def step():
        # ER's top-k count int(0.2 * label.sum() * H * W) (train_mcl.py:178,188) is taken INSIDE the step, on the device (the
        # reference reads it back every iteration; a count precomputed outside the timed region would be work skipped)
        return muscle_amd.mcl_step(model, opt, batch, a.epoch, valid_channel=batch["label"].sum(), grad_hook=hook)

    if a.graph and (world > 1 or a.epoch >= 8):
        raise SystemExit("--graph covers phase 1 on one GPU (epoch < 8)")
    eager_step = step
    if a.graph:
        gstep = muscle_amd.GraphedStep(model, opt, a.epoch)
        for _ in range(gstep.warmup + 1):          # eager settling steps + the capturing call
            gstep(batch)
        step = lambda: gstep(batch)                # noqa: E731  (copies the batch into the static buffers, replays)

    if a.epoch >= 8:
        _calibrate_bn(model, batch["view1"])
    for _ in range(a.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    host_t0 = [0.0]

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    # Per-launch HIP events for the roofline: the same work enqueued eagerly right after the timed region, with the
    # weight-gradient side stream off.  The timed steps cannot carry them: graph replays have no host-side launch to
    # bracket, and a weight-gradient GEMM that shares the chip with HBM-bound kernels has no duration of its own.
    from muscle_amd import engine
    overlap = engine.WGRAD_SIDE_STREAM
    engine.WGRAD_SIDE_STREAM = False
    inst_steps = min(a.steps, 3)
    timer.on = (rank == 0)
    t1 = time.perf_counter()
    for _ in range(inst_steps):
        eager_step()
    barrier()
    inst_dt = time.perf_counter() - t1
    timer.on = False
    calls_main = timer.calls
    engine.WGRAD_SIDE_STREAM = overlap
    # Beside the contract value: the same K steps in the OTHER GEMM arithmetic (include/muscle_hip.h mx_set_gemm_mode)
    beside = None
    if not a.no_other:
        muscle_amd.set_gemm_mode(ARITH_MODE[other])
        eager_step()
        barrier()
        t2 = time.perf_counter()
        for _ in range(a.steps):
            eager_step()
        barrier()
        dts = time.perf_counter() - t2
        muscle_amd.set_gemm_mode(ARITH_MODE[a.arith])
        if world > 1:
            t = torch.tensor([dts], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t)
        beside = {"value": a.batch * world * a.steps / dts, "unit": "images/sec", "ms_per_step": dts / a.steps * 1e3,
                  "arithmetic": ARITH_TEXT[other]}
        # the same per-launch events in this arithmetic, so that both roofline fractions are in the line
        main_pairs, main_dw, main_shapes = timer.pairs, timer.dw, timer.shapes
        timer.pairs, timer.dw, timer.shapes = [], [], []
        muscle_amd.set_gemm_mode(ARITH_MODE[other])
        engine.WGRAD_SIDE_STREAM = False
        timer.on = (rank == 0)
        for _ in range(inst_steps):
            eager_step()
        barrier()
        timer.on = False
        engine.WGRAD_SIDE_STREAM = overlap
        muscle_amd.set_gemm_mode(ARITH_MODE[a.arith])
        if rank == 0:
            o_ms, o_n = timer.total_ms()
            o_peak, o_share = timer.blended_peak()
            o_flops = pointwise_flops_per_image(cfg, a.size) * a.batch * inst_steps
            if a.epoch < 12 and o_ms > 0:
                o_ach = o_flops / (o_ms * 1e-3) / 1e12
                beside["roofline"] = {"bound": "mfma", "achieved": o_ach, "peak": o_peak, "unit": "TFLOP/s", "frac": o_ach / o_peak,
                                      "flop_share_split": o_share, "gemm_ms_per_step": o_ms / inst_steps,
                                      "avg_launch_us": o_ms * 1e3 / max(o_n, 1)}
        timer.pairs, timer.dw, timer.shapes = main_pairs, main_dw, main_shapes
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    imgs = a.batch * world * a.steps
    gemm_ms, gemm_launches = timer.total_ms()
    if a.gemm_table:
        rows = timer.table(inst_steps)
        with open(a.gemm_table, "w") as f:
            f.write("# pointwise-GEMM launches of one %s step (%s, %dx%d, batch %d), side stream off, HIP events per launch; bound_us = max(flop / "
                    "pipe peak, fp32 operand + result bytes / 5 TB/s); x_bound = us / bound_us\n" % (a.arith, a.model, a.size, a.size, a.batch))
            f.write("%-26s %7s %5s %7s %5s %5s %8s %8s %7s %5s %8s %7s\n" % ("entry", "M", "N", "K", "split", "/step", "us", "ms/step", "TFLOP/s",
                                                                          "bound", "bound_us", "x_bound"))
            for r in rows:
                f.write("%-26s %7d %5d %7d %5s %5.1f %8.1f %8.3f %7.1f %5s %8.1f %7.2f\n" % (
                    r["entry"], r["M"], r["N"], r["K"], "y" if r["split"] else "n", r["per_step"], r["us"], r["ms_per_step"], r["tflops"],
                    r["bound"], r["bound_us"], r["x_bound"]))
            f.write("# total %.2f ms/step; at the bounds %.2f ms/step\n" % (sum(r["ms_per_step"] for r in rows),
                                                                          sum(r["bound_us"] * r["per_step"] for r in rows) / 1e3))
    full = a.epoch >= 12
    flops_img = pointwise_flops_per_image(cfg, a.size)
    if full:
        # phase 2 (train_mcl.py:196-229): two view forwards (one of them under no_grad) + one view backward
        mv = arch.forward_macs(cfg, view)
        sv = cfg.stem_out_size(view)
        stem_v = 28 * cfg.stem_out * sv * sv
        flops_img += 2 * 2 * mv["pointwise"] + 4 * mv["pointwise"] + 2 * 2 * stem_v + 2 * stem_v
    flops = flops_img * a.batch * inst_steps
    achieved = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None
    peak, split_share = timer.blended_peak()
    traffic, tj = None, {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and (a.model, a.size, a.batch, full) == ("efficientnet-b7", 448, 32, False):
        tj = json.load(open(tpath))
        traffic = tj.get("gemm_hbm_bytes_per_launch")     # measured for this workload only
    res = {
        "metric": "images/sec (whole node), MCL EfficientNet-B7 448x448 bs=32/GPU" if (a.model, a.size, a.batch) == ("efficientnet-b7", 448, 32)
        else f"images/sec (whole node), MCL {a.model} {a.size}x{a.size} bs={a.batch}/GPU",
        "value": imgs / dt, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"train_mcl.py loop body, step-full (epoch-{a.epoch} gates: phase 1 focal+softmargin+pairwise+ER+IMC, "
                                f"backward, Adam; phase 2 PixPro+EMD on two {view}x{view} views, backward, Adam), " if full else
                                f"train_mcl.py loop body, step-A (epoch-{a.epoch} gates: focal+softmargin+pairwise+ER"
                                f"{'+IMC' if a.epoch >= 4 else ''}, one backward, one Adam step), ") +
                               f"MuSCLe({a.model}, last_pooling=False, 21 classes), random-init weights",
                   "per_gpu_batch": a.batch, "global_batch": a.batch * world, "image": f"{a.size}x{a.size}",
                   "parallelism": f"dp{world}" if world > 1 else "single", "optimizer": "Adam(lr=1e-4, wd=5e-5) fused",
                   "launch": ("hipGraph replay" if a.graph else "eager") + (", weight-gradient GEMMs on a second stream" if overlap else ""),
                   "arithmetic": f"{a.arith}: {ARITH_TEXT[a.arith]}"},
        "losses": {k: (float(v.detach()) if torch.is_tensor(v) else v) for k, v in out.items()},
        "roofline": {"bound": "mfma",
                     "kernel": "gemm_nt_split3_kernel<*> / gemm_nt_kernel<*> / wgrad_split_ws_kernel / wgrad_split_kernel<*> / wgrad_tile_kernel<*> / wgrad_small_kernel<*> / "
                               "gemm_kernel<*> (the pointwise-conv GEMMs: forward, data gradient, weight gradient; stem)",
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": (achieved / peak) if achieved else None, "traffic": traffic,
                     "peak_is": f"flop-weighted over the timed launches: {split_share:.3f} of the flop ran in split arithmetic on the bf16 "
                                f"pipe (peak {MFMA_BF16_PEAK_TFLOPS:.0f} / 6 = {SPLIT_PEAK_TFLOPS:.1f} TFLOP/s fp32-equivalent), the rest on "
                                f"exact-fp32 MFMA (peak {MFMA_F32_PEAK_TFLOPS})",
                     "flop_share_split": split_share,
                     # not measured by this run: the in-kernel clock of the split forward / data-gradient kernel after 1.5 s of back-to-back
                     # launches (tools/hip/gemm_lab stamps M K N 3, profiles/r04_planes_stamps.txt); `peak` above is quoted at 2.4 GHz
                     "clock_held_under_load_GHz": [1.84, 2.01],
                     "measured_over": f"{inst_steps} eager steps right after the timed region (HIP events per launch, weight-gradient "
                                      f"side stream off; {inst_dt / inst_steps * 1e3:.1f} ms/step in that mode)",
                     "launches_per_step": gemm_launches // max(inst_steps, 1),
                     "avg_launch_us": gemm_ms * 1e3 / max(gemm_launches, 1),
                     "time_share_of_step": gemm_ms * 1e-3 / inst_dt,
                     "gemm_ms_per_step": gemm_ms / max(inst_steps, 1),
                     "non_gemm_ms_per_step": (inst_dt * 1e3 - gemm_ms) / max(inst_steps, 1),
                     "algorithmic_gflop_per_image": flops_img / 1e9},
    }
    if (a.model, a.size, full) == ("efficientnet-b7", 448, False):
        # whole-step fractions of the two ceilings of SURVEY.md section 8(d) (per GPU), and the slowest HBM-bound kernel
        per_gpu = imgs / dt / world
        res["roofline"]["step_frac_mfma"] = per_gpu / STEP_CEILING_MFMA_IMGS
        res["roofline"]["step_frac_hbm"] = per_gpu / STEP_CEILING_HBM_IMGS
        # whole-step traffic and launch count from the committed PMC / kernel-trace pass of this workload (profiles/traffic.json)
        res["roofline"]["step_traffic_GB"] = tj.get("step_traffic_GB")
        res["roofline"]["step_traffic_algorithmic_GB"] = 6.46 * a.batch            # minimum-materialisation schedule
        res["roofline"]["launches_per_step_all"] = tj.get("launches_per_step_all")
        res["roofline"]["mx_calls_per_step"] = calls_main // max(inst_steps, 1)
    if beside is not None:
        res["fp32_mfma" if other == "fp32" else "split_mfma"] = beside
    dws = timer.dw_summary()
    if dws:
        worst = min(dws.items(), key=lambda kv: kv[1]["GBps"])
        res["roofline"]["hbm_kernel"] = {"kernel": f"dw_bwd_fused_kernel<{worst[0][1:]}> (stride-1 depthwise backward, 3 reads + 1 write)",
                                         "achieved": worst[1]["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                         "frac": worst[1]["GBps"] / HBM_PEAK_GBPS,
                                         "ms_per_step": worst[1]["ms"] / max(inst_steps, 1),
                                         "all": {k: {"GBps": round(v["GBps"], 1), "ms_per_step": round(v["ms"] / max(inst_steps, 1), 3)} for k, v in dws.items()}}
    if world == 1 and not a.no_configs and (a.model, a.size, a.batch, full) == ("efficientnet-b7", 448, 32, False):
        del model, opt, out
        torch.cuda.empty_cache()
        res["configs"] = {"config2": short_run("efficientnet-b0", 16, 448, a.epoch, 10, 50, dev, 2000, graph=True)}      # BASELINE.json configs[1]
        res["configs"]["config2"].update(ceiling_hbm=6550.0, frac_of_ceiling=res["configs"]["config2"]["value"] / 6550.0)
        torch.cuda.empty_cache()
        res["configs"]["stepfull"] = short_run("efficientnet-b7", 32, 448, 12, 2, 10, dev, 1000)       # epoch >= 12 gates, headline model
        torch.cuda.empty_cache()
        res["configs"]["config4_1gpu"] = short_run_dec(dev)                                            # BASELINE.json configs[3], one GPU
        torch.cuda.empty_cache()
        res["configs"]["config5"] = short_run_infer(dev)                                               # BASELINE.json configs[4]
        torch.cuda.empty_cache()
    muscle_amd.set_gemm_mode(0)
    if world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a.model, a.size, view, a.epoch)
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
