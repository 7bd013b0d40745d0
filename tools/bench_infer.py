"""BASELINE.json configs[4]: CAM generation forward (infer_mcl.py), B7, square synthetic inputs 448/512/768 at batch 64,
eval mode, plus the fused post-processing (one 500x375 'original image', 8 passes).  Not the contract bench."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import muscle_amd
from muscle_amd import infer, arch
from muscle_amd._lib import call, ptr, stream

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="efficientnet-b7"); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--sizes", default="448,512,768"); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--forward-only", action="store_true", help="only the batched eval forward (for kernel profiles)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = muscle_amd.MuSCLe(21, a.model, layers=3, last_pooling=False).to(dev).eval()
if os.environ.get('MUSCLE_EVAL_FOLD', '1') == '1':
    model.fold_eval_bn()          # BN folded into the conv weights once per model load
cfg = arch.net_cfg(a.model, False)
for size in (int(s) for s in a.sizes.split(",")):
    x = torch.randn(a.batch, 3, size, size, device=dev)
    with torch.no_grad():
        model(x, cam="cam_lr"); torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            model(x, cam="cam_lr")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    fl = 2 * arch.forward_macs(cfg, size)["total"] * a.batch
    gb = arch.min_materialisation_bytes(cfg, size, inference=True) * a.batch if "inference" in arch.min_materialisation_bytes.__code__.co_varnames else None
    print(f"{a.model} eval forward cam_lr  batch {a.batch} {size}x{size}: {dt*1e3:8.1f} ms  {a.batch/dt:7.1f} img/s  "
          f"{fl/dt/1e12:6.1f} TFLOP/s  peak mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
    del x
if a.forward_only:
    sys.exit(0)
# post-processing of one image: 8 passes (4 scales x flip) accumulated into [20,375,500] for CAM and SGC, then normalised
H, W, K = 375, 500, 21
acc = torch.zeros(K - 1, H, W, device=dev)
maps = []
for s in (0.5, 1.0, 1.5, 2.0):
    hs, ws = int(round(H * s)), int(round(W * s))
    h, w = (hs + 15) // 16, (ws + 15) // 16
    maps += [(torch.rand(1, h, w, 24, device=dev), hs, ws)] * 2
def post():
    acc.zero_()
    for i, (m, hs, ws) in enumerate(maps):
        call("mx_infer_accum", ptr(m), ptr(acc), m.shape[1], m.shape[2], 24, K, hs, ws, H, W, i % 2, stream())
    call("mx_infer_norm", ptr(acc), K - 1, H * W, stream())
post(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): post()
torch.cuda.synchronize()
print(f"post-processing (8 passes -> [20,{H},{W}] sum + min-max, one of CAM/SGC): {(time.perf_counter()-t0)/20*1e6:.0f} us", flush=True)

# infer_mcl.py's loop body per image, end to end: multi-scale + flip list (built on the device by MSFStager, or on the host
# with PIL as VOC12ClsDatasetMSF does), 8 forwards at batch 1, accumulate / normalise, the kept channels to the host.
import PIL.Image
from muscle_amd.data import MSFStager
rng = np.random.default_rng(0)
pil = PIL.Image.fromarray(rng.integers(0, 256, (375, 500, 3), dtype=np.uint8), "RGB")
label = torch.zeros(1, 20); label[0, 3] = 1; label[0, 11] = 1
ms = MSFStager(dev, max_side=1100)
mean = np.array([[[0.485, 0.456, 0.406]]]); std = np.array([[[0.229, 0.224, 0.225]]])


def host_list(im):
    out = []
    for s in (0.5, 1.0, 1.5, 2.0):
        a = np.asarray(im.resize((round(im.size[0] * s), round(im.size[1] * s)), resample=PIL.Image.BICUBIC))
        x = np.transpose((a / 255 - mean) / std, (2, 0, 1))
        out += [torch.from_numpy(x.copy())[None], torch.from_numpy(np.flip(x, -1).copy())[None]]
    return [t.to(dev).float() for t in out]


for name, build in (("MSF list on the device", lambda: ms(pil)), ("MSF list on the host (PIL, one core)", lambda: host_list(pil))):
    for _ in range(2):
        infer.infer_cam(model, build(), label, 375, 500)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        infer.infer_cam(model, build(), label, 375, 500)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(f"infer_mcl loop body, one 500x375 image, scales 0.5/1/1.5/2 x flip, {a.model}, {name}: {dt*1e3:7.1f} ms per image  {1/dt:5.1f} img/s", flush=True)
