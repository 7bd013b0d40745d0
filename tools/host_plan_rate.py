"""Host side of the input path: images per second per core of `plan_item` (decode excluded / included) on the fixture
JPEGs - what sizes the DataLoader worker pool of muscle_amd.data.StagedLoader (CPU only)."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, PIL.Image
import golden_util as gu
from muscle_amd import data as D
G = gu.load("input_views.npz")
raw = [G[f"jpg{i}"].tobytes() for i in range(6)]
torch.set_num_threads(1)
for aug, devj, devr in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 5.0:
        im = PIL.Image.open(io.BytesIO(raw[n % 6])).convert("RGB")
        D.plan_item(im, augment=aug, device_jitter=devj, device_resize=devr); n += 1
    dt = time.perf_counter() - t0
    print(f"plan_item incl. JPEG decode, augment={aug}, ColorJitter on the {'device' if devj else 'host (PIL)'}, resize on the "
          f"{'device' if devr else 'host (PIL)'}: {n / dt:6.1f} img/s on one core "
          f"({dt / n * 1e3:.1f} ms per image)")
