"""Per-shape roofline of the pointwise GEMMs of one B7 / 448x448 / batch 32 step-A (fwd + dgrad + wgrad of every 1x1
conv up to the last tap), grouped by distinct (M, K, N): count, MFMA time at 157.3 TFLOP/s, minimum HBM time at
6.3 TB/s (each operand and the output moved once) and the roofline sum.  CPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muscle_amd.arch import net_cfg
from collections import OrderedDict

def shapes(name="efficientnet-b7", N=32, size=448, last_pooling=False):
    cfg = net_cfg(name, last_pooling)
    h = cfg.stem_out_size(size)
    out = OrderedDict()
    def add(kind, M, K, Nn):
        out[(kind, M, K, Nn)] = out.get((kind, M, K, Nn), 0) + 1
    add("stem", N * h * h, 28, cfg.stem_out)
    for b in cfg.blocks:
        ho = b.out_size(h)
        if b.index > cfg.taps[6]:
            break
        if b.expand:
            add("expand", N * h * h, b.cin, b.cexp)
        add("project", N * ho * ho, b.cexp, b.cout)
        h = ho
    return out

if __name__ == "__main__":
    PEAK, BW = 157.3e12, 6.3e12
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totf = 0.0
    print(f"{'kind':8s} {'M':>8s} {'K':>5s} {'N':>5s} {'cnt':>3s} | per-launch us: mfma  hbm(fwd) hbm(dgrad) hbm(wgrad) | roof ms (all launches)")
    for (kind, M, K, Nn), c in shapes().items():
        fl = 2.0 * M * K * Nn
        t_m = fl / PEAK
        b_f = 4.0 * (M * K + K * Nn + M * Nn)
        t = {"fwd": max(t_m, b_f / BW), "dgrad": max(t_m, b_f / BW), "wgrad": max(t_m, 4.0 * (M * K + M * Nn + K * Nn) / BW)}
        if kind == "stem":
            t["dgrad"] = 0.0
        for k in tot:
            tot[k] += c * t[k]
        totf += c * fl * (2 if kind == "stem" else 3)
        print(f"{kind:8s} {M:8d} {K:5d} {Nn:5d} {c:3d} | {t_m*1e6:8.1f} {b_f/BW*1e6:8.1f} | {c*(t['fwd']+t['dgrad']+t['wgrad'])*1e3:7.2f}")
    print("roofline sums (ms):", {k: round(v * 1e3, 2) for k, v in tot.items()}, "total", round(sum(tot.values()) * 1e3, 2),
          "| GFLOP/step", round(totf / 1e9, 1), "| pure-MFMA ms", round(totf / PEAK * 1e3, 2))
