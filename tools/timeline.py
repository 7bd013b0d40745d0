"""Per-step timeline of a rocprofv3 --kernel-trace CSV of bench.py: wall time, per-queue kernel-time sums and busy
(union) time for the forward and the backward phase, and the per-kernel totals of one step.
    python tools/timeline.py gpurun_out/prof/x_kernel_trace.csv [step]"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ad = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]      # two launches per step
main_q = collections.Counter(r["Queue_Id"] for r in rows).most_common(1)[0][0]


def busy(rs):
    ev = sorted((r["s"], r["e"]) for r in rs)
    if not ev:
        return 0
    tot, (cs, ce) = 0, ev[0]
    for s, e in ev[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


ksum = lambda rs: sum(r["e"] - r["s"] for r in rs)  # noqa: E731
nsteps = len(ad) // 2 - 1
for k in range(1, nsteps + 1):
    seg = rows[ad[2 * k - 1] + 1: ad[2 * k + 1] + 1]
    t0, t1 = seg[0]["s"], max(r["e"] for r in seg)
    q1 = [r for r in seg if r["Queue_Id"] == main_q]
    q2 = [r for r in seg if r["Queue_Id"] != main_q]
    wg = [r for r in q2 if "gemm_kernel" in r["Kernel_Name"] or "wgrad" in r["Kernel_Name"]]
    tb = wg[0]["s"] if wg else t1
    fw, bw = [r for r in q1 if r["e"] <= tb], [r for r in q1 if r["s"] >= tb]
    print(f"step {k}: wall {(t1 - t0) / 1e6:.2f} ms | launches main {len(q1)} side {len(q2)} | kernel-time sums main {ksum(q1) / 1e6:.2f} "
          f"side {ksum(q2) / 1e6:.2f} | busy (union of both queues) {busy(seg) / 1e6:.2f}")
    print(f"   forward  (until the first side-stream weight gradient): wall {(tb - t0) / 1e6:.2f}  main sum {ksum(fw) / 1e6:.2f} busy {busy(fw) / 1e6:.2f}")
    print(f"   backward: wall {(t1 - tb) / 1e6:.2f}  main sum {ksum(bw) / 1e6:.2f} busy {busy(bw) / 1e6:.2f} | side sum {ksum(q2) / 1e6:.2f} busy {busy(q2) / 1e6:.2f}")
k = int(sys.argv[2]) if len(sys.argv) > 2 else min(2, nsteps)
seg = rows[ad[2 * k - 1] + 1: ad[2 * k + 1] + 1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    n = r["Kernel_Name"].split("(")[0][:64]
    agg[n][0] += 1
    agg[n][1] += (r["e"] - r["s"]) / 1e3
print(f"\nstep {k}: per-kernel totals (durations as traced, i.e. under whatever overlap the step had)")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {n:64s} {v[0]:5d} {v[1]:9.0f} us {v[1] / v[0]:8.1f} avg")
