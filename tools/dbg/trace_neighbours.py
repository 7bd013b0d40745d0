"""Which kernels sit around each `__amd_rocclr_copyBuffer` / torch fill in a rocprofv3 kernel trace (who launches them?)."""
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
ctx = collections.Counter()
for i, n in enumerate(names):
    if pat in n:
        prev = next((names[j] for j in range(i - 1, -1, -1) if pat not in names[j]), "-")
        nxt = next((names[j] for j in range(i + 1, len(names)) if pat not in names[j]), "-")
        ctx[(prev[:60], nxt[:60])] += 1
for (p, n), c in ctx.most_common(25):
    print(f"{c:5d}  after {p:60s} before {n}")
