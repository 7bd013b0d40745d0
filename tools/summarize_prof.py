"""Condense a rocprofv3 --kernel-trace --stats run (kernel_stats.csv) into a committed summary."""
import csv, sys, glob, os
src = sys.argv[1]; out = sys.argv[2]; steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
f = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out, "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(f)}); {steps:g} steps profiled (incl. warm-up)\n")
    o.write(f"# total kernel time {tot/1e6:.2f} ms = {tot/1e6/steps:.2f} ms/step\n")
    o.write("name,calls,total_ms,ms_per_step,avg_us,min_us,max_us,percent\n")
    for r in rows:
        if float(r["Percentage"]) < 0.05: continue
        o.write(f"\"{r['Name'][:120]}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['TotalDurationNs'])/1e6/steps:.3f},"
                f"{float(r['AverageNs'])/1e3:.1f},{float(r['MinNs'])/1e3:.1f},{float(r['MaxNs'])/1e3:.1f},{float(r['Percentage']):.2f}\n")
print(open(out).read()[:6000])
