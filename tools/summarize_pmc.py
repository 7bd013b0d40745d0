"""Merge rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES; each run with --kernel-trace) into a
per-kernel-family table: launches, mean duration, HBM-side bytes per launch (FETCH_SIZE x2 per the gfx950 correction for
16-B/lane reads, MI355X_MICROARCH.md 'HBM'), GB/s, MFMA pipe utilisation."""
import csv, glob, json, os, re, sys, collections

def load(d):
    cc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    rows = list(csv.DictReader(open(cc)))
    # everything before the first step's first kernel (the stem's im2col) is set-up: ~1 700 `__amd_rocclr_copyBuffer` of the model's
    # parameters going to the device and the generators' fills - round 4 counted them into `launches_per_step_all` (172 + 20 "per step")
    first = min((int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"].startswith("stem_im2col")), default=0)
    for r in rows:
        if int(r["Dispatch_Id"]) < first:
            continue
        fam = family(r["Kernel_Name"])
        out[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], r["Counter_Name"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[fam] += 1
            out[fam]["_dur"] += dur.get(r["Dispatch_Id"], 0.0)
    return out, cnt

def family(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if n.startswith("gemm_kernel"):
        m = re.match(r"gemm_kernel<(\d)", n)
        return "gemm_kernel<%s>" % {"0": "NT fwd", "1": "NN dgrad", "2": "TN wgrad"}[m.group(1)]
    if n.startswith("gemm_nt_kernel"):
        return "gemm_nt_kernel<NT fwd + dgrad>"
    if n.startswith("gemm_nt_split3_kernel"):
        return "gemm_nt_split3_kernel<NT fwd + dgrad, split arithmetic, weight planes>"
    if n.startswith("gemm_nt_split_kernel"):
        return "gemm_nt_split_kernel<NT fwd + dgrad, split arithmetic>"
    if n.startswith(("wgrad_split_ws_kernel", "wgrad_split_pipe_kernel")):
        return "wgrad_split_kernel"                      # round 5: the split weight gradient's kernels are one family
    if n.startswith("dw_bwd_fused_kernel"):
        return "dw_bwd_fused_kernel<%s>" % re.match(r"dw_bwd_fused_kernel<(\d)", n).group(1)
    return re.sub(r"<.*", "", n)

GEMM_FAMILIES = ("gemm_kernel", "gemm_nt_kernel", "gemm_nt_split_kernel", "gemm_nt_split3_kernel", "wgrad_small_kernel", "wgrad_tile_kernel", "wgrad_split_kernel")
GEMM_AUX = ("wgrad_parts_reduce_kernel",)      # bytes belong to the weight-gradient GEMMs, launches are not counted

dirs = dict(a.split("=") for a in sys.argv[2:])
steps_profiled = int(dirs.pop("steps", "0"))        # steps=N: the passes cover N whole steps -> whole-step traffic and launch count
res = {}
fetch, cf = load(dirs["fetch"]); write, _ = load(dirs["write"]); mfma, cm = load(dirs["mfma"])
for fam in sorted(cf, key=lambda f: -fetch[f]["_dur"]):
    n = cf[fam]
    d = fetch[fam]["_dur"] / n
    fb = 2.0 * fetch[fam].get("FETCH_SIZE", 0.0) * 1024 / n          # KB -> B, x2 correction
    wb = write[fam].get("WRITE_SIZE", 0.0) * 1024 / max(1, n)
    dm = mfma[fam]["_dur"] / max(1, cm.get(fam, 1))
    busy = mfma[fam].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1, cm.get(fam, 1))
    util = busy / (dm * 2.4e9 * 1024) if dm > 0 else 0.0             # 1024 SIMDs, 2.4 GHz
    if fetch[fam]["_dur"] / sum(v["_dur"] for v in fetch.values()) < 0.004: continue
    res[fam] = {"launches": n, "avg_us": round(d * 1e6, 1), "read_MB_per_launch": round(fb / 1e6, 1), "write_MB_per_launch": round(wb / 1e6, 1),
                "hbm_side_GBps": round((fb + wb) / d / 1e9, 0) if d > 0 else None, "mfma_util": round(util, 3)}
json.dump({"source": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES}, three passes over "
           "bench.py --steps 2 --warmup 1 (kernels are serialised and slower under counter collection; durations are from the "
           "FETCH pass); FETCH_SIZE doubled (gfx950: 128-B read requests tallied at 64 B); Infinity-Cache hits are counted "
           "as HBM-side traffic by these counters; mfma_util = MFMA busy cycles / (duration x 1024 SIMDs x 2.4 GHz)",
           "kernels": res}, open(sys.argv[1], "w"), indent=1)
print(open(sys.argv[1]).read())
# the bench line's roofline.traffic: memory-side bytes per pointwise-GEMM launch (all GEMM families together)
gl = sum(cf[f] for f in cf if f.startswith(GEMM_FAMILIES))
gb = sum(2.0 * fetch[f].get("FETCH_SIZE", 0.0) * 1024 + write[f].get("WRITE_SIZE", 0.0) * 1024 for f in cf if f.startswith(GEMM_FAMILIES + GEMM_AUX))
if len(sys.argv) > 1 and gl:
    tp = os.path.join(os.path.dirname(sys.argv[1]), "traffic.json")
    extra = {}
    if steps_profiled > 0:
        tot = sum(2.0 * fetch[f].get("FETCH_SIZE", 0.0) * 1024 + write[f].get("WRITE_SIZE", 0.0) * 1024 for f in cf)
        extra = {"steps_profiled": steps_profiled, "step_traffic_GB": tot / steps_profiled / 1e9,
                 "launches_per_step_all": sum(cf.values()) / steps_profiled,
                 "step_traffic_is": "sum over ALL kernels of 2 x FETCH_SIZE + WRITE_SIZE (memory-side of L2: Infinity-Cache hits included) per step"}
    json.dump({"source": "tools/summarize_pmc.py over the three --pmc passes named in " + os.path.basename(sys.argv[1]), **extra,
               "gemm_launches_profiled": gl, "gemm_hbm_bytes_per_launch": gb / gl,
               "per_family_MB_per_launch": {f: {"read": res[f]["read_MB_per_launch"], "write": res[f]["write_MB_per_launch"], "mfma_util": res[f]["mfma_util"]}
                                            for f in res if f.startswith(GEMM_FAMILIES)}}, open(tp, "w"), indent=1)
    print("wrote", tp, gb / gl / 1e6, "MB per GEMM launch over", gl, "launches")
