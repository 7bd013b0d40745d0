# SQ stall / issue counters of the forward-side HBM-bound kernels (depthwise forward, squeeze pooling) under tools/microbench.py dw pool
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fwd1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_fwd1 -- python3 tools/microbench.py dw pool > gpurun_out/pmc_fwd1.log 2>&1
echo rc=$?
grep "^dw\|^pool" gpurun_out/pmc_fwd1.log | head -30
python3 - <<'PY'
import csv,glob,collections,re
cc=glob.glob('gpurun_out/pmc_fwd1/**/*_counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); seen=set()
for r in csv.DictReader(open(cc[0])):
    n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
    if not (n.startswith("dw_fwd") or n.startswith("colreduce_kernel<FPool") or n.startswith("se_bn1_pool")): continue
    key=n[:44]+" grid="+r.get("Grid_Size","?")
    agg[key][r["Counter_Name"]]+=float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); cnt[key]+=1
for n in sorted(agg):
    a=agg[n]; wc=a["SQ_WAVE_CYCLES"] or 1
    print(n, "n=%d"%cnt[n], " ".join("%s=%.3f"%(k.replace("SQ_",""),v/wc) for k,v in sorted(a.items()) if k!="SQ_WAVE_CYCLES"))
PY
