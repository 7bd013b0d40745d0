# SQ stall / issue counters of the depthwise kernels under tools/microbench.py dwfused (two passes of 8 counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_dw1 gpurun_out/pmc_dw2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_dw1 -- python3 tools/microbench.py dwfused > gpurun_out/pmc_dw1.log 2>&1
echo rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_dw2 -- python3 tools/microbench.py dwfused > gpurun_out/pmc_dw2.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv,glob,collections,re
for d in ("pmc_dw1","pmc_dw2"):
    cc=glob.glob('gpurun_out/%s/**/*_counter_collection.csv'%d,recursive=True)
    if not cc: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); seen=set()
    for r in csv.DictReader(open(cc[0])):
        n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
        if not n.startswith("dw_bwd_fused"): continue
        key=n+" grid="+r.get("Grid_Size","?")
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); cnt[key]+=1
    for n in sorted(agg):
        a=agg[n]; wc=a["SQ_WAVE_CYCLES"] or 1
        print(n[:70], "n=%d"%cnt[n], " ".join("%s=%.3f"%(k.replace("SQ_",""),v/wc) for k,v in sorted(a.items()) if k!="SQ_WAVE_CYCLES"), "wave_cycles/launch=%.3g"%(wc/cnt[n]))
PY
