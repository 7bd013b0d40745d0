cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_lab -- tools/hip/build/gemm_lab time > gpurun_out/pmc_lab.log 2>&1
echo rc=$?
python - <<'PY'
import csv,glob,collections,re
cc=glob.glob('gpurun_out/pmc_lab/**/*_counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); seen=set()
for r in csv.DictReader(open(cc)):
    n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
    agg[n][r["Counter_Name"]]+=float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); cnt[n]+=1
for n in sorted(agg,key=lambda n:-agg[n]["SQ_WAVE_CYCLES"])[:12]:
    a=agg[n]; wc=a["SQ_WAVE_CYCLES"] or 1
    print(f"{n[:55]:55s} n={cnt[n]:4d} wait_any {a['SQ_WAIT_ANY']/wc:.2f} wait_inst {a['SQ_WAIT_INST_ANY']/wc:.2f} active {a['SQ_ACTIVE_INST_ANY']/wc:.2f} valu {a['SQ_ACTIVE_INST_VALU']/wc:.2f} wait_lds {a['SQ_WAIT_INST_LDS']/wc:.2f} bankconf/wavecyc {a['SQ_LDS_BANK_CONFLICT']/wc:.3f} mfma_busy/wavecyc {a['SQ_VALU_MFMA_BUSY_CYCLES']/wc/4:.2f}")
PY
