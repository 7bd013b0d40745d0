mkdir -p gpurun_out; rm -f gpurun_out/dw_tile8.txt
MX_DW_FUSED_TILE=3 timeout -k 10 200 python -m pytest tests/test_gpu_dwfused.py -q -k "fused" 2>&1 | tail -2 >> gpurun_out/dw_tile8.txt
for rep in 1 2; do for t in default 3; do
  echo "== tile $t" >> gpurun_out/dw_tile8.txt
  if [ $t = default ]; then timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep "dwfused k3" | sed 's/ | unfused.*//' >> gpurun_out/dw_tile8.txt
  else MX_DW_FUSED_TILE=$t timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep "dwfused k3" | sed 's/ | unfused.*//' >> gpurun_out/dw_tile8.txt; fi
done; done
cat gpurun_out/dw_tile8.txt
bash tools/dbg/env_ab.sh MX_DWF_WIDE 1 0 2 2>&1 | cut -c1-60
