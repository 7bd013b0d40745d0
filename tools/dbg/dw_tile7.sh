mkdir -p gpurun_out; rm -f gpurun_out/dw_tile7.txt
MX_DW_FUSED_TILE=2 timeout -k 10 200 python -m pytest tests/test_gpu_dwfused.py -q 2>&1 | tail -3 >> gpurun_out/dw_tile7.txt
for rep in 1 2; do
for t in default 2; do
  echo "== tile $t" >> gpurun_out/dw_tile7.txt
  if [ $t = default ]; then timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep dwfused >> gpurun_out/dw_tile7.txt
  else MX_DW_FUSED_TILE=$t timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep dwfused >> gpurun_out/dw_tile7.txt; fi
done; done
echo "== tile 2 groups 1536" >> gpurun_out/dw_tile7.txt
MX_DW_GROUPS=1536 MX_DW_FUSED_TILE=2 timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep dwfused >> gpurun_out/dw_tile7.txt
cat gpurun_out/dw_tile7.txt
