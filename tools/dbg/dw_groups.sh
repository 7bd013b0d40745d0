mkdir -p gpurun_out; rm -f gpurun_out/dw_groups.txt
for g in 512 768 1024 1536 2048 4096; do
  echo "== MX_DW_GROUPS=$g" >> gpurun_out/dw_groups.txt
  MX_DW_GROUPS=$g timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep dwfused >> gpurun_out/dw_groups.txt
done
for t in 0 1; do
  echo "== MX_DW_FUSED_TILE=$t (default groups)" >> gpurun_out/dw_groups.txt
  MX_DW_FUSED_TILE=$t timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep dwfused >> gpurun_out/dw_groups.txt
done
echo "== MX_DW_FUSED_TILE=1 MX_DW_GROUPS=4096" >> gpurun_out/dw_groups.txt
MX_DW_FUSED_TILE=1 MX_DW_GROUPS=4096 timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep dwfused >> gpurun_out/dw_groups.txt
cat gpurun_out/dw_groups.txt
