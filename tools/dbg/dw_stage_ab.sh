# A/B of the fused depthwise backward's staging depth (MX_DW_STAGE_MAX): the two builds are swapped in as libmuscle_hip.so
set -e
mkdir -p gpurun_out
for rep in 1 2; do
for v in ch4 ch6; do
  cp tools/hip/build/lib_$v.so muscle_amd/libmuscle_hip.so
  echo "== $v (rep $rep)" >> gpurun_out/dw_stage_ab.txt
  timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep dwfused >> gpurun_out/dw_stage_ab.txt
done
done
cat gpurun_out/dw_stage_ab.txt
