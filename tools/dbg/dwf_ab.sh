mkdir -p gpurun_out; rm -f gpurun_out/dwf_ab.txt
timeout -k 10 300 python -m pytest tests/test_gpu_dwfused.py -q -x 2>&1 | tail -4 >> gpurun_out/dwf_ab.txt
MX_DWF_WIDE=0 timeout -k 10 300 python -m pytest tests/test_gpu_dwfused.py -q -x -k forward 2>&1 | tail -2 >> gpurun_out/dwf_ab.txt
for rep in 1 2; do for w in 0 1; do
  echo "== MX_DWF_WIDE=$w" >> gpurun_out/dwf_ab.txt
  MX_DWF_WIDE=$w timeout -k 10 200 python tools/microbench.py dw 2>&1 | grep "^dw" | sed 's/ | bwd_data.*//' >> gpurun_out/dwf_ab.txt
done; done
cat gpurun_out/dwf_ab.txt
