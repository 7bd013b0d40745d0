set -e
mkdir -p gpurun_out
rm -f gpurun_out/dw_new_ab.txt
cp muscle_amd/libmuscle_hip.so /tmp/lib_cur.so
for v in new pipe; do
  cp tools/hip/build/lib_$v.so muscle_amd/libmuscle_hip.so
  echo "== $v: unit test" >> gpurun_out/dw_new_ab.txt
  timeout -k 10 300 python -m pytest tests/test_gpu_dwfused.py -q 2>&1 | tail -15 >> gpurun_out/dw_new_ab.txt || true
done
for rep in 1 2; do for v in new pipe; do
  cp tools/hip/build/lib_$v.so muscle_amd/libmuscle_hip.so
  echo "== $v (rep $rep)" >> gpurun_out/dw_new_ab.txt
  timeout -k 10 200 python tools/microbench.py dwfused 2>&1 | grep dwfused >> gpurun_out/dw_new_ab.txt
done; done
cp /tmp/lib_cur.so muscle_amd/libmuscle_hip.so
cat gpurun_out/dw_new_ab.txt
