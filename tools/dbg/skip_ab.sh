timeout -k 10 200 python -m pytest tests/test_gpu_dwfused.py -q 2>&1 | tail -2
cp muscle_amd/libmuscle_hip.so /tmp/keep.so
for rep in 1 2; do for v in base skip; do cp tools/hip/build/lib_$v.so muscle_amd/libmuscle_hip.so; echo "== $v"; timeout -k 10 100 python tools/microbench.py dwfused 2>&1 | grep "dwfused k5" | sed 's/ | unfused.*//'; done; done
cp /tmp/keep.so muscle_amd/libmuscle_hip.so
bash tools/dbg/lib_ab.sh tools/hip/build/lib_base.so tools/hip/build/lib_skip.so 2
