cp muscle_amd/libmuscle_hip.so /tmp/keep.so
for v in div se; do cp tools/hip/build/lib_$v.so muscle_amd/libmuscle_hip.so; echo "== lib_$v"; timeout -k 10 100 python tools/dbg/se_time.py; done
cp /tmp/keep.so muscle_amd/libmuscle_hip.so
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -x -q -k "se or SE or excit" 2>&1 | tail -2
