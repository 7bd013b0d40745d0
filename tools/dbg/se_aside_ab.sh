set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_determinism.py tests/test_gpu_config2.py -x -q > gpurun_out/se_aside_tests.log 2>&1
for v in 1 0 1 0; do
  MUSCLE_SE_PARAMS_ASIDE=$v timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-arith --no-configs > gpurun_out/se_aside_$v.json 2>gpurun_out/se_aside_$v.err
  echo "aside=$v $(python -c "import json;d=json.loads(open('gpurun_out/se_aside_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])")"
done
tail -2 gpurun_out/se_aside_tests.log
