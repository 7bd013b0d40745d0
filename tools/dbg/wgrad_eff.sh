for e in 0.8 0.55; do for f in 1 0; do
  echo "== MX_WGRAD_SPLIT_EFF=$e MX_WGRAD_SPLIT_FILL=$f"
  MX_WGRAD_SPLIT_EFF=$e MX_WGRAD_SPLIT_FILL=$f timeout -k 10 200 python tools/time_wgrad.py 2>&1 | grep "Co=960\|Co=160\|Co=480\|Co=80 \|Co=1344\|Co=224" | sed 's/ | atomic.*//'
done; done
MX_WGRAD_SPLIT_EFF=0.55 timeout -k 10 300 python -m pytest tests/test_gpu_split.py tests/test_gpu_wgrad.py -x -q 2>&1 | tail -2
for r in 1 2; do
for cfg in "0.8 1" "0.55 1" "0.55 0"; do set -- $cfg
  MX_WGRAD_SPLIT_EFF=$1 MX_WGRAD_SPLIT_FILL=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-arith --no-configs > gpurun_out/we.json 2>/dev/null
  echo "eff=$1 fill=$2 $(python -c "import json;d=json.loads(open('gpurun_out/we.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['gemm_ms_per_step'],d['roofline']['achieved'])")"
done; done
