# usage: env_sweep.sh VAR v1 v2 ...  - one bench.py run (20 steps) per value of VAR on one box
mkdir -p gpurun_out
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-arith --no-configs > gpurun_out/sw.json 2>gpurun_out/sw.err
  echo "$var=$v $(python -c "import json;d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])" 2>&1 | tail -1)" | tee -a gpurun_out/env_sweep.txt
done
