set -e
bash tools/dbg/env_ab.sh MUSCLE_FOLD_BN0_BOTH 0 1 2
for g in 16 18 20 24; do
  MX_WGRAD_WS_GROUPS=$g timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-arith --no-configs > gpurun_out/ws_g.json 2>gpurun_out/ws_g.err
  echo "MX_WGRAD_WS_GROUPS=$g $(python -c "import json;d=json.loads(open('gpurun_out/ws_g.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])")" | tee -a gpurun_out/env_ab.txt
done
timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-arith --no-configs > gpurun_out/ws_g.json 2>gpurun_out/ws_g.err
echo "default $(python -c "import json;d=json.loads(open('gpurun_out/ws_g.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])")" | tee -a gpurun_out/env_ab.txt
