mkdir -p gpurun_out
for v in a b; do
  extra=""; [ $v = b ] && extra="--gemm-table gpurun_out/cfgchk_table.txt"
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $extra > gpurun_out/cfgchk_$v.json 2> gpurun_out/cfgchk_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/cfgchk_$v.json').read().strip().splitlines()[-1])
c=d['configs']
print('$v', round(d['ms_per_step'],2), 'config2', round(c['config2']['ms_per_step'],2), 'stepfull', round(c['stepfull']['ms_per_step'],1), 'config4', round(c['config4_1gpu']['ms_per_step'],1), 'config5', round(c['config5']['448']['ms_per_batch'],1))
PY
done
