# usage: env_ab_fp32.sh VAR A B [reps] - env_ab.sh on the exact-fp32 leg (bench.py --arith fp32, 10 steps)
set -e
mkdir -p gpurun_out
var=$1; a=$2; b=$3; reps=${4:-2}
for r in $(seq $reps); do for v in $a $b; do
  env $var=$v timeout -k 10 200 python bench.py --arith fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-other-arith --no-configs > gpurun_out/abf_${var}_$v.json 2>gpurun_out/abf_${var}_$v.err
  echo "fp32 $var=$v $(python -c "import json;d=json.loads(open('gpurun_out/abf_${var}_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['achieved'])")" | tee -a gpurun_out/env_ab.txt
done; done
