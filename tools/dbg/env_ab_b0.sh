# usage: env_ab_b0.sh VAR A B [reps] - env_ab.sh on configs[1] (B0 / 448 / batch 16, 60 steps)
set -e
mkdir -p gpurun_out
var=$1; a=$2; b=$3; reps=${4:-2}
for r in $(seq $reps); do for v in $a $b; do
  env $var=$v timeout -k 10 200 python bench.py --model efficientnet-b0 --batch 16 --steps 100 --warmup 5 $B0_EXTRA --no-cpu-baseline --no-other-arith --no-configs > gpurun_out/abb0_${var}_$v.json 2>gpurun_out/abb0_${var}_$v.err
  echo "B0 $var=$v $(python -c "import json;d=json.loads(open('gpurun_out/abb0_${var}_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])")" | tee -a gpurun_out/env_ab.txt
done; done
