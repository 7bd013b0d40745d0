# usage: env_ab.sh VAR A B [reps]  - alternating bench.py runs (20 steps) with VAR=A / VAR=B on one box
set -e
mkdir -p gpurun_out
var=$1; a=$2; b=$3; reps=${4:-2}
for r in $(seq $reps); do for v in $a $b; do
  env $var=$v timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-arith --no-configs > gpurun_out/ab_${var}_$v.json 2>gpurun_out/ab_${var}_$v.err
  echo "$var=$v $(python -c "import json;d=json.loads(open('gpurun_out/ab_${var}_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['losses'] if 'losses' in d else '')")" | tee -a gpurun_out/env_ab.txt
done; done
