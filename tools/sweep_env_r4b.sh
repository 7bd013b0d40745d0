# round 4, second sweep: host-side / dispatch thresholds with the new kernels, in the step
run() { env "$@" python bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 15 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2), round(d['value'],1))"; }
run A=1
run MUSCLE_MATERIALISE_ABOVE=64
run MUSCLE_MATERIALISE_ABOVE=224
run MX_DW_FUSED_TILE=1
run MX_SPLIT3_XCD_CHUNK=8
run MX_SPLIT3_XCD_CHUNK=4
run A=2
run MX_DW_GROUPS=2048
run MX_COLREDUCE_BLOCKS=2048
run MX_STREAM_BLOCKS=8192
run MUSCLE_FOLD_BN0_BOTH=1
run A=3
