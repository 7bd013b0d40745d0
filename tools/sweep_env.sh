run() { env "$@" python bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$*', round(d['ms_per_step'],2))"; }
run A=1
run MUSCLE_FOLD_BN0=1
run MUSCLE_MATERIALISE_ABOVE=64
run MUSCLE_MATERIALISE_ABOVE=224
run MUSCLE_MATERIALISE_ABOVE=384
run MX_DW_GROUPS=2048
run MX_DW_GROUPS=512
run MX_COLREDUCE_BLOCKS=2048
run MX_COLREDUCE_BLOCKS=512
run MX_STREAM_BLOCKS=8192
run MX_STREAM_BLOCKS=2048
run MUSCLE_WGRAD_STREAM=0
run A=2
