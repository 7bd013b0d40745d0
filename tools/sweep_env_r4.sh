# round 4: the side-stream knobs around their defaults, IN THE STEP (weight gradients beside the backward chain)
run() { env "$@" python bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 15 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2), round(d['value'],1))"; }
run A=1
run MX_WGRAD_TILE_GROUPS=8
run MX_WGRAD_TILE_GROUPS=16
run MX_WGRAD_TILE_RMIN=1
run MX_WGRAD_TILE_RMIN=3
run A=2
run MX_WGRAD_TILE_PERSIST=1
run MX_WGRAD_TILE_PERSIST=2
run MX_WGRAD_SMALL_RMIN=16384
run MX_WGRAD_SMALL_RMIN=262144
run MUSCLE_WGRAD_PRIO=-1
run A=3
