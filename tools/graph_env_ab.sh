run() { env "$@" timeout -k 10 200 python bench.py --model efficientnet-b0 --batch 16 --size 448 --steps 100 --warmup 10 --no-cpu-baseline --no-other-arith --no-configs $G 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$G', '$*', round(d['value'],1), round(d['ms_per_step'],3))"; }
G=""; run A=1
G="--graph"; run A=1
G="--graph"; run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
G="--graph"; run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
G="--graph"; run DEBUG_HIP_FORCE_GRAPH_QUEUES=4 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
G="--graph"; run DEBUG_HIP_FORCE_GRAPH_QUEUES=8 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
G=""; run A=1
