cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/r5a_stats
MUSCLE_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5a_stats -- python3 bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 5 --warmup 2 > gpurun_out/r5a_stats.log 2>&1 || { echo stats failed; tail -5 gpurun_out/r5a_stats.log; exit 1; }
python tools/summarize_prof.py gpurun_out/r5a_stats gpurun_out/r05_a_split_noside_kernel_stats.csv 10 > /dev/null
python tools/dbg/trace_neighbours.py gpurun_out/r5a_stats copyBuffer > gpurun_out/r5a_copybuffer.txt
python tools/dbg/trace_neighbours.py gpurun_out/r5a_stats FillFunctor > gpurun_out/r5a_fill.txt
find gpurun_out/r5a_stats -name "*_kernel_trace.csv" -size +30M -delete
head -30 gpurun_out/r5a_copybuffer.txt
