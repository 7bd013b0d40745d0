# kernel tables of BASELINE configs[3] (decoder step, bench_dec.py) and configs[1] (B0 / 448 / batch 16), side stream off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/r5_dec_stats gpurun_out/r5_b0_stats
MUSCLE_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_dec_stats -- python3 tools/bench_dec.py > gpurun_out/r5_dec_stats.log 2>&1 || { echo dec failed; tail -5 gpurun_out/r5_dec_stats.log; exit 1; }
python tools/summarize_prof.py gpurun_out/r5_dec_stats gpurun_out/r05_config4_dec_kernel_stats.csv 12 > /dev/null
MUSCLE_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_b0_stats -- python3 bench.py --model efficientnet-b0 --batch 16 --no-cpu-baseline --no-other-arith --no-configs --steps 20 --warmup 5 > gpurun_out/r5_b0_stats.log 2>&1 || { echo b0 failed; tail -5 gpurun_out/r5_b0_stats.log; exit 1; }
python tools/summarize_prof.py gpurun_out/r5_b0_stats gpurun_out/r05_config2_b0_kernel_stats.csv 28 > /dev/null
find gpurun_out/r5_dec_stats gpurun_out/r5_b0_stats -name "*_kernel_trace.csv" -delete
tail -3 gpurun_out/r5_dec_stats.log; tail -c 400 gpurun_out/r5_b0_stats.log
