# launch-geometry table of configs[1] (B0 / 448 / batch 16), side stream off (tools/occupancy_table.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/occ_b0
MUSCLE_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/occ_b0 -- python3 bench.py --model efficientnet-b0 --batch 16 --no-cpu-baseline --no-other-arith --no-configs --steps 20 --warmup 5 > gpurun_out/occ_b0.log 2>&1 || { echo b0 failed; tail -5 gpurun_out/occ_b0.log; exit 1; }
python tools/occupancy_table.py gpurun_out/occ_b0 28 80 > gpurun_out/occ_b0.txt
find gpurun_out/occ_b0 -name "*_kernel_trace.csv" -delete
tail -c 300 gpurun_out/occ_b0.log; head -70 gpurun_out/occ_b0.txt
