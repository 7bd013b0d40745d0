# launch-geometry table of the headline step (B7 / 448 / batch 32), side stream off (tools/occupancy_table.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/occ_b7
MUSCLE_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/occ_b7 -- python3 bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 8 --warmup 2 > gpurun_out/occ_b7.log 2>&1 || { echo b7 failed; tail -5 gpurun_out/occ_b7.log; exit 1; }
python tools/occupancy_table.py gpurun_out/occ_b7 10 90 > gpurun_out/occ_b7.txt
find gpurun_out/occ_b7 -name "*_kernel_trace.csv" -delete
head -64 gpurun_out/occ_b7.txt
