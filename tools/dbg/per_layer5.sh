# per-layer HBM table of the depthwise / SE passes from kernel traces with the weight-gradient side stream OFF and ON
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do
  rm -rf gpurun_out/pl_$v
  MUSCLE_WGRAD_STREAM=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pl_$v -- python3 bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 3 --warmup 2 > gpurun_out/pl_$v.log 2>&1 || { echo failed $v; tail -3 gpurun_out/pl_$v.log; exit 1; }
  python tools/per_layer_hbm.py $(find gpurun_out/pl_$v -name "*_kernel_trace.csv") > gpurun_out/r05_per_layer_hbm_side$v.txt 2>&1
  find gpurun_out/pl_$v -name "*_kernel_trace.csv" -delete
  tail -2 gpurun_out/r05_per_layer_hbm_side$v.txt
done
