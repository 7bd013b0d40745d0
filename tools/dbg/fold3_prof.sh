cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do
  rm -rf gpurun_out/f3_$v
  MUSCLE_FOLD_BN0_WGRAD=$v MUSCLE_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f3_$v -- python3 bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 5 --warmup 2 > gpurun_out/f3_$v.log 2>&1 || { echo failed $v; exit 1; }
  python tools/summarize_prof.py gpurun_out/f3_$v gpurun_out/f3_${v}_stats.csv 10 > /dev/null
  find gpurun_out/f3_$v -name "*_kernel_trace.csv" -delete
  head -2 gpurun_out/f3_${v}_stats.csv | tail -1
  grep -E "wgrad_split_ws|bn_bwd_apply|gemm_nt_split3_kernel<8, 0>|gemm_nt_split3_kernel<6, 0>|parts_reduce" gpurun_out/f3_${v}_stats.csv | cut -c1-110
done
