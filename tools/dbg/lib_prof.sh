# usage: lib_prof.sh A.so B.so pattern - kernel-trace stats of a short bench.py run with each build; prints the rows matching pattern
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in $1 $2; do
  n=$(basename $v .so); rm -rf gpurun_out/lp_$n
  export MUSCLE_HIP_LIB=$(realpath $v)
  MUSCLE_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lp_$n -- python3 bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 8 --warmup 2 $BENCH_ARGS > gpurun_out/lp_$n.log 2>&1 || { echo failed; tail -5 gpurun_out/lp_$n.log; exit 1; }
  python tools/summarize_prof.py gpurun_out/lp_$n gpurun_out/lp_$n.csv 10 > /dev/null
  find gpurun_out/lp_$n -name "*_kernel_trace.csv" -delete
  echo "== $n"; head -3 gpurun_out/lp_$n.csv | tail -2; grep -E "$3" gpurun_out/lp_$n.csv | cut -c1-60,100-
done
