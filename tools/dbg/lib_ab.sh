# usage: lib_ab.sh A.so B.so [reps] - alternating bench.py runs (20 steps) of two builds of the library on one box.
# BENCH_ARGS adds bench.py flags (e.g. "--model efficientnet-b0 --batch 16"), STEPS the timed steps.
# The builds are selected through MUSCLE_HIP_LIB (muscle_amd/_lib.py); the in-tree library is never overwritten.
set -e
mkdir -p gpurun_out
a=$1; b=$2; reps=${3:-2}
for r in $(seq $reps); do for v in $a $b; do
  MUSCLE_HIP_LIB=$(realpath $v) timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 4 --no-cpu-baseline --no-other-arith --no-configs $BENCH_ARGS > gpurun_out/lib_ab.json 2>gpurun_out/lib_ab.err
  echo "$(basename $v) $(python -c "import json;d=json.loads(open('gpurun_out/lib_ab.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['gemm_ms_per_step'],d['roofline']['non_gemm_ms_per_step'])")" | tee -a gpurun_out/lib_ab.txt
done; done
