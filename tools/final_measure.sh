# The round's closing measurements on ONE box: the driver-style line, the per-shape GEMM table, the standalone kernel table
# (side stream off) and the three PMC passes.  usage: final_measure.sh TAG   (outputs under gpurun_out/TAG_*)
tag=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --steps 20 --warmup 5 --gemm-table gpurun_out/${tag}_gemm_table.txt > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo bench failed; tail -5 gpurun_out/${tag}_bench.err; exit 1; }
echo "bench done: $(python -c "import json;d=json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])")"
rm -rf gpurun_out/${tag}_stats
MUSCLE_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 5 --warmup 2 > gpurun_out/${tag}_stats.log 2>&1 || { echo stats failed; exit 1; }
python tools/summarize_prof.py gpurun_out/${tag}_stats gpurun_out/${tag}_split_noside_kernel_stats.csv 10 > /dev/null
echo stats done
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rm -rf gpurun_out/pmc_${tag}_$c
  timeout -k 10 330 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -- python3 bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 2 --warmup 1 > gpurun_out/pmc_${tag}_$c.log 2>&1 || exit 1
  echo done $c
done
python tools/summarize_pmc.py gpurun_out/${tag}_pmc_per_kernel.json fetch=gpurun_out/pmc_${tag}_FETCH_SIZE write=gpurun_out/pmc_${tag}_WRITE_SIZE mfma=gpurun_out/pmc_${tag}_SQ_VALU_MFMA_BUSY_CYCLES steps=5 | tail -30
