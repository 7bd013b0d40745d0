cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout -k 10 330 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_r4_$c -- python3 bench.py --no-cpu-baseline --no-other-arith --no-configs --steps 2 --warmup 1 > gpurun_out/pmc_r4_$c.log 2>&1 || exit 1
  echo done $c
done
python tools/summarize_pmc.py gpurun_out/r04_pmc_per_kernel.json fetch=$(ls -d gpurun_out/pmc_r4_FETCH_SIZE) write=$(ls -d gpurun_out/pmc_r4_WRITE_SIZE) mfma=$(ls -d gpurun_out/pmc_r4_SQ_VALU_MFMA_BUSY_CYCLES) steps=5 | tail -40
ls gpurun_out/*.json
