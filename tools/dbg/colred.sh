for b in 1024 2048 4096; do
  echo "== MX_COLREDUCE_BLOCKS=$b"
  MX_COLREDUCE_BLOCKS=$b timeout -k 10 200 python tools/microbench.py pool reduce 2>&1 | grep "^pool\|^reduce" | cut -c1-150
done
bash tools/dbg/env_sweep.sh MX_COLREDUCE_BLOCKS 1024 2048 4096 1024 2048 2>&1 | cut -c1-70
