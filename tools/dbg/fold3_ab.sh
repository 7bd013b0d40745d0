set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split.py -x -q -k "side_fold or bnbwd" 2>&1 | tail -3
bash tools/dbg/env_ab.sh MUSCLE_FOLD_BN0_WGRAD 0 1 2
