# A/B of the split weight-gradient kernels in the step on ONE box: MX_WGRAD_PIPE = 0 (first kernel) / 2 (wave-specialised, round 5)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_wgrad.py -x -q 2>&1 | tail -3
bash tools/dbg/env_ab.sh MX_WGRAD_PIPE 0 2 2
