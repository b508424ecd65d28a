#!/bin/bash
# round 2, GPU visit 20 (last minutes): k_down32ws with TWO loader waves per SIMD (768-thread workgroups, debug build: DVAE_DOWN_LT=512)
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
for lt in 256 512 256 512; do echo -n "DVAE_DOWN_LT=$lt  "; DVAE_DOWN_LT=$lt timeout 60 python tools/kone.py 2>&1 | tail -n 1; done | tee gpurun_out/down_lt.txt
DVAE_DOWN_LT=512 timeout 200 python -m pytest tests/test_gpu_bench_sizes.py tests/test_gpu_kernels.py -m gpu -q --timeout=120 --no-header -x -k "conv_persistent or convT_persistent or test_conv or test_convT" 2>&1 | tail -n 3 | tee -a gpurun_out/down_lt.txt
