#!/bin/bash
# round 2, GPU visit 22 (the last seconds of the budget): the round-3 candidate k_down32wsd (deferred epilogue, debug build, DVAE_DOWN_D=1)
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
for d in 0 1 0 1; do echo -n "DVAE_DOWN_D=$d  "; DVAE_DOWN_D=$d timeout 30 python tools/kone.py 2>&1 | tail -n 1; done | tee gpurun_out/down_d.txt
DVAE_DOWN_D=1 timeout 60 python -m pytest tests/test_gpu_bench_sizes.py -m gpu -q --timeout=50 --no-header -x -k "test_conv_persistent_loops and 261 or test_conv_persistent_loops and 1027 or test_convT_persistent_loops and 261 or test_convT_persistent_loops and 1027" 2>&1 | tail -n 3 | tee -a gpurun_out/down_d.txt
