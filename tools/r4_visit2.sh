#!/bin/bash
# round-4 visit: 32x32 k-split tiles for small batches (debug build)
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
T=gpurun_out/${TAG:-r04_v7}
{
for t in 32 64 32 64; do echo "DVAE_GDMA_TILE=$t"; DVAE_GDMA_TILE=$t timeout 100 python tools/gemm_ab.py 512 256 128 2>&1 | grep -E "fwd|dgrad" | cut -c25-120; done
echo "default tile choice"; timeout 100 python tools/gemm_ab.py 2048 1024 512 256 128 2>&1 | cut -c25-120
} | tee ${T}_tile_ab.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_discriminator.py -m gpu -q --timeout=300 --no-header -k "linear or discriminator" 2>&1 | tail -n 5
