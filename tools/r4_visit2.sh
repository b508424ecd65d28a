#!/bin/bash
# round-4 visit: k_gdma v2 variants (debug build)
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
T=gpurun_out/${TAG:-r04_v5}
one() { timeout 100 python tools/gemm_ab.py 2048 2>&1 | grep -E "fwd" | cut -c25-100; }
{
echo "default (loader prio 2):";            one
echo "ABLATE=8 (loaders prio 0):"; DVAE_GDMA_ABLATE=8 one
echo "GEO=4 (8 loader waves):";    DVAE_GDMA_GEO=4 one
echo "default:";            one
echo "GEO=4 (8 loader waves):";    DVAE_GDMA_GEO=4 one
} | tee ${T}_variants.txt
timeout 150 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids | tee ${T}_gemm_ab.txt
