#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
TAG=${TAG:-r04_v31}
{
  for a in 0 32 64 0 32 64; do DVAE_THIN_WS_ABLATE=$a timeout 120 python tools/thin_ab.py 1024 2>&1 | grep "bits\|bit mask"; done
  DVAE_THIN_WS_GRID=256 timeout 120 python tools/thin_ab.py 1024 2>&1 | grep -v wgrad
} | grep -v amdgpu.ids | tee gpurun_out/${TAG}_thin_ab.txt
