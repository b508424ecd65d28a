#!/bin/bash
# round 4: k_wgrad_thin_ws -- parity tests, A/B against k_wgrad_thin (debug build)
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
TAG=${TAG:-r04_v34}
timeout 600 python -m pytest tests/test_gpu_mask_bits.py "tests/test_gpu_kernels.py::test_conv_fwd_dgrad_wgrad" "tests/test_gpu_kernels.py::test_convT_fwd_dgrad_wgrad" tests/test_gpu_uint8_input.py -m gpu -q --timeout=300 --no-header 2>&1 | tail -n 25 | cut -c1-300 | tee gpurun_out/${TAG}_pytest.txt
{
  DVAE_THIN_WS=0 timeout 120 python tools/thin_ab.py 1024 256 2>&1 | grep wgrad
  timeout 120 python tools/thin_ab.py 1024 256 2>&1 | grep wgrad
  timeout 120 python tools/thin_ab.py --c1 1024 256 2>&1 | grep wgrad
  DVAE_THIN_WS=0 timeout 120 python tools/thin_ab.py --c1 1024 256 2>&1 | grep wgrad
} | grep -v amdgpu.ids | tee gpurun_out/${TAG}_thin_ab.txt
