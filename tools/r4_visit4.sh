#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
TAG=${TAG:-r04_v38}
timeout 300 python -m pytest "tests/test_gpu_kernels.py::test_btcvae_fwd_bwd" "tests/test_gpu_kernels.py::test_btcvae_kat_reference_values" tests/test_gpu_step.py -m gpu -q --timeout=300 --no-header 2>&1 | tail -n 4 | cut -c1-300 | tee gpurun_out/${TAG}_pytest.txt
for r in 1 2; do timeout 200 python tools/kbench.py 1024 2>&1 | grep btcvae; done | tee gpurun_out/${TAG}_btcvae.txt
timeout 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae_celeba', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_btcvae.txt
