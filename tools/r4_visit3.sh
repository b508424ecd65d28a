#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
T=gpurun_out/${TAG:-r04_v18}
timeout 900 python -m pytest tests/test_gpu_mask_bits.py -m gpu -q --timeout=300 --no-header 2>&1 | grep -E "passed|failed|FAILED" | cut -c1-250 | head -40
timeout 300 python tools/kbench.py 1024 2>&1 | grep -v amdgpu.ids > ${T}_kbench.txt; grep -E "bits|down_thin|staged: conv dgrad \(up|staged: convT fwd" ${T}_kbench.txt | head -12
BA="--no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for v in 1 0 1 0 1 0; do
  DVAE_DEBUG=1 DVAE_MASK_BITS=$v timeout 200 python bench.py --steps 100 --warmup 10 $BA 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae_celeba DVAE_MASK_BITS=$v', d['value'], d['ms_per_step'])"
done | tee ${T}_maskbits_ab.txt
