#!/bin/bash
# round-4 visit: bit-plane ReLU masks
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
T=gpurun_out/${TAG:-r04_v11}
timeout 900 python -m pytest tests/test_gpu_mask_bits.py -m gpu -q --timeout=300 --no-header -x 2>&1 | tail -n 15
BA="--no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for c in btcvae_celeba factor_celeba btcvae_dsprites factor_dsprites; do
  timeout 200 python bench.py --config $c --steps 80 --warmup 20 $BA 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'])"
done | tee ${T}_configs.txt
timeout 200 python bench.py --batch 128 --steps 100 --warmup 20 $BA 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae B=128', d['value'], d['ms_per_step'])" | tee -a ${T}_configs.txt
timeout 300 python tools/kbench.py 1024 2>&1 | grep -v amdgpu.ids > ${T}_kbench.txt; grep -E "conv1|convT3|conv dgrad|convT fwd" ${T}_kbench.txt | head -20
