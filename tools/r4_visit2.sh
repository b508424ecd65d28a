#!/bin/bash
# round-4 visit: narrow discriminator layers alone, A/B against the generic paths (debug build), and in the factor_celeba step
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
T=gpurun_out/${TAG:-r04_v10}
for v in 1 0; do DVAE_NARROW=$v timeout 100 python tools/gemm_ab.py --narrow 2050 1024 256 2>&1 | grep -v amdgpu; done | tee ${T}_narrow_ab.txt
BA="--no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for v in 1 0 1 0; do
  DVAE_NARROW=$v timeout 200 python bench.py --config factor_celeba --steps 60 --warmup 15 $BA 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('factor_celeba DVAE_NARROW=$v', d['value'], d['ms_per_step'])"
done | tee ${T}_factor.txt
