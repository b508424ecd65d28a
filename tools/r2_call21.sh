#!/bin/bash
# round 2, GPU visit 21: whole step with DVAE_DOWN_LT=256 / 512 (debug build)
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
bench() { timeout 100 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
for lt in 256 512 256 512; do echo -n "btcvae_celeba DVAE_DOWN_LT=$lt: "; DVAE_DOWN_LT=$lt bench; done | tee gpurun_out/down_lt_step.txt
